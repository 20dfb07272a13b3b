import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from types import SimpleNamespace
from oracle import denoiser as D, extra_features as XF, fixtures as fx, noise as N
import test_gpu_config2 as T2

C, T, n_nodes, dims = 5, 10, [40, 23, 36], fx.VESSEL_DIMS
top_k, seed, base, max_n = 4, 77, 9, 40
m = fx.ragged_mask(n_nodes); B, n = m.shape; md = m.cuda()
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
tr = N.Transition(N.Schedule(T, "absorbing_edges", T2.NU), [1.0], np.ones(C, np.float32) / C)
dd = D.DenoiserDims(2, dims["hidden_dims"]["n_head"], 1, C, 0)
Xn = m[..., None].float()
# oracle chain once
sampler, nm, ef, sd = T2.make_cow(T=T, max_n=max_n, C=C, dims=dims)
states = []
lab = torch.where(live, 0, -1)
for step, s in enumerate(reversed(range(T))):
    t = s + 1
    E = fx.one_hot_edges(lab.clamp(min=0), m, C)
    z = SimpleNamespace(X=Xn.cuda(), E=E.cuda(), y=torch.zeros(B, 0, device="cuda"), node_mask=md,
                        charges=torch.zeros(B, n, 0, device="cuda"), t=torch.full((B, 1), t / T, device="cuda"))
    states.append(z)
    pred = sampler.forward(z)
    u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
    r = N.reverse_step(tr, Xn, E, torch.zeros(B, n, 1), pred.E.cpu(), m, torch.full((B, 1), t), torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)
    lab = r["labels_E"]
ref_feat = [ef(z) for z in states]
ref_pred = [sampler.forward(z).E.clone() for z in states]
torch.cuda.synchronize()
bad = 0
for rep in range(40):
    if rep % 4 == 0:
        sampler, nm, ef, sd = T2.make_cow(T=T, max_n=max_n, C=C, dims=dims)     # a fresh engine, like a fresh test
    for k, z in enumerate(states):
        f = ef(z)
        p = sampler.forward(z).E
        dl = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=(k % 3) + 1, return_state=True)[0]
        torch.cuda.synchronize()
        dx = float((f.X - ref_feat[k].X).abs().max()); dy = float((f.y - ref_feat[k].y).abs().max()); de = float((f.E - ref_feat[k].E).abs().max())
        dp = float((p - ref_pred[k]).abs().max())
        if dx or dy or de or dp > 1e-5:
            bad += 1
            cols = (f.X - ref_feat[k].X).abs().amax(dim=(0, 1)).cpu().numpy().round(5)
            print("rep", rep, "state", k, "X", dx, "y", dy, "E", de, "pred", dp, "Xcols", cols, flush=True)
print("mismatches:", bad)
