import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from types import SimpleNamespace
from oracle import denoiser as D, extra_features as XF, fixtures as fx, noise as N
import test_gpu_config2 as T2

C, T, n_nodes, dims = 5, 10, [40, 23, 36], fx.VESSEL_DIMS
top_k, seed, base, max_n = 4, 77, 9, 40
sampler, nm, ef, sd = T2.make_cow(T=T, max_n=max_n, C=C, dims=dims)
m = fx.ragged_mask(n_nodes); B, n = m.shape; md = m.cuda()
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
tr = N.Transition(N.Schedule(T, "absorbing_edges", T2.NU), [1.0], np.ones(C, np.float32) / C)
dd = D.DenoiserDims(2, dims["hidden_dims"]["n_head"], 1, C, 0)
lab = torch.where(live, 0, -1); Xn = m[..., None].float()
for step, s in enumerate(reversed(range(T))):
    t = s + 1
    E = fx.one_hot_edges(lab.clamp(min=0), m, C)
    z = SimpleNamespace(X=Xn.cuda(), E=E.cuda(), y=torch.zeros(B, 0, device="cuda"), node_mask=md,
                        charges=torch.zeros(B, n, 0, device="cuda"), t=torch.full((B, 1), t / T, device="cuda"))
    dev = ef(z); dX, dE, dy = dev.X.cpu(), dev.E.cpu(), dev.y.cpu()
    ox, oy, oe = XF.adjacency_features(Xn, E, m)
    oevl, oevc = XF.eigen_features(E, m, pad_noise=T2.PAD(B, n))
    pred = sampler.forward(z)
    dev2 = ef(z)
    print("   ef repeat: X %.2e y %.2e E %.2e" % (float((dev2.X.cpu() - dX).abs().max()), float((dev2.y.cpu() - dy).abs().max()), float((dev2.E.cpu() - dE).abs().max())))
    tt = torch.full((B, 1), t / T)
    Xin = torch.cat([Xn, ox, dX[..., 3:]], -1); Ein = torch.cat([E, oe], -1)
    yin = torch.cat([torch.tensor(n_nodes)[:, None] / max_n, oy, oevl, tt], 1)
    yin_d = torch.cat([dy, tt], 1)
    with torch.no_grad():
        _, _, lo, _ = D.graph_transformer_forward(sd, dd, Xin, Ein, yin, m)
        _, _, ld, _ = D.graph_transformer_forward(sd, dd, torch.cat([Xn, dX], -1), torch.cat([E, dE], -1), yin_d, m)
        _, _, l3, _ = D.graph_transformer_forward(sd, dd, Xin, Ein, yin_d, m)
    pe = pred.E.cpu()
    sc = float(lo.abs().max())
    print(step, "edges", int((lab > 0).sum()) // 2, "dev-vs-oraclefeat %.2e dev-vs-devfeat %.2e oraclefeat+devy %.2e" % (
        float((pe - lo).abs().max()) / sc, float((pe - ld).abs().max()) / sc, float((pe - l3).abs().max()) / sc),
        "ydiff %.2e" % float((yin - yin_d).abs().max()), "evl", (yin - yin_d).abs().max(0).values[-11:].numpy().round(6))
    u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
    r = N.reverse_step(tr, Xn, E, torch.zeros(B, n, 1), lo, m, torch.full((B, 1), t), torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)
    lab = r["labels_E"]
    dl, dc, _, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=step + 1, return_state=True)
    print("   labels differ:", int((dl.cpu().long() != lab).sum()))
