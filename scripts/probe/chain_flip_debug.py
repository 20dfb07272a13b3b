import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
C, T, top_k, seed, base = 6, 12, 4, 99, 40
model, sd = tiny_model(C, C, 1, 1)
nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
sampler = DiscreteSampler(model, nm)
n_nodes = [9, 4, 7, 1]
m = fx.ragged_mask(n_nodes)
B, n = m.shape
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
tr = N.Transition(N.Schedule(T, "absorbing_edges", NU), [1.0], np.ones(C, np.float32) / C)
dd = D.DenoiserDims(2, fx.TINY_DIMS["hidden_dims"]["n_head"], 1, C, 0)
lab = torch.where(live, 0, -1)
X = m[..., None].float()
for step, s in enumerate(reversed(range(T))):
    t = s + 1
    E = fx.one_hot_edges(lab.clamp(min=0), m, C)
    y = torch.full((B, 1), t / T)
    _, _, logits, _ = D.graph_transformer_forward(sd, dd, X, E, y, m)
    u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
    r = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), t), torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)
    dl, dc, _, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=step + 1, return_state=True)
    bad = (dl.cpu().long() != r["labels_E"]).nonzero()
    if len(bad):
        print("first mismatch after step", step, "pairs", bad.tolist()[:6])
        for b, i, j in bad.tolist()[:4]:
            p = r["prob_E"][b, i, j].double().numpy(); cdf = np.cumsum(p)
            print("  pair", (b, i, j), "oracle", int(r["labels_E"][b, i, j]), "device", int(dl[b, i, j]), "u0", float(u0[b, i, j]), "cdf", cdf.round(6), "margin", np.abs(cdf - float(u0[b, i, j])).min())
        break
    lab = r["labels_E"]
else:
    print("no mismatch")
