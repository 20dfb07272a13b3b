import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_sampling as TS
import test_gpu_reverse_step as TR
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
C, T, top_k, seed, base = 6, 12, 4, 99, 40
n_nodes = [9, 4, 7, 1]
m = fx.ragged_mask(n_nodes); B, n = m.shape
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
def both():
    model, sd = tiny_model(C, C, 1, 1)
    nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
    sampler = DiscreteSampler(model, nm)
    dl, dc, _, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=1, return_state=True)
    tr = N.Transition(N.Schedule(T, "absorbing_edges", NU), [1.0], np.ones(C, np.float32) / C)
    dd = D.DenoiserDims(2, fx.TINY_DIMS["hidden_dims"]["n_head"], 1, C, 0)
    lab = torch.where(live, 0, -1)
    X = m[..., None].float()
    s = T - 1; t = T
    E = fx.one_hot_edges(lab.clamp(min=0), m, C)
    _, _, logits, _ = D.graph_transformer_forward(sd, dd, X, E, torch.full((B, 1), t / T), m)
    u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
    r = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), t), torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)
    from semantically_consistent_graph_generation_amd.transformer_model import model_input
    dlog = model(model_input(X.cuda(), E.cuda(), torch.full((B, 1), t / T).cuda(), m.cuda())).E.cpu()
    return dl.cpu().long().clone(), r["labels_E"].clone(), logits.clone(), dlog.clone(), torch.as_tensor(u0).clone(), {k: v.clone() for k, v in sd.items()}
a = both()
print("fresh: device vs oracle label mismatches", int((a[0] != a[1]).sum()))
TR.test_absorbing_properties_at_scale()
b = both()
print("after the at-scale test: device vs oracle label mismatches", int((b[0] != b[1]).sum()))
print("device labels changed", int((a[0] != b[0]).sum()), "| oracle labels changed", int((a[1] != b[1]).sum()), "| oracle logits changed", float((a[2] - b[2]).abs().max()),
      "| device logits changed", float((a[3] - b[3]).abs().max()), "| uniforms changed", float((a[4] - b[4]).abs().max()),
      "| weights changed", max(float((a[5][k] - b[5][k]).abs().max()) for k in a[5]))
