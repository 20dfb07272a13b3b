"""The T=12 chain (one step) before and after leaving small-integer garbage in the caching allocator's free blocks."""
import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
from semantically_consistent_graph_generation_amd.transformer_model import model_input
C, T, top_k, seed, base = 6, 12, 4, 99, 40
model, sd = tiny_model(C, C, 1, 1)
nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
n_nodes = [9, 4, 7, 1]
m = fx.ragged_mask(n_nodes); B, n = m.shape
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
E0 = fx.one_hot_edges(torch.zeros(B, n, n, dtype=torch.long), m, C)
def run():
    sampler = DiscreteSampler(model, nm)
    out = model(model_input(m[..., None].float().cuda(), E0.cuda(), torch.full((B, 1), 1.0).cuda(), m.cuda()))
    labels, cands, mask, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=1, return_state=True)
    torch.cuda.synchronize()
    return out.E.cpu().clone(), labels.cpu().clone(), cands.cpu().clone()
a = run()
def dirty(byte):
    xs = []
    for rep in range(200):
        for sz in (64, 300, 1000, 5000, 20000, 70000, 300000, 1000000):
            xs.append(torch.full((sz,), byte, dtype=torch.uint8, device="cuda"))
    xs += [torch.full((sz,), byte, dtype=torch.uint8, device="cuda") for sz in (1 << 22, 1 << 24, 1 << 26, 1 << 28)]
    torch.cuda.synchronize(); del xs
for byte in (1, 3, 255, 0x3f):
    dirty(byte)
    b = run()
    print("free blocks filled with byte", byte, ": logits equal", torch.equal(a[0], b[0]), "max diff", float((a[0] - b[0]).abs().max()),
          "label mismatches", int((a[1] != b[1]).sum()), "cand mismatches", int((a[2] != b[2]).sum()))
