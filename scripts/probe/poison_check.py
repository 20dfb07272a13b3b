"""Runs the T=12 chain twice: on a fresh allocator and after filling / freeing device memory with NaN patterns, so that any
kernel reading memory it (or its producer) never wrote shows up as a difference."""
import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
C, T, top_k, seed, base = 6, 12, 4, 99, 40
model, sd = tiny_model(C, C, 1, 1)
nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
n_nodes = [9, 4, 7, 1]
def run():
    sampler = DiscreteSampler(model, nm)
    labels, cands, mask, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, return_state=True)
    torch.cuda.synchronize()
    return labels.cpu().clone(), cands.cpu().clone()
a = run()
def poison(val):
    xs = [torch.full((sz,), val, device="cuda") for sz in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26)] * 3
    torch.cuda.synchronize(); del xs
for val in (float("nan"), 1e30, -7.0):
    poison(val)
    b = run()
    print("poison", val, "label mismatches", int((a[0] != b[0]).sum()), "cand mismatches", int((a[1] != b[1]).sum()))
# the same for the bigger CoW-shaped model / forward only
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
gen = torch.Generator().manual_seed(0)
B, n, Cc = 8, 20, 14
m = fx.ragged_mask(torch.tensor([20, 13, 7, 1, 16, 20, 5, 12]), n)
lab = fx.random_labels(m, Cc, 0.2, 3)
E = fx.one_hot_edges(lab.long(), m, Cc)
model2 = GraphTransformer(5, fx.dims_ns(1, Cc, 3), fx.CONSTRUCT_DIMS["hidden_mlp_dims"], fx.CONSTRUCT_DIMS["hidden_dims"], fx.dims_ns(1, Cc, 0)).eval()
sd2 = fx.make_state_dict({k: tuple(v.shape) for k, v in model2.state_dict().items()}, 3)
model2.load_state_dict(sd2); model2 = model2.cuda()
def fwd():
    out = model2(model_input(m[..., None].float().cuda(), E.cuda(), torch.ones(B, 3).cuda() * 0.3, m.cuda()))
    torch.cuda.synchronize()
    return out.E.cpu().clone(), out.X.cpu().clone()
f0 = fwd()
for val in (float("nan"), 1e30):
    poison(val)
    model3 = GraphTransformer(5, fx.dims_ns(1, Cc, 3), fx.CONSTRUCT_DIMS["hidden_mlp_dims"], fx.CONSTRUCT_DIMS["hidden_dims"], fx.dims_ns(1, Cc, 0)).eval()
    model3.load_state_dict(sd2); model2 = model3.cuda()
    f1 = fwd()
    print("forward after poison", val, "E equal", torch.equal(f0[0], f1[0]), "X equal", torch.equal(f0[1], f1[1]), "nan", bool(torch.isnan(f1[0]).any()))
