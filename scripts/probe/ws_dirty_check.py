"""Denoiser forward with its workspace pre-filled with different byte patterns: the outputs must not depend on it."""
import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.transformer_model import model_input, GraphTransformer
C = 6
n_nodes = [9, 4, 7, 1]
m = fx.ragged_mask(n_nodes); B, n = m.shape
E0 = fx.one_hot_edges(torch.zeros(B, n, n, dtype=torch.long), m, C)
inp = lambda: model_input(m[..., None].float().cuda(), E0.cuda(), torch.full((B, 1), 1.0).cuda(), m.cuda())
outs = {}
for byte in (0, 1, 0x3f, 0x7f, 255):
    model, sd = tiny_model(C, C, 1, 1)
    model._ensure_engine(torch.device("cuda:0"))
    model._workspace(B, n, torch.device("cuda:0"))
    for k, ws in model._ws.items():
        ws.fill_(byte)
    o = model(inp())
    torch.cuda.synchronize()
    outs[byte] = (o.E.cpu().clone(), o.X.cpu().clone())
    print("workspace filled with", byte, ": E equal to the zero-filled run", torch.equal(outs[0][0], outs[byte][0]), float((outs[0][0] - outs[byte][0]).abs().max()),
          "X equal", torch.equal(outs[0][1], outs[byte][1]), "nan", bool(torch.isnan(o.E).any()))
    o2 = model(inp()); torch.cuda.synchronize()
    print("    second call equal to the first", torch.equal(o2.E.cpu(), outs[byte][0]))
