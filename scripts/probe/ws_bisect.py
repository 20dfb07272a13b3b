"""Which bytes of the denoiser workspace are read before they are written: fill ranges with 0xFF and compare the logits."""
import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_sampling import *
from semantically_consistent_graph_generation_amd.transformer_model import model_input
C = 6
n_nodes = [9, 4, 7, 1]
m = fx.ragged_mask(n_nodes); B, n = m.shape
E0 = fx.one_hot_edges(torch.zeros(B, n, n, dtype=torch.long), m, C)
inp = lambda: model_input(m[..., None].float().cuda(), E0.cuda(), torch.full((B, 1), 1.0).cuda(), m.cuda())
model, sd = tiny_model(C, C, 1, 1)
dev = torch.device("cuda:0")
model._ensure_engine(dev); model._workspace(B, n, dev)
ws = next(iter(model._ws.values()))
ws.zero_(); ref = model(inp()).E.cpu().clone()
def differs(lo, hi):
    ws.zero_(); ws[lo:hi] = 255
    return not torch.equal(model(inp()).E.cpu(), ref)
N_ = ws.numel(); print("workspace bytes", N_, "offset of the aligned base", (-ws.data_ptr()) % 256)
# find all maximal sensitive 256-byte chunks (coarse: 4 KiB pages first)
page = 4096
bad_pages = [p for p in range(0, N_, page) if differs(p, min(p + page, N_))]
print("sensitive 4-KiB pages:", bad_pages[:40], "count", len(bad_pages))
for p in bad_pages[:4]:
    sub = [q for q in range(p, min(p + page, N_), 64) if differs(q, min(q + 64, N_))]
    print("  page", p, "sensitive 64-B chunks:", sub[:16], "count", len(sub))
