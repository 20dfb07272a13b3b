#!/usr/bin/env python3
"""Time edge_block_bf16_kernel alone with HIP events (timing / ablation builds via SCGG_LIB_PATH): B=1, n=1024, de=128."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantically_consistent_graph_generation_amd.stress import NodeEdgeBlockBF16
from semantically_consistent_graph_generation_amd.transformer_model import NodeEdgeBlock
dx, dy, n_head, B, n, de = 256, 64, 8, 1, 1024, int(os.environ.get("DE", "128"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = NodeEdgeBlock(dx, de, dy, n_head).to(dev)
op = NodeEdgeBlockBF16(blk)
m = torch.ones(B, n, dtype=torch.bool, device=dev)
X = torch.randn(B, n, dx, device=dev)
E = torch.randn(B, n, n, de, device=dev)
E = (0.5 * (E + E.transpose(1, 2))).to(torch.bfloat16).contiguous()
y = torch.randn(B, dy, device=dev)
lists = op.work_lists(m)
out = torch.zeros(B, n, n, de, dtype=torch.bfloat16, device=dev)
attn = torch.empty(B * n * ((n + 15) // 16), 3, dx, device=dev)
for _ in range(3):
    op(X, E, y, m, lists=lists, out=out, attn=attn)
torch.cuda.synchronize()
import time
t = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op(X, E, y, m, lists=lists, out=out, attn=attn)
    torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
print(json.dumps({"lib": os.environ.get("SCGG_LIB_PATH", "default"), "call_ms_min": min(t) * 1e3}))
