"""Probe (needs a build with EXTRA=-DSCGG_STAMP_Y): cycle stamps of y_layer_kernel (update launches) in a warm forward of the
benchmark's shape.  stamps: 0 entry, 1 inputs + pooled statistics in LDS, 2 y_y|x_y|e_y, 3 y_out (two layers), 4 LN1,
5 lin_y1, 6 lin_y2, 7 LN2 + y stored, 8 FiLM vectors."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from oracle import fixtures as fx

dev = torch.device("cuda:0")
sampler, model = bench.build_sampler(500, dev)
B = 512
mask = fx.ragged_mask(bench.node_counts(B, 0), 32)
lab = fx.random_labels(mask, 14, 0.08, 3)
live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(32, dtype=torch.bool)[None]
lab8 = torch.where(live, lab, torch.full_like(lab, -1)).to(torch.int8).to(dev)
md = mask.to(dev)
st = sampler._alloc_state(md)
dll = ctypes.CDLL(os.path.join(R, "semantically_consistent_graph_generation_amd", "libscgg.so"))
for _ in range(4):
    out = model(sampler._build_inputs(st, lab8, md, 250))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 16))()
assert dll.scgg_debug_y_stamps(buf, 1) == 0
out = model(sampler._build_inputs(st, lab8, md, 250))
torch.cuda.synchronize()
assert dll.scgg_debug_y_stamps(buf, 0) == 0
for wg in range(4):
    s = [buf[wg * 16 + k] for k in range(9)]
    print(f"wg{wg}: " + " ".join(f"{s[i + 1] - s[i]:6d}" for i in range(8)) + f"  total {s[8] - s[0]}")
    p_ = [buf[wg * 16 + k] for k in (0, 9, 10, 11, 12)]
    print("      inside phase 0->1: input fill %d, mask compaction %d, Xtoy %d, Etoy %d" % tuple(p_[i + 1] - p_[i] for i in range(4)))
