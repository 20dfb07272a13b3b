"""Probe (needs a build with EXTRA=-DSCGG_STAMP_NODE): cycle stamps of node_layer_x3_kernel's phases in the first forward of the
benchmark's shape; prints, per sampled (workgroup, wave), the cycles between consecutive stamps.
stamps: 0 entry, 1 lists read + ring primed, 2 prologue done (barrier), 3 x_out done, 4 LN1 done, 5 linX1 done, 6 linX2 done,
8 LN2 done, 7 barrier, 9 q|k|v done."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd import _lib

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
for _ in range(4):                                   # warm caches; the stamps of the first launch after a reset are kept
    out = model(sampler._build_inputs(st, lab8, md, 250))
torch.cuda.synchronize()
assert dll.scgg_debug_node_stamps_reset() == 0
out = model(sampler._build_inputs(st, lab8, md, 250))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 2 * 16))()
assert dll.scgg_debug_node_stamps(buf) == 0
order = [0, 1, 2, 3, 4, 5, 6, 8, 7, 9]
for wg in range(4):
    for wv in range(2):
        s = [buf[(wg * 2 + wv) * 16 + k] for k in range(16)]
        d = [s[order[i + 1]] - s[order[i]] for i in range(len(order) - 1)]
        print(f"wg{wg} wave{'7' if wv else '0'}: " + " ".join(f"{x:6d}" for x in d) + f"  total {s[9] - s[0]}")
