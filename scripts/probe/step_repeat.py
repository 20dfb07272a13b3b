"""Probe: one reverse step's inputs + forward on a fixed state, many times (structural features on the caller's stream, eigen
features on the engine's side stream, then the forward); counts results that differ from the first (must be 0).
  python scripts/probe/step_repeat.py [B=64] [reps=3000]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from oracle import fixtures as fx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda:0")
sampler, model = bench.build_sampler(500, dev)
mask = fx.ragged_mask(bench.node_counts(B, 0), 32)
lab = fx.random_labels(mask, 14, 0.08, 3)
live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(32, dtype=torch.bool)[None]
lab8 = torch.where(live, lab, torch.full_like(lab, -1)).to(torch.int8).to(dev)
md = mask.to(dev)
st = sampler._alloc_state(md)


def once():
    out = model(sampler._build_inputs(st, lab8, md, 250))
    return out.E.clone(), st["X_in"].clone(), st["y_in"].clone(), st["E_in"].clone()


ref = once()
bad = 0
for r in range(reps):
    if r % 3 == 0:
        junk = torch.randn(1 << (10 + r % 12), device="cuda")      # perturb timing
    cur = once()
    bad += int(not all(torch.equal(a, b) for a, b in zip(cur, ref)))
print(f"B={B}: {bad} of {reps} repeated steps differ from the first")
