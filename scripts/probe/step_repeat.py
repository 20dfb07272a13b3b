"""Probe: the state route's input assembly (structural + eigen features, side stream) + denoiser forward, repeated on one
fixed state: every repetition must give the same bits."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from oracle import fixtures as fx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
sampler, model = bench.build_sampler(500, dev)
nn_ = bench.node_counts(B, 0)
mask = fx.ragged_mask(nn_, 32)
lab = fx.random_labels(mask, 14, 0.08, 3)
live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(32, dtype=torch.bool)[None]
lab8 = torch.where(live, lab, torch.full_like(lab, -1)).to(torch.int8).to(dev)
md = mask.to(dev)
st = sampler._alloc_state(md)
def once():
    inp = sampler._build_inputs(st, lab8, md, 250)
    out = model(inp)
    return out.E.clone(), st["X_in"].clone(), st["y_in"].clone(), st["E_in"].clone()
ref = once()
torch.cuda.synchronize()
bad = [0, 0, 0, 0]
for r in range(reps):
    cur = once()
    for k in range(4):
        if not torch.equal(cur[k], ref[k]):
            bad[k] += 1
            if bad[k] <= 3:
                print("rep", r, ["logits", "X_in", "y_in", "E_in"][k], "max diff", float((cur[k] - ref[k]).abs().max()), flush=True)
print("B", B, "mismatches logits/X_in/y_in/E_in:", bad, "of", reps)
