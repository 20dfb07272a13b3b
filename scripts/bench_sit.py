#!/usr/bin/env python3
"""Side measurement of the node-coordinate DDPM (row f3): SiT-T, 999 ancestral steps, B = 512 graphs of n ~ U[12,32] nodes
(the batch the edge chain then starts from).  Not the bench.py contract line.

  python scripts/bench_sit.py [--steps 200]      # steps < 999: time that many iterations and scale

Prints one JSON line: ms per ancestral step, seconds for the full 999-step chain, dense flops per step."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    from semantically_consistent_graph_generation_amd.node_coordinate_diffusion import SiT_models
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = SiT_models["SiT-T"](in_channels=3).to(dev)
    for p in model.parameters():
        torch.nn.init.normal_(p, std=0.05)
    g = torch.Generator().manual_seed(7)
    nn_ = torch.randint(max(1, 3 * a.n // 8), a.n + 1, (a.batch,), generator=g)
    mask = (torch.arange(a.n)[None] < nn_[:, None]).to(dev)
    xT = torch.randn(a.batch, a.n, 3, device=dev)
    model.sample(xT, mask, steps=3, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = model.sample(xT, mask, steps=a.steps, seed=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    rows, H = int(nn_.sum()), 192
    flops = rows * 6 * 2 * (H * 3 * H + H * H + 2 * H * 4 * H) + int((nn_.double() ** 2).sum()) * 6 * 4 * H
    print(json.dumps({"workload": f"SiT-T node-coordinate DDPM, B={a.batch}, n~U[{max(1, 3 * a.n // 8)},{a.n}]", "steps_timed": a.steps,
                      "ms_per_ancestral_step": ms, "s_per_999_step_chain": ms * 999 / 1e3, "valid_tokens": rows,
                      "gflop_per_step_valid_rows": flops / 1e9, "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
