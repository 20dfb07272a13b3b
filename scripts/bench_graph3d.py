#!/usr/bin/env python3
"""Side measurement of the position-aware denoiser (row f1) at the shipped vessel configuration
(src/configs/model/discrete_vessels.yaml: dx 128, de 64, 4 heads, 5 layers, no extra features) on a sampling-size batch:
B = 512 graphs of n ~ U[12, 32] nodes, CoW classes (C = 14).  Not the bench.py contract line.

  python scripts/bench_graph3d.py [--steps 20] [--chain]

Prints one JSON line: ms per forward (HIP events on the launch stream), the fused edge kernel's share as counted flops,
and with --chain ms per reverse step of the whole sampling step (one-hot, forward, posterior + sampling)."""
import argparse
import json
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

DIMS = dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
            hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)


def ns(X, E, y, charges=0):
    return SimpleNamespace(X=X, E=E, y=y, charges=charges)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--chain", action="store_true")
    a = ap.parse_args()
    from semantically_consistent_graph_generation_amd import noise_model as NM
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    C = 14
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = Graph3DTransformer(a.layers, ns(1, C, 1), DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C, 0)).eval().to(dev)
    g = torch.Generator().manual_seed(7)
    nn_ = torch.randint(max(1, 3 * a.n // 8), a.n + 1, (a.batch,), generator=g)
    mask = (torch.arange(a.n)[None] < nn_[:, None]).to(dev)
    E = torch.zeros(a.batch, a.n, a.n, C, device=dev)
    E[..., 0] = (mask[:, :, None] & mask[:, None, :] & ~torch.eye(a.n, dtype=torch.bool, device=dev)[None]).float()
    inp = PlaceHolder(X=mask[..., None].float(), E=E, y=torch.full((a.batch, 1), 0.5, device=dev), pos=torch.randn(a.batch, a.n, 3, device=dev),
                      node_mask=mask, radius=None)
    for _ in range(3):
        model(inp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        model(inp)
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / a.steps
    pairs = int((nn_.double() ** 2).sum())
    de, ffe, dx, H = 64, 128, 128, 4
    flop_pair = 2 * (4 * de * de + 2 * de * ffe + 2 * H * de + H * dx + H * dx)       # Linears + logits + weighted V
    out = {"workload": f"Graph3DTransformer, vessel dims, {a.layers} layers, B={a.batch}, n~U[{max(1, 3 * a.n // 8)},{a.n}], C=14",
           "ms_per_forward": fwd_ms, "graphs_per_s_forward_only": a.batch / fwd_ms * 1e3, "valid_pairs": pairs,
           "edge_flop_per_pair_per_layer": flop_pair, "edge_gflop_per_forward": pairs * flop_pair * a.layers / 1e9}
    if a.chain:
        cfg = SimpleNamespace(model=SimpleNamespace(nu=NU, transition="absorbing_edges", diffusion_steps=500, top_k=4))
        nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
        sampler = DiscreteSampler(model, nm)
        sampler.sample_batch(nn_, seed=1, n_max=a.n, steps=2, return_state=True)
        torch.cuda.synchronize()
        e0.record()
        sampler.sample_batch(nn_, seed=1, n_max=a.n, steps=a.steps, return_state=True)
        e1.record()
        torch.cuda.synchronize()
        out["ms_per_reverse_step"] = e0.elapsed_time(e1) / a.steps
        out["graphs_per_s_500_step_chain"] = a.batch / (out["ms_per_reverse_step"] * 500) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
