#!/usr/bin/env python3
"""Side measurements at the vessel (ATM) shapes of BASELINE.json configs[2] and configs[4]: not the bench.py contract line.

  python scripts/bench_vessel.py               # config 3: B=4 graphs of 256 nodes, vessel dims, 5 layers, C=5, extra features on
  python scripts/bench_vessel.py --n 1024 --batch 1 --steps 3     # config 5's size through the same path

Prints one JSON line: ms per reverse step of the whole chain step (features + denoiser + posterior + sampling), and the
tiled adjacency pass alone with the fp32 MFMA rate of its K products (2 n_pad^3 flops each, K = max_degree).
"""
import argparse
import json
import os
import sys
import time
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
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--adjacency-only", action="store_true", help="skip the chain (the denoiser's unit list stops at n = 512)")
    a = ap.parse_args()
    from semantically_consistent_graph_generation_amd import noise_model as NM
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    C = 5
    cfg = SimpleNamespace(model=SimpleNamespace(nu=NU, transition="absorbing_edges", diffusion_steps=a.T, top_k=4,
                                                num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5,
                                                num_eigenvalues=9, eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=a.n))
    in_dims = ef.update_input_dims(ns(1, C, 1))
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = GraphTransformer(5, in_dims, DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C, 0)).eval().to(dev)
    nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
    sampler = DiscreteSampler(model, nm, ef)
    g = torch.Generator().manual_seed(7)
    nn_ = torch.randint(3 * a.n // 4, a.n + 1, (a.batch,), generator=g)
    nn_[0] = a.n
    if a.adjacency_only:
        step_ms = None
        m0 = torch.arange(a.n)[None] < nn_[:, None]
        lab = torch.zeros(a.batch, a.n, a.n, dtype=torch.int8)
        for b in range(a.batch):                                                  # a random tree + a few extra edges per graph
            nb = int(nn_[b])
            par = (torch.rand(nb - 1, generator=g) * torch.arange(1, nb)).long()
            idx = torch.arange(1, nb)
            lab[b, idx, par] = 1
            lab[b, par, idx] = 1
        live = (m0[:, :, None] & m0[:, None, :]) & ~torch.eye(a.n, dtype=torch.bool)[None]
        lab = torch.where(live, lab, torch.full_like(lab, -1)).to(dev)
    else:
        sampler.sample_batch(nn_, seed=1, n_max=a.n, steps=2, return_state=True)      # warm-up (allocations, packing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lab, _, _, _ = sampler.sample_batch(nn_, seed=1, n_max=a.n, steps=a.steps, return_state=True)
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t0) * 1e3 / a.steps
    # the tiled adjacency pass alone on the sampled state
    mask = (torch.arange(a.n)[None] < nn_[:, None]).to(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ef.adjacency_features(lab, mask, C)
    torch.cuda.synchronize()
    reps = 10
    ev0.record()
    for _ in range(reps):
        ef.adjacency_features(lab, mask, C)
    ev1.record()
    torch.cuda.synchronize()
    adj_ms = ev0.elapsed_time(ev1) / reps
    np_ = (a.n + 63) // 64 * 64
    flops = 10 * 2.0 * np_ ** 3 * a.batch
    print(json.dumps({"workload": f"vessel dims, B={a.batch}, n={a.n}, 5 layers, C=5, extra features on", "ms_per_reverse_step": step_ms,
                      "steps": a.steps, "adjacency_pass_ms": adj_ms, "adjacency_products_tflops_if_all_time_were_gemm": flops / adj_ms / 1e9,
                      "edges_sampled": int((lab > 0).sum().item()) // 2}))


if __name__ == "__main__":
    main()
