#!/usr/bin/env python3
"""BASELINE.json configs[4] ("Stress"): B graphs of n = 1024 nodes, de = 128, dx = 256: the bf16 fused NodeEdgeBlock kernel
(scgg_edge_block_bf16) with its E-tensor HBM rate, and the cycle-feature adjacency products on the matrix pipe.  Prints one
JSON line.  Not the bench.py contract line (that is configs[1])."""
import argparse, json, os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--de", type=int, default=128)
    ap.add_argument("--dx", type=int, default=256, help="256 (stress / ConStruct) or 128 with --de 64 (vessel dims: the HBM-bound instance)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.stress import NodeEdgeBlockBF16
    from semantically_consistent_graph_generation_amd.transformer_model import NodeEdgeBlock
    dx, dy, n_head, B, n, de = a.dx, 64, 8 if a.dx == 256 else 4, a.batch, a.n, a.de
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
    # the pair kernel alone: events around the whole call minus the node-sized parts is not separable from Python, so time the
    # call and report it; rocprofv3 --kernel-trace gives the kernel's own duration (profiles/)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        op(X, E, y, m, lists=lists, out=out, attn=attn)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    # the pair kernel alone: events directly around its launch on the launch stream (torch's current stream), median of reps
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for k in range(a.reps):
        op.kernel_events = kev[k]
        op(X, E, y, m, lists=lists, out=out, attn=attn)
    op.kernel_events = None
    torch.cuda.synchronize()
    kms = sorted(s.elapsed_time(e) for s, e in kev)[a.reps // 2]
    pairs = B * n * n
    e_bytes = 2.0 * pairs * de * 2
    flops = 6.0 * pairs * de * dx
    PEAK_HBM, PEAK_BF16 = 8000.0, 2500.0                       # GB/s, TFLOP/s (MI355X_MICROARCH.md)
    # adjacency products (cycle features) at the same n
    cfg = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5, num_eigenvalues=9,
                                                eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=n))
    lab = (torch.rand(B, n, n, device=dev) < 2.0 / n)
    lab = (lab | lab.transpose(1, 2)).to(torch.int8)
    lab = torch.where(torch.eye(n, dtype=torch.bool, device=dev)[None], torch.full_like(lab, -1), lab)
    ef.adjacency_features(lab, m, 5)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        ef.adjacency_features(lab, m, 5)
    e1.record()
    torch.cuda.synchronize()
    adj_ms = e0.elapsed_time(e1) / 5
    np_ = (n + 63) // 64 * 64
    gbps, tflops = e_bytes / kms / 1e6, flops / kms / 1e9
    print(json.dumps({"workload": f"stress: B={B}, n={n}, de={de}, dx={dx}, bf16 NodeEdgeBlock + A^k", "block_call_ms": ms,
                      "kernel": f"scgg::edge_block_bf16_kernel<{dx // 32},{de // 32},12>", "kernel_ms": kms,
                      "roofline": {"bound": "hbm" if flops / e_bytes < PEAK_BF16 * 1e3 / PEAK_HBM else "mfma",
                                   "algorithmic_bytes_per_launch": e_bytes, "algorithmic_flops_per_launch": flops,
                                   "flop_per_byte": flops / e_bytes, "ridge_flop_per_byte": PEAK_BF16 * 1e3 / PEAK_HBM,
                                   "hbm": {"achieved": gbps, "peak": PEAK_HBM, "unit": "GB/s", "frac": gbps / PEAK_HBM},
                                   "mfma": {"achieved": tflops, "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": tflops / PEAK_BF16},
                                   "traffic": None, "traffic_note": "HBM bytes from PMC passes: profiles/ (when collected)"},
                      "E_stream_bytes": e_bytes, "E_stream_GBps_over_call": e_bytes / ms / 1e6, "hbm_frac_of_8TBps": e_bytes / ms / 1e6 / 8000,
                      "block_TFLOPs_over_call": flops / ms / 1e9, "adjacency_pass_ms": adj_ms,
                      "adjacency_products_TFLOPs_over_pass": 10 * 2.0 * np_ ** 3 * B / adj_ms / 1e9}))


if __name__ == "__main__":
    main()
