"""Benchmark of the north-star path on MI355X: sampled graphs/sec for the full T-step reverse diffusion.

A "step" is one complete reverse chain (T = 500 denoiser forwards + extra features + reverse steps) over one batch of
B = 512 synthetic CoW-shaped graphs per GPU (BASELINE.json configs[1]: ConStruct dims, extra features on, n <= 32,
projector None).  N > 1 (launched by torch.distributed.run) shards independent graph batches: every rank samples its
own B graphs (weak scaling), no collective on the data path; the timed region is bracketed by barrier +
synchronize and the MAX over ranks is reported.

Also printed in the same JSON line:
  roofline      the fused edge-layer kernel (dominant), measured live with HIP events on the launch stream inside the
                timed region (every 16th forward).  The kernel evaluates every fp32 product as six bf16 x bf16 partial
                products on v_mfma_f32_16x16x32_bf16 (exact three-term split, fp32 accumulate: DESIGN.md section 4.1), so
                achieved = EXECUTED bf16 flops (6 x the algorithmic flops, valid pairs x (6 de dx + 4 de ffE)) per
                launch / average duration, against the dense bf16 MFMA peak (2500 TFLOP/s); the algorithmic fp32 rate
                and its ratio to the fp32 MFMA peak (157.3 TFLOP/s, the bound of the fp32-MFMA kernel this one
                replaced) are reported beside it.  traffic = HBM bytes per launch from the committed rocprofv3 PMC
                passes of this command.
  cpu_baseline  the CPU oracle (eager torch fp32 "port" of the reference algorithm) timed on this host for a bounded
                sample (B = 32 graphs, a few reverse steps), extrapolated to graphs/s for the full chain.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

DIMS = dict(hidden_mlp_dims={"X": 256, "E": 128, "y": 128},
            hidden_dims={"dx": 256, "de": 64, "dy": 64, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 128})
C_E, N_LAYERS, TOP_K, N_MAX = 14, 5, 4, 32
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA", dense
X3_PRODUCTS = 6                   # bf16 partial products per fp32 product in the x3 path


def measured_traffic():
    """HBM-side bytes of one edge-layer launch at this workload, from the committed rocprofv3 PMC passes (FETCH_SIZE
    doubled per the guide's gfx950 correction + WRITE_SIZE; profiles/README.md).  PMC counters cannot be collected
    from inside the timed process, so the number is read back from the profile summary; None when it is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "edge_layer_traffic.json")) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def ns(X, E, y, charges=0):
    return SimpleNamespace(X=X, E=E, y=y, charges=charges)


def node_counts(B, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    n = torch.randint(12, N_MAX + 1, (B,), generator=g)
    n[0] = N_MAX
    return n


def build_sampler(T, device):
    from semantically_consistent_graph_generation_amd import noise_model as NM
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    cfg = SimpleNamespace(model=SimpleNamespace(nu=NU, transition="absorbing_edges", diffusion_steps=T, top_k=TOP_K,
                                                num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5,
                                                num_eigenvalues=9, eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=N_MAX))
    in_dims = ef.update_input_dims(ns(1, C_E, 1))
    torch.manual_seed(0)
    model = GraphTransformer(N_LAYERS, in_dims, DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C_E, 0)).eval().to(device)
    nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C_E) / C_E, torch.zeros(0), 0)
    return DiscreteSampler(model, nm, ef), model


def cpu_baseline(T, budget_steps=8, B=32):
    """The oracle's reverse step (extra features + denoiser + posterior + sampling) on the host cores."""
    import numpy as np
    from oracle import denoiser as D, extra_features as XF, fixtures as fx, noise as N
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    torch.manual_seed(0)
    shapes = GraphTransformer(N_LAYERS, ns(10, 25, 43), DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C_E, 0)).state_dict()
    sd = {k: v.clone() for k, v in shapes.items()}
    m = fx.ragged_mask(node_counts(B, 0), N_MAX)
    n = N_MAX
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tr = N.Transition(N.Schedule(T, "absorbing_edges", NU), [1.0], np.ones(C_E, np.float32) / C_E)
    dd = D.DenoiserDims(N_LAYERS, 8, 1, C_E, 0)
    lab = torch.where(live, 0, -1)
    X = m[..., None].float()
    times = []
    with torch.no_grad():
        for k in range(budget_steps + 1):
            s = T - 1 - k
            t0 = time.perf_counter()
            E = fx.one_hot_edges(lab.clamp(min=0), m, C_E)
            xe, ee, ye = XF.extra_features(X, E, m, N_MAX)
            y = torch.cat([ye, torch.full((B, 1), (s + 1) / T)], dim=1)
            _, _, logits, _ = D.graph_transformer_forward(sd, dd, torch.cat([X, xe], -1), torch.cat([E, ee], -1), y, m)
            u0, uk = N.philox_edge_uniforms(0, s, np.arange(B), n, TOP_K)
            lab = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), s + 1),
                                 torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)["labels_E"]
            if k > 0:
                times.append(time.perf_counter() - t0)
    step = sorted(times)[len(times) // 2]
    return {"value": B / (T * step), "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (eager torch fp32) reverse step, B={B} CoW-shaped graphs, median of {budget_steps} steps "
                      f"({step * 1e3:.0f} ms/step) extrapolated to the {T}-step chain"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=500, help="diffusion steps per chain (BASELINE: 500)")
    ap.add_argument("--batch", type=int, default=512, help="graphs per GPU per chain (BASELINE: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--projector", default=None, help="cfg.model.rev_proj (cow_proj, cow_proj_vanilla): SURVEY config 4; "
                                                      "not the headline configuration")
    ap.add_argument("--pipeline", type=int, default=1, help="independent batches advanced in turn per step (sample_batches); "
                                                            "a step then samples pipeline * batch graphs per GPU")
    a = ap.parse_args()

    from semantically_consistent_graph_generation_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    sampler, model = build_sampler(a.T, dev)
    # weak scaling: the job samples world * batch graphs; rank r owns the global graph ids [r*batch, (r+1)*batch)
    shard = sharding.shard_range(world * a.batch, rank, world)
    n_nodes = node_counts(a.batch, rank)
    valid_pairs = int((n_nodes.long() ** 2).sum())
    hd = DIMS["hidden_dims"]
    flops_per_launch = valid_pairs * (6 * hd["de"] * hd["dx"] + 4 * hd["de"] * hd["dim_ffE"])

    def projector_factory(seed):
        if not a.projector:
            return None
        import numpy as np
        from semantically_consistent_graph_generation_amd.projector import projector_for
        cls, rs = projector_for(a.projector), np.random.RandomState(seed)      # one shuffle stream per chain

        def make(z_T, device):
            return cls(z_T, device, rng=rs)
        make.project_labels = True
        return make

    def chain(k):
        reqs = [dict(n_nodes=n_nodes, seed=1000 + k * a.pipeline + p, graph_id_base=shard.start, n_max=N_MAX, return_state=True,
                     projector=projector_factory(1000 + k * a.pipeline + p)) for p in range(a.pipeline)]
        return sampler.sample_batches(reqs)[-1]

    for k in range(a.warmup):
        chain(k)
    if a.warmup == 0:
        sampler.sample_batch(n_nodes, seed=1, n_max=N_MAX, steps=1, return_state=True)   # builds the engine
    torch.cuda.synchronize()
    model.profile_edge_layer(16, 4096)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        labels = chain(a.warmup + k)[0]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, device=dev)
    n_launch, ms = model.read_edge_layer_profile()
    edges = int((labels > 0).sum()) // 2

    if rank == 0:
        kernel_ms = ms / max(n_launch, 1)
        algorithmic = flops_per_launch / (kernel_ms * 1e-3) / 1e12 if n_launch else 0.0
        achieved = X3_PRODUCTS * algorithmic
        out = {
            "metric": "sampled graphs/sec (full T-step reverse diffusion)", "value": world * a.batch * a.pipeline * a.steps / elapsed,
            "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CoW-shaped full {a.T}-step reverse sampling, B={a.batch}/GPU, n<=32 (n_i~U[12,32]), "
                                   "ConStruct dims (dx256 de64 dy64 h8, 5 layers), extra features on, C=14, top_k=4, "
                                   f"absorbing_edges, projector {a.projector}"
                                   + (" (BASELINE configs[1])" if not a.projector and a.pipeline == 1 else
                                      f", {a.pipeline} batches in turn per step"),
                       "graphs_per_gpu": a.batch, "diffusion_steps": a.T, "valid_pairs_per_forward": valid_pairs,
                       "sampled_edges_last_chain": edges},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_MFMA_TFLOPS,
                         "algorithmic_f32_TFLOPs": algorithmic,
                         "algorithmic_over_f32_mfma_peak": algorithmic / PEAK_F32_MFMA_TFLOPS,
                         "traffic": measured_traffic() if a.batch == 512 else None,
                         "algorithmic_bytes_per_launch": 3 * valid_pairs * hd["de"] * 4,
                         "kernel": "scgg::edge_unit_layer_x3_kernel<8,2,4,12> (fused XEy layer, edge path; fp32 results from 6 bf16 "
                                   "MFMA 16x16x32 partial products per product, fp32 accumulate)",
                         "avg_launch_ms": kernel_ms, "launches_timed": n_launch,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "edge_stream_GBps": 2 * valid_pairs * hd["de"] * 4 / (kernel_ms * 1e-3) / 1e9 if n_launch else 0.0},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.T)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
