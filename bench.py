"""Benchmark of the north-star path on MI355X: sampled graphs/sec for the full T-step reverse diffusion.

A "step" is one complete reverse chain (T = 500 denoiser forwards + extra features + reverse steps) over one batch of
B = 512 synthetic CoW-shaped graphs per GPU (BASELINE.json configs[1]: ConStruct dims, extra features on, n <= 32,
projector None).  N > 1 shards independent graph batches: every rank samples its own B graphs (weak scaling; rank r owns
the global graph ids [r B, (r+1) B), which are also the Philox counter words, so a graph does not depend on the number
of ranks), no collective on the data path; the timed region is bracketed by barrier + synchronize and the MAX over
ranks is reported; after it the int8 labels are gathered on rank 0 (the only collective besides the barriers).
Launch: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one rank;
run directly with --gpus N > 1 it starts N fresh child processes of itself, one per device, BEFORE anything here
touches the GPU (a process that has initialised the GPU is never re-executed), rendezvous on 127.0.0.1.

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
                sample (B = 32 graphs = BASELINE configs[0]: one denoiser forward + one reverse step, median of 10 after
                2 warm-ups), extrapolated to graphs/s for the full chain; a 1-thread figure beside it; and the same
                configs[0] step on the GPU (config1_gpu_ms).
"""
import argparse
import json
import os
import socket
import subprocess
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


def node_counts(B, rank, world=1):
    """CoW-shaped sizes n_i ~ U[12,32] of the job's world * B graphs, drawn once for the global list (so a graph's size
    does not depend on the number of ranks; the first B are the N = 1 list); every rank's first graph has n = 32."""
    g = torch.Generator().manual_seed(1234)
    n = torch.randint(12, N_MAX + 1, (world * B,), generator=g)
    n[::B] = N_MAX
    return n[rank * B:(rank + 1) * B].clone()


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


def cpu_baseline(T, timed=10, warm=2, B=32):
    """The oracle's reverse step (extra features + denoiser + posterior + sampling) on the host cores: BASELINE configs[0]
    (B = 32, n <= 32, one forward + one reverse step), median of ``timed`` after ``warm`` warm-ups, all threads; the same
    with one thread (3 steps) beside it."""
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
    X = m[..., None].float()

    def run(n_steps, skip):
        lab = torch.where(live, 0, -1)
        times = []
        with torch.no_grad():
            for k in range(n_steps + skip):
                s = T - 1 - k
                t0 = time.perf_counter()
                E = fx.one_hot_edges(lab.clamp(min=0), m, C_E)
                xe, ee, ye = XF.extra_features(X, E, m, N_MAX)
                y = torch.cat([ye, torch.full((B, 1), (s + 1) / T)], dim=1)
                _, _, logits, _ = D.graph_transformer_forward(sd, dd, torch.cat([X, xe], -1), torch.cat([E, ee], -1), y, m)
                u0, uk = N.philox_edge_uniforms(0, s, np.arange(B), n, TOP_K)
                lab = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), s + 1),
                                     torch.full((B, 1), s), np.zeros((B, n), np.float32), u0, uk)["labels_E"]
                if k >= skip:
                    times.append(time.perf_counter() - t0)
        return sorted(times)[len(times) // 2]
    threads = torch.get_num_threads()
    step = run(timed, warm)
    torch.set_num_threads(1)
    try:
        step1 = run(3, 1)
    finally:
        torch.set_num_threads(threads)
    return {"value": B / (T * step), "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"oracle (eager torch fp32) reverse step at BASELINE configs[0] (B={B} CoW-shaped graphs, one forward + "
                      f"one reverse step), median of {timed} after {warm} warm-ups ({step * 1e3:.0f} ms/step) extrapolated "
                      f"to the {T}-step chain",
            "config1_cpu_ms": step * 1e3, "value_1thread": B / (T * step1), "config1_cpu_ms_1thread": step1 * 1e3}


def config1_gpu_ms(sampler, B=32, timed=10, warm=2):
    """BASELINE configs[0] on the GPU: one denoiser forward + one reverse step at B = 32 (median of 10 after 2 warm-ups)."""
    nn_ = node_counts(B, 0)
    times = []
    for k in range(timed + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sampler.sample_batch(nn_, seed=5, n_max=N_MAX, steps=1, return_state=True)
        torch.cuda.synchronize()
        if k >= warm:
            times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2] * 1e3


# ------------------------------------------------------------------------------------------------------ launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, argv, script=None):
    """``python bench.py --gpus N`` without torch.distributed.run: start N fresh children of this script, one per device
    (RANK = LOCAL_RANK = r, rendezvous on 127.0.0.1), and exit with the worst of their codes.  Runs before anything in
    this process has touched the GPU; rank 0's stdout (the JSON line) is this process's stdout."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def stub_chain(graph_ids, n, seed):
    """CPU stand-in for a sampled state (launcher / gather tests on gloo): int8 labels that depend only on (seed, global
    graph id), symmetric, -1 on the diagonal."""
    g = torch.as_tensor(list(graph_ids), dtype=torch.long)[:, None, None]
    i = torch.arange(n)[None, :, None]
    j = torch.arange(n)[None, None, :]
    lab = ((g * 131 + torch.minimum(i, j) * 17 + torch.maximum(i, j) * 29 + seed) % 14).to(torch.int8)
    return torch.where(i == j, torch.full_like(lab, -1), lab)


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
    ap.add_argument("--stub-gloo", action="store_true", help="CPU rehearsal of the launcher / timing / gather logic: gloo "
                                                             "backend, a stub chain instead of the sampler (tests only; the "
                                                             "printed line is marked data=stub and is not a measurement)")
    ap.add_argument("--dump-labels", default=None, help="rank 0 saves the gathered labels of the last chain here (tests)")
    a = ap.parse_args()

    # ---- N > 1 without torch.distributed.run: this process only launches the ranks (before any GPU call)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))

    from semantically_consistent_graph_generation_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         f"(or without torch.distributed.run: bench.py starts the ranks itself)")
    dist = None
    if a.stub_gloo:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if a.stub_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev)

    def sync():
        if not a.stub_gloo:
            torch.cuda.synchronize()

    # weak scaling: the job samples world * batch graphs; rank r owns the global graph ids [r*batch, (r+1)*batch)
    n_total = world * a.batch
    shard = sharding.shard_range(n_total, rank, world)
    n_nodes = node_counts(a.batch, rank, world)
    valid_pairs = int((n_nodes.long() ** 2).sum())
    hd = DIMS["hidden_dims"]
    flops_per_launch = valid_pairs * (6 * hd["de"] * hd["dx"] + 4 * hd["de"] * hd["dim_ffE"])
    sampler = model = None
    if not a.stub_gloo:
        sampler, model = build_sampler(a.T, dev)

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
        if a.stub_gloo:
            time.sleep(0.01 * (rank + 1))
            return [(stub_chain(range(shard.start, shard.stop), N_MAX, 1000 + k),)]
        reqs = [dict(n_nodes=n_nodes, seed=1000 + k * a.pipeline + p, graph_id_base=shard.start, n_max=N_MAX, return_state=True,
                     projector=projector_factory(1000 + k * a.pipeline + p)) for p in range(a.pipeline)]
        return sampler.sample_batches(reqs)

    for k in range(a.warmup):
        chain(k)
    if a.warmup == 0 and not a.stub_gloo:
        sampler.sample_batch(n_nodes, seed=1, n_max=N_MAX, steps=1, return_state=True)   # builds the engine
    sync()
    if model is not None:
        model.profile_edge_layer(16, 4096)
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(a.steps):
        labels = chain(a.warmup + k)[-1][0]
    sync()
    if dist:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, device=dev)
    n_launch, ms = model.read_edge_layer_profile() if model is not None else (0, 0.0)
    # ---- the job's result: the labels of the last chain of every rank, in global graph order, on rank 0 (outside the
    #      timed region; the only data collective of the sampling path)
    t1 = time.perf_counter()
    full = sharding.gather_labels(labels, shard, n_total, dst=0)
    sync()
    gather_ms = (time.perf_counter() - t1) * 1e3

    if rank == 0:
        assert full.shape[0] == n_total, (full.shape, n_total)
        edges = int((full > 0).sum()) // 2
        checksum = int(full.to(torch.int64).sum()) & 0xFFFFFFFF
        if a.dump_labels:
            torch.save(full.cpu(), a.dump_labels)
        kernel_ms = ms / max(n_launch, 1)
        algorithmic = flops_per_launch / (kernel_ms * 1e-3) / 1e12 if n_launch else 0.0
        achieved = X3_PRODUCTS * algorithmic
        out = {
            "metric": "sampled graphs/sec (full T-step reverse diffusion)", "value": world * a.batch * a.pipeline * a.steps / elapsed,
            "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "stub (launcher rehearsal, not a measurement)" if a.stub_gloo else "synthetic",
            "config": {"workload": f"CoW-shaped full {a.T}-step reverse sampling, B={a.batch}/GPU, n<=32 (n_i~U[12,32]), "
                                   "ConStruct dims (dx256 de64 dy64 h8, 5 layers), extra features on, C=14, top_k=4, "
                                   f"absorbing_edges, projector {a.projector}"
                                   + (" (BASELINE configs[1])" if not a.projector and a.pipeline == 1 else
                                      f", {a.pipeline} batches in turn per step"),
                       "graphs_per_gpu": a.batch, "diffusion_steps": a.T, "valid_pairs_per_forward": valid_pairs,
                       "sampled_edges_last_chain": edges, "graphs_gathered": int(full.shape[0]),
                       "gathered_labels_checksum": checksum, "gather_ms": gather_ms},
        }
        if not a.stub_gloo:
            out["roofline"] = {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_MFMA_TFLOPS,
                "frac_note": "executed bf16 flops (6 partial products per fp32 product) / dense bf16 MFMA peak; the same number "
                             "as algorithmic flops / time / (peak / 6 = 417 TFLOP/s)",
                "algorithmic_f32_TFLOPs": algorithmic,
                "algorithmic_over_x3_roof_417TF": algorithmic / (PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS),
                "algorithmic_over_f32_mfma_peak": algorithmic / PEAK_F32_MFMA_TFLOPS,
                "frac_algorithmic_f32_peak": algorithmic / PEAK_F32_MFMA_TFLOPS,     # SURVEY 8(d)'s recipe read literally: algorithmic
                # fp32 flops / time / the dtype's own dense MFMA peak (157.3 TFLOP/s); `frac` above prices the executed bf16 products
                "traffic": measured_traffic() if a.batch == 512 else None,
                "algorithmic_bytes_per_launch": 3 * valid_pairs * hd["de"] * 4,
                "kernel": "scgg::edge_unit_layer_x3_kernel<8,2,4,12> (fused XEy layer, edge path; fp32 results from 6 bf16 "
                          "MFMA 16x16x32 partial products per product, fp32 accumulate)",
                "avg_launch_ms": kernel_ms, "launches_timed": n_launch,
                "algorithmic_flops_per_launch": flops_per_launch,
                "edge_stream_GBps": 2 * valid_pairs * hd["de"] * 4 / (kernel_ms * 1e-3) / 1e9 if n_launch else 0.0}
        if not a.no_cpu_baseline and world == 1 and not a.stub_gloo:
            out["cpu_baseline"] = cpu_baseline(a.T)
            out["cpu_baseline"]["config1_gpu_ms"] = config1_gpu_ms(sampler)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
