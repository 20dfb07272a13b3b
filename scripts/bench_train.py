#!/usr/bin/env python3
"""Side measurement: the training step (apply_noise + extra features + train-mode forward with dropout + TrainLoss + backward
+ gradient all-reduce + AdamW) at BASELINE.json configs[2]'s shape (ATM: B=4 graphs of up to 256 nodes per GPU, vessel dims,
5 layers, C=5).  Not the bench.py contract line.  Prints one JSON line on rank 0.

``--gpus N``: data-parallel replicas, one process per GPU (the reference: Lightning DDP, src/ConStruct/main.py:62-70,99-102),
every rank its own synthetic batch, gradients averaged by ``training.GradientBuckets`` (one bucketed RCCL all_reduce over xGMI,
started from the autograd hooks), timing bracketed by barriers, MAX over ranks, value = graphs of all ranks per second.  Run by
hand the script starts its own N ranks with bench.py's launcher (children created before any GPU call); under
torch.distributed.run it uses the ranks it is given.  ``--stub-gloo``: CPU rehearsal of launcher / all-reduce / timing on the
gloo backend with a small stand-in network (tests; not a measurement)."""
import argparse, json, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

DIMS = dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
            hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)


def ns(X, E, y, charges=0):
    return SimpleNamespace(X=X, E=E, y=y, charges=charges)


def synthetic_batch(a, rank, C, dev):
    """Random trees (vessel-like) of 3n/4 .. n nodes, seeded per rank: every rank trains on its own graphs."""
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    g = torch.Generator().manual_seed(7 + 1000 * rank)
    nn_ = torch.randint(3 * a.n // 4, a.n + 1, (a.batch,), generator=g)
    nn_[0] = a.n
    m = torch.arange(a.n)[None] < nn_[:, None]
    lab = torch.zeros(a.batch, a.n, a.n, dtype=torch.long)
    for b in range(a.batch):
        nb = int(nn_[b])
        idx = torch.arange(1, nb)
        par = (torch.rand(nb - 1, generator=g) * idx).long()
        cls = torch.randint(1, C, (nb - 1,), generator=g)
        lab[b, idx, par] = cls
        lab[b, par, idx] = cls
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(a.n, dtype=torch.bool)[None]
    E0 = torch.nn.functional.one_hot(lab, C).float() * live[..., None]
    dense = PlaceHolder(X=m[..., None].float().to(dev), charges=torch.zeros(a.batch, a.n, 0, device=dev), E=E0.to(dev),
                        y=torch.zeros(a.batch, 0, device=dev), node_mask=m.to(dev), pos=None)
    return dense, lab.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4, help="graphs per GPU per step")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--stub-gloo", action="store_true")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:          # before anything touches the GPU
        import bench
        raise SystemExit(bench.launch_ranks(a.gpus, sys.argv[1:], script=__file__))
    from semantically_consistent_graph_generation_amd import sharding
    from semantically_consistent_graph_generation_amd.training import GradientBuckets
    rank, local_rank, world = sharding.env_rank_world()
    if world != a.gpus:
        raise SystemExit(f"bench_train.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    dist = None
    if a.stub_gloo:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench_train.py needs an MI355X: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world) if a.stub_gloo else dist.init_process_group("nccl", device_id=dev)

    def sync():
        if not a.stub_gloo:
            torch.cuda.synchronize()

    C = 5
    torch.manual_seed(0)                                       # the same initial weights on every rank
    if a.stub_gloo:
        model = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
        gx = torch.Generator().manual_seed(50 + rank)
        xs, ys = torch.randn(a.batch, 8, generator=gx), torch.randn(a.batch, 4, generator=gx)

        def loss_of(it):
            time.sleep(0.005 * (rank + 1))
            return ((model(xs) - ys) ** 2).mean()
    else:
        from semantically_consistent_graph_generation_amd import noise_model as NM
        from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
        from semantically_consistent_graph_generation_amd.noise_model import Rng
        from semantically_consistent_graph_generation_amd.training import training_step
        from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
        cfg = SimpleNamespace(model=SimpleNamespace(nu=NU, transition="absorbing_edges", diffusion_steps=500, top_k=4, num_degree=10, max_degree=10,
                                                    cycle_features=True, num_eigenvectors=5, num_eigenvalues=9, eigenfeatures=True))
        ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=a.n))
        in_dims = ef.update_input_dims(ns(1, C, 1))
        model = GraphTransformer(5, in_dims, DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C, 0)).to(dev).train()
        nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
        dense, labd = synthetic_batch(a, rank, C, dev)

        def loss_of(it):
            return training_step(model, nm, ef, dense, labd, rng=Rng(seed=11 + rank, step=it), dropout_seed=1000 + it + 7919 * rank)[0]
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, amsgrad=True, weight_decay=1e-12)
    gb = GradientBuckets(list(model.parameters()), hook=True) if world > 1 else None
    losses = []

    def step(it):
        if gb is not None:
            gb.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        loss = loss_of(it)
        loss.backward()
        if gb is not None:
            gb.finish()                                        # averaged gradients; parameters unused on every rank keep grad None
        opt.step()
        return loss.detach()
    for it in range(a.warmup):
        step(it)
    sync()
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for it in range(a.steps):
        losses.append(step(a.warmup + it))
    host_issue = time.perf_counter() - t0                      # the host has queued every launch; the device may still run
    sync()
    if dist:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, device=dev)
    # replicas must stay identical: a checksum of the parameters, compared across ranks
    chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1).to(dev)
    same = True
    if dist:
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool((hi - lo).abs() <= 1e-9 * hi.abs().clamp(min=1.0))
    if rank == 0:
        ms = elapsed * 1e3 / a.steps
        out = {"workload": (f"training step, vessel dims, B={a.batch}/GPU, n<={a.n}, 5 layers, C=5, extra features on, dropout 0.1 (layers), "
                            f"AdamW, {'gradient all_reduce over ' + str(world) + ' ranks' if world > 1 else 'one GPU'}"),
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "graphs_per_s": world * a.batch / ms * 1e3,
               "host_issue_ms_per_step": host_issue * 1e3 / a.steps, "scaling": "weak", "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "replicas_identical": same,
               "data": "stub (launcher rehearsal, not a measurement)" if a.stub_gloo else "synthetic",
               "train_mode": os.environ.get("SCGG_TRAIN_MODE", "fused")}
        if not a.stub_gloo:
            out["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 2 ** 30
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
