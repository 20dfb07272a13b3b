"""Multi-GPU layout of the sampling path: one process per GPU, graphs sharded, no collective on the data path.

Reverse chains of different graphs never interact (reference diffusion_model_discrete_3d.py:583-684 loops over a batch
whose graphs are independent; the reference shards the list of graphs over Lightning DDP ranks, :560-581), so a rank
samples a contiguous slice of the global list of graphs and the only communication is the final gather of the sampled
labels plus the max-over-ranks of the wall time.  Every graph keeps its GLOBAL id as the Philox counter word
(csrc/rng.h), so the sampled graphs do not depend on the number of ranks.

Works with any initialised ``torch.distributed`` backend: ``nccl`` (= RCCL over xGMI) on the GPU box, ``gloo`` in the
CPU tests.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    start: int          # first global graph id of this rank's slice == graph_id_base for the Philox counters
    stop: int

    @property
    def count(self) -> int:
        return self.stop - self.start


def env_rank_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process when absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(n_graphs: int, rank: int, world: int) -> Shard:
    """Contiguous, balanced slices: the first ``n_graphs % world`` ranks hold one extra graph."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_graphs < 0:
        raise ValueError("n_graphs must be >= 0")
    base, extra = divmod(n_graphs, world)
    start = rank * base + min(rank, extra)
    return Shard(rank, world, start, start + base + (1 if rank < extra else 0))


def shard_node_counts(n_nodes_all: torch.Tensor, rank: int, world: int):
    """This rank's slice of the global list of graph sizes and its Shard."""
    sh = shard_range(int(n_nodes_all.shape[0]), rank, world)
    return n_nodes_all[sh.start:sh.stop], sh


def max_over_ranks(seconds: float, device=None) -> float:
    """The job's wall time: MAX over ranks (bench.py contract).  Identity without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_labels(labels: torch.Tensor, shard: Shard, n_graphs: int, dst: int = 0):
    """Final gather of the sampled edge labels [count, n, n] (any integer dtype) onto rank ``dst``: returns the
    [n_graphs, n, n] tensor in global graph order there, None elsewhere.  Slices may differ by one graph, so every
    rank pads to the largest slice and ``dst`` trims."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return labels
    world = dist.get_world_size()
    biggest = shard_range(n_graphs, 0, world).count
    padded = labels.new_zeros((biggest,) + tuple(labels.shape[1:]))
    padded[: shard.count] = labels
    bufs = [torch.empty_like(padded) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(padded, bufs, dst=dst)
    if dist.get_rank() != dst:
        return None
    return torch.cat([bufs[r][: shard_range(n_graphs, r, world).count] for r in range(world)], dim=0)
