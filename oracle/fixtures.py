"""Deterministic synthetic inputs / weights shared by the golden generator and the tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Large weight sets are never committed: they are
re-created from (key name, shape, seed) by ``make_state_dict`` on both sides.
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace

import numpy as np
import torch


def make_state_dict(shapes: dict, seed: int) -> dict:
    """name -> fp32 tensor; every tensor drawn from its own generator seeded by (seed, crc32(name)).

    2-D weights ~ U(-1,1)/sqrt(fan_in) * 1.5, LayerNorm weights ~ 1 + 0.2 U(-1,1), biases ~ 0.2 U(-1,1)."""
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
        u = torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1
        if len(shape) == 2:
            t = u * (1.5 / np.sqrt(shape[1]))
        elif "norm" in name and name.endswith("weight"):
            t = 1 + 0.2 * u
        else:
            t = 0.2 * u
        sd[name] = t.contiguous()
    return sd


def ragged_mask(n_nodes, n_max=None) -> torch.Tensor:
    n_nodes = torch.as_tensor(n_nodes)
    n_max = int(n_nodes.max()) if n_max is None else n_max
    return torch.arange(n_max)[None, :] < n_nodes[:, None]


def random_labels(node_mask: torch.Tensor, C: int, density: float, seed: int) -> torch.Tensor:
    """Symmetric int64 edge labels [B,n,n] in [0,C): ~density of the valid off-diagonal pairs carry an edge."""
    g = torch.Generator().manual_seed(seed)
    B, n = node_mask.shape
    has = torch.rand(B, n, n, generator=g) < density
    lab = torch.randint(1, C, (B, n, n), generator=g) * has
    lab = torch.triu(lab, diagonal=1)
    lab = lab + lab.transpose(1, 2)
    pm = node_mask[:, :, None] & node_mask[:, None, :]
    return lab * pm


def one_hot_edges(labels: torch.Tensor, node_mask: torch.Tensor, C: int) -> torch.Tensor:
    """Masked one-hot float E[B,n,n,C]: zero vector on the diagonal and on padded pairs (utils.py:154)."""
    n = labels.shape[1]
    pm = (node_mask[:, :, None] & node_mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    return torch.nn.functional.one_hot(labels, C).float() * pm[..., None]


def dims_ns(X, E, y, charges=0):
    return SimpleNamespace(X=X, E=E, y=y, charges=charges)


CONSTRUCT_DIMS = dict(  # src/configs/model/discrete.yaml:19-22
    hidden_mlp_dims={"X": 256, "E": 128, "y": 128},
    hidden_dims={"dx": 256, "de": 64, "dy": 64, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 128},
)
VESSEL_DIMS = dict(     # src/configs/model/discrete_vessels.yaml:21-25
    hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
    hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128},
)
TINY_DIMS = dict(
    hidden_mlp_dims={"X": 24, "E": 32, "y": 12},
    hidden_dims={"dx": 32, "de": 32, "dy": 16, "n_head": 4, "dim_ffX": 40, "dim_ffE": 64, "dim_ffy": 128},
)
