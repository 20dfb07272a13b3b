"""Drop-in for ``src/ConStruct/diffusion/extra_features.py`` of the reference: ``ExtraFeatures(cfg, dataset_infos)``,
``update_input_dims`` and ``__call__(noisy_data) -> PlaceHolder(X, E, y)``; ``DummyExtraFeatures``.

The O(n^2)/O(n^3) part (adjacency powers, cycle counts, reachability, Adamic-Adar index, histograms, Laplacian)
runs in libscgg.so (``scgg_adjacency_features``) on the int8 edge state.  The eigendecomposition is the library call
the reference uses (``torch.linalg.eigh`` -> rocSOLVER); the O(n) gathers around it are torch ops on the device and
free of host syncs (the reference's Python ``max()`` over device tensors, :149/:178, is replaced by always padding
with k columns, which selects the same entries).
"""
from __future__ import annotations

import torch

from . import _lib
from .noise_model import NoiseModel
from .placeholder import PlaceHolder


class DummyExtraFeatures:
    """reference :17-25: zero-width features."""

    def __call__(self, noisy_data):
        dev = noisy_data.X.device
        return PlaceHolder(X=torch.zeros((*noisy_data.X.shape[:-1], 0), device=dev),
                           E=torch.zeros((*noisy_data.E.shape[:-1], 0), device=dev),
                           y=torch.zeros((noisy_data.y.shape[0], 0), device=dev))

    def update_input_dims(self, input_dims):
        return input_dims


def _get(cfg, name):
    m = cfg.model if not isinstance(cfg, dict) else cfg["model"]
    return m[name] if isinstance(m, dict) else getattr(m, name)


class ExtraFeatures:
    def __init__(self, cfg, dataset_infos):
        self.eigenfeatures = True                  # the reference's 1-tuple is always truthy (:30, quirk 7)
        self.max_n_nodes = dataset_infos.max_n_nodes
        self.num_degree = _get(cfg, "num_degree")
        self.max_degree = _get(cfg, "max_degree")
        self.cycle_features = bool(_get(cfg, "cycle_features"))
        self.num_eigenvectors = _get(cfg, "num_eigenvectors")
        self.num_eigenvalues = _get(cfg, "num_eigenvalues")

    def update_input_dims(self, input_dims):
        """reference :43-58 (mutates and returns input_dims)."""
        input_dims.y += self.num_eigenvalues + 1
        input_dims.y += self.num_degree + 2
        input_dims.y += input_dims.X
        input_dims.y += input_dims.E
        input_dims.X += self.num_eigenvectors + 1
        input_dims.X += 3
        input_dims.y += 5
        input_dims.E += self.max_degree + 1
        return input_dims

    # -------------------------------------------------------------------------------------------------------
    def adjacency_features(self, labels, node_mask, C_e, e_dst=None, e_off=0, want_onehot=False, want_lap=True):
        """Device call: -> (x[B,n,3], y[B,4+nd+2+1+C], e_dst, lap or None).  ``e_dst`` [B,n,n,ld] receives the edge
        features at channel offset ``e_off`` (and the one-hot labels at [0,C) when ``want_onehot``)."""
        B, n = node_mask.shape
        dev = node_mask.device
        kE = self.max_degree + 1
        if e_dst is None:
            e_dst = torch.empty(B, n, n, kE, device=dev)
        ld = e_dst.shape[-1]
        x = torch.empty(B, n, 3, device=dev)
        y = torch.empty(B, 4 + self.num_degree + 2 + 1 + C_e, device=dev)
        lap = torch.empty(B, n, n, device=dev) if want_lap else None
        _lib.check(_lib.lib().scgg_adjacency_features(
            _lib.ptr(labels), None, _lib.ptr(node_mask.view(torch.uint8)), B, n, C_e, 1, self.num_degree, self.max_degree,
            int(self.cycle_features), _lib.ptr(x), _lib.ptr(y), _lib.ptr(e_dst), ld, e_off,
            _lib.ptr(e_dst) if want_onehot else None, _lib.ptr(lap), _lib.stream_ptr()), "adjacency_features")
        return x, y, e_dst, lap

    def eigen_features(self, lap, node_mask):
        """EigenFeatures.compute_features after the Laplacian (reference :96-112, :138-195)."""
        B, n = node_mask.shape
        m = node_mask.float()
        ev, vec = torch.linalg.eigh(lap)
        ev = ev / m.sum(dim=1, keepdim=True)
        vec = vec * m[:, :, None] * m[:, None, :]
        ncc = (ev < 1e-5).sum(dim=-1)
        k = self.num_eigenvalues
        evp = torch.cat([ev, 2 * torch.ones(B, k, device=ev.device)], dim=1)
        first_ev = torch.gather(evp, 1, torch.arange(k, device=ev.device)[None] + ncc[:, None])
        evalue_feat = torch.cat([ncc[:, None].to(ev.dtype), first_ev], dim=1)
        # LCC indicator: nodes whose (rounded) first-eigenvector entry differs from the most common value.  Padded
        # entries get distinct large values so they can only win a tie among all-distinct valid entries, where the
        # smallest valid value wins (the reference randomises them with randn, :171-173).
        v0 = torch.round(vec[:, :, 0], decimals=3)
        v0 = torch.where(node_mask, v0, 1e6 + torch.arange(n, device=ev.device, dtype=v0.dtype)[None])
        mode = torch.mode(v0, dim=1).values
        not_lcc = ((v0 != mode[:, None]) & node_mask).float()
        k = self.num_eigenvectors
        vp = torch.cat([vec, torch.zeros(B, n, k, device=ev.device)], dim=2)
        idx = (torch.arange(k, device=ev.device)[None, None] + ncc[:, None, None]).expand(-1, n, -1)
        first_vec = torch.gather(vp, 2, idx) * m[:, :, None]
        return evalue_feat, torch.cat([not_lcc[..., None], first_vec], dim=-1)

    def __call__(self, noisy_data):
        """reference :60-77."""
        node_mask = noisy_data.node_mask.contiguous()
        C_e = noisy_data.E.shape[-1] if noisy_data.E is not None else noisy_data._labels_classes
        labels = NoiseModel._labels_of(noisy_data, C_e)
        x, y, e, lap = self.adjacency_features(labels, node_mask, C_e)
        nfrac = node_mask.sum(dim=1, keepdim=True) / self.max_n_nodes
        evl, evc = self.eigen_features(lap, node_mask)
        return PlaceHolder(X=torch.cat([x, evc], dim=-1), E=e, y=torch.cat([nfrac, y, evl], dim=1))
