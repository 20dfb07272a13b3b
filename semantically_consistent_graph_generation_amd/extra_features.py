"""Drop-in for ``src/ConStruct/diffusion/extra_features.py`` of the reference: ``ExtraFeatures(cfg, dataset_infos)``,
``update_input_dims`` and ``__call__(noisy_data) -> PlaceHolder(X, E, y)``; ``DummyExtraFeatures``.

The O(n^2)/O(n^3) part (adjacency powers, cycle counts, reachability, Adamic-Adar index, histograms, Laplacian)
runs in libscgg.so (``scgg_adjacency_features``) on the int8 edge state.  The eigendecomposition the reference
delegates to ``torch.linalg.eigh`` and the feature extraction around it run in ``scgg_eigen_features`` (batched Jacobi,
one wave per graph): rocSOLVER's eigh costs 50 ms per step at B=512, n=32 (profiles/r01_b).  No host syncs (the
reference's Python ``max()`` over device tensors, :149/:178, disappears: out-of-range columns are the padding value).

Beyond 32 nodes (vessel graphs, n = 256 / 1024) the adjacency products run as batched fp32 MFMA GEMMs
(``scgg_adjacency_features_tiled``, workspace owned here), and the eigendecomposition is the library call the reference
makes (``torch.linalg.eigh`` = rocSOLVER on the device) followed by the reference's own O(B n) tensor bookkeeping.
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
        self.lds_limit = 32                        # largest n of the one-workgroup kernels (adjacency and Jacobi eigen)
        self.tridiag_limit = 256                   # largest n of the tridiagonalisation kernel (packed matrix in LDS); beyond: rocSOLVER
        self._ws = {}                              # device -> workspace tensor of the tiled adjacency pass

    def _workspace(self, B, n, dev):
        need = int(_lib.lib().scgg_adjacency_features_workspace_bytes(B, n))
        ws = self._ws.get(dev)
        if ws is None or ws.numel() < need:
            ws = self._ws[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws, need

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
    def adjacency_features(self, labels, node_mask, C_e, e_dst=None, e_off=0, want_onehot=False, want_lap=True,
                           x_dst=None, x_off=0, y_dst=None, y_off=0):
        """Device call: -> (x, y, e_dst, lap or None).  The three feature groups are written at a channel offset into
        caller buffers when given (``x_dst`` [B,n,ldx], ``y_dst`` [B,ldy], ``e_dst`` [B,n,n,lde]; with ``want_onehot``
        the masked one-hot labels also go to ``e_dst[..., :C]``), else into fresh tensors x[B,n,3], y[B,4+nd+2+1+C],
        e[B,n,n,max_degree+1]."""
        B, n = node_mask.shape
        dev = node_mask.device
        if e_dst is None:
            e_dst = torch.empty(B, n, n, self.max_degree + 1, device=dev)
        if x_dst is None:
            x_dst = torch.empty(B, n, 3, device=dev)
        if y_dst is None:
            y_dst = torch.empty(B, 4 + self.num_degree + 2 + 1 + C_e, device=dev)
        lap = torch.empty(B, n, n, device=dev) if want_lap else None
        if n > self.lds_limit:
            ws, need = self._workspace(B, n, dev)
            _lib.check(_lib.lib().scgg_adjacency_features_tiled(
                _lib.ptr(labels), None, _lib.ptr(node_mask.view(torch.uint8)), B, n, C_e, 1, self.num_degree,
                self.max_degree, int(self.cycle_features), _lib.ptr(x_dst), x_dst.shape[-1], x_off, _lib.ptr(y_dst),
                y_dst.shape[-1], y_off, _lib.ptr(e_dst), e_dst.shape[-1], e_off, _lib.ptr(e_dst) if want_onehot else None,
                _lib.ptr(lap), _lib.ptr(ws), need, _lib.stream_ptr()), "adjacency_features_tiled")
            return x_dst, y_dst, e_dst, lap
        _lib.check(_lib.lib().scgg_adjacency_features(
            _lib.ptr(labels), None, _lib.ptr(node_mask.view(torch.uint8)), B, n, C_e, 1, self.num_degree, self.max_degree,
            int(self.cycle_features), _lib.ptr(x_dst), x_dst.shape[-1], x_off, _lib.ptr(y_dst), y_dst.shape[-1], y_off,
            _lib.ptr(e_dst), e_dst.shape[-1], e_off, _lib.ptr(e_dst) if want_onehot else None, _lib.ptr(lap),
            _lib.stream_ptr()), "adjacency_features")
        return x_dst, y_dst, e_dst, lap

    def laplacian(self, labels, node_mask):
        """The padded combinatorial Laplacian alone (reference compute_laplacian :121-127, padding :88-94), the same
        numbers as ``adjacency_features(..., want_lap=True)`` returns: lets the eigensolver start first."""
        B, n = node_mask.shape
        lap = torch.empty(B, n, n, device=node_mask.device)
        _lib.check(_lib.lib().scgg_laplacian(_lib.ptr(labels), _lib.ptr(node_mask.view(torch.uint8)), B, n, _lib.ptr(lap),
                                             _lib.stream_ptr()), "laplacian")
        return lap

    def eigen_features(self, lap, node_mask, evl_dst=None, evl_off=0, evc_dst=None, evc_off=0, want_raw=False):
        """EigenFeatures.compute_features after the Laplacian (reference :96-112, :138-195) on the device: batched
        Jacobi eigensolver + feature extraction in one kernel.  -> (evalue_feat, evector_feat[, w, V])."""
        B, n = node_mask.shape
        dev = node_mask.device
        if evl_dst is None:
            evl_dst = torch.empty(B, 1 + self.num_eigenvalues, device=dev)
        if evc_dst is None:
            evc_dst = torch.empty(B, n, 1 + self.num_eigenvectors, device=dev)
        if n > self.lds_limit and n <= self.tridiag_limit and self.num_eigenvectors <= 7 and self.num_eigenvalues <= 15:
            # 32 < n <= 256: tridiagonalisation in LDS, only the eigenvector columns the features read (eigen_tridiag.hip).
            # want_raw -> (w [B,n], the selected columns [B,n,1+num_eigenvectors]: column 0, then ncc .. ncc+k-1)
            w = torch.empty(B, n, device=dev) if want_raw else None
            V = torch.empty(B, n, 1 + self.num_eigenvectors, device=dev) if want_raw else None
            _lib.check(_lib.lib().scgg_eigen_features_tridiag(
                _lib.ptr(lap), _lib.ptr(node_mask.view(torch.uint8)), B, n, self.num_eigenvalues, self.num_eigenvectors,
                _lib.ptr(evl_dst), evl_dst.shape[-1], evl_off, _lib.ptr(evc_dst), evc_dst.shape[-1], evc_off, _lib.ptr(w),
                _lib.ptr(V), _lib.stream_ptr()), "eigen_features_tridiag")
            return (evl_dst, evc_dst, w, V) if want_raw else (evl_dst, evc_dst)
        if n > self.lds_limit:
            w, V = self._eigen_features_library(lap, node_mask, evl_dst, evl_off, evc_dst, evc_off)
            return (evl_dst, evc_dst, w, V) if want_raw else (evl_dst, evc_dst)
        w = torch.empty(B, n, device=dev) if want_raw else None
        V = torch.empty(B, n, n, device=dev) if want_raw else None
        _lib.check(_lib.lib().scgg_eigen_features(
            _lib.ptr(lap), _lib.ptr(node_mask.view(torch.uint8)), B, n, self.num_eigenvalues, self.num_eigenvectors,
            _lib.ptr(evl_dst), evl_dst.shape[-1], evl_off, _lib.ptr(evc_dst), evc_dst.shape[-1], evc_off, _lib.ptr(w),
            _lib.ptr(V), _lib.stream_ptr()), "eigen_features")
        return (evl_dst, evc_dst, w, V) if want_raw else (evl_dst, evc_dst)

    def _eigen_features_library(self, lap, node_mask, evl_dst, evl_off, evc_dst, evc_off):
        """n > 256: ``torch.linalg.eigh`` (rocSOLVER; the reference's own call, :96) and the reference's bookkeeping
        (:138-195) as device tensor ops without its host syncs: the gather windows are padded by their full width (value 2 /
        0 beyond column n, as :149-152 / :178-181 pad), and the mode of the rounded first eigenvector (:169-175) is taken
        over the valid nodes (padded slots get distinct values above every valid one: the limit of the reference's randn
        padding in which a padded value is never the mode, see oracle/extra_features.py)."""
        B, n = node_mask.shape
        dev = node_mask.device
        m = node_mask.to(lap.dtype)
        w, V = torch.linalg.eigh(lap)
        ev = w / m.sum(dim=1, keepdim=True)
        vec = V * m[:, :, None] * m[:, None, :]
        ncc = (ev < 1e-5).sum(dim=-1)
        ke, kv = self.num_eigenvalues, self.num_eigenvectors
        evp = torch.cat([ev, torch.full((B, ke), 2.0, device=dev)], dim=1)
        idx = torch.arange(ke, device=dev)[None] + ncc[:, None]
        evl_dst[:, evl_off] = ncc.to(ev.dtype)
        evl_dst[:, evl_off + 1: evl_off + 1 + ke] = torch.gather(evp, 1, idx)
        pad = 1.0e6 + torch.arange(n, device=dev, dtype=lap.dtype)
        v0 = torch.where(node_mask, torch.round(vec[:, :, 0], decimals=3), pad[None])
        mode = torch.mode(v0, dim=1).values
        evc_dst[:, :, evc_off] = ((v0 != mode[:, None]) & node_mask).to(lap.dtype)
        vp = torch.cat([vec, torch.zeros(B, n, kv, device=dev)], dim=2)
        idx = (torch.arange(kv, device=dev)[None, None] + ncc[:, None, None]).expand(-1, n, -1)
        evc_dst[:, :, evc_off + 1: evc_off + 1 + kv] = torch.gather(vp, 2, idx) * m[:, :, None]
        return w, V

    def __call__(self, noisy_data):
        """reference :60-77."""
        node_mask = noisy_data.node_mask.contiguous()
        B, n = node_mask.shape
        C_e = noisy_data.E.shape[-1]
        labels = NoiseModel._labels_of(noisy_data, C_e)
        ky = 4 + self.num_degree + 2 + 1 + C_e
        X = torch.empty(B, n, 3 + 1 + self.num_eigenvectors, device=node_mask.device)
        y = torch.empty(B, 1 + ky + 1 + self.num_eigenvalues, device=node_mask.device)
        _, _, e, lap = self.adjacency_features(labels, node_mask, C_e, x_dst=X, x_off=0, y_dst=y, y_off=1)
        self.eigen_features(lap, node_mask, evl_dst=y, evl_off=1 + ky, evc_dst=X, evc_off=3)
        y[:, 0:1] = node_mask.sum(dim=1, keepdim=True) / self.max_n_nodes
        return PlaceHolder(X=X, E=e, y=y)
