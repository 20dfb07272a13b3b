"""Drop-in for ``src/ConStruct/models/transformer_model.py`` of the reference: ``GraphTransformer`` with the same
constructor arguments, the same ``state_dict`` names/shapes (so a Lightning checkpoint's ``model.*`` keys load) and
the same ``forward(model_input) -> PlaceHolder`` contract (reference transformer_model.py:251-389).

Only the parameter containers live here.  The arithmetic of forward runs in libscgg.so (HIP, gfx950) through
``scgg_denoiser_forward``; there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib
from .placeholder import PlaceHolder


def _dim(d, key):
    return getattr(d, key) if not isinstance(d, dict) else d[key]


def _two_linears(d_in, d_hid, d_out):
    # Linear at positions 0 and 3 (the reference's Sequential has activation/dropout modules in between)
    return nn.Sequential(nn.Linear(d_in, d_hid), nn.Identity(), nn.Identity(), nn.Linear(d_hid, d_out))


class _Pool(nn.Module):  # parameter holder for Xtoy / Etoy (layers.py:5-43): Linear over [mean|min|max|var]
    def __init__(self, d, dy):
        super().__init__()
        self.lin = nn.Linear(4 * d, dy)


class NodeEdgeBlock(nn.Module):
    """Parameters of the reference NodeEdgeBlock (transformer_model.py:118-163)."""

    def __init__(self, dx, de, dy, n_head, predicts_y=True, **_):
        super().__init__()
        assert dx % n_head == 0, f"dx: {dx} -- nhead: {n_head}"
        self.dx, self.de, self.dy, self.n_head, self.df = dx, de, dy, n_head, dx // n_head
        self.predicts_y = predicts_y
        for name in ("q", "k", "v"):
            setattr(self, name, nn.Linear(dx, dx))
        self.e_add = nn.Linear(de, dx)
        self.e_mul = nn.Linear(de, dx)
        for name in ("y_e_mul", "y_e_add", "y_x_mul", "y_x_add"):
            setattr(self, name, nn.Linear(dy, dx))      # dx, not de (quirk 2)
        if predicts_y:
            self.y_y = nn.Linear(dy, dy)
            self.x_y = _Pool(dx, dy)
            self.e_y = _Pool(de, dy)
        self.x_out = nn.Linear(dx, dx)
        self.e_out = nn.Linear(dx, de)
        if predicts_y:
            self.y_out = nn.Sequential(nn.Linear(dy, dy), nn.Identity(), nn.Linear(dy, dy))


class XEyTransformerLayer(nn.Module):
    """Parameters of the reference XEyTransformerLayer (transformer_model.py:26-74)."""

    def __init__(self, dx, de, dy, n_head, dim_ffX=2048, dim_ffE=128, dim_ffy=2048, dropout=0.1,
                 layer_norm_eps=1e-5, predicts_y=True, **_):
        super().__init__()
        self.predicts_y = predicts_y
        self.dropout = dropout
        self.self_attn = NodeEdgeBlock(dx, de, dy, n_head, predicts_y=predicts_y)
        self.linX1, self.linX2 = nn.Linear(dx, dim_ffX), nn.Linear(dim_ffX, dx)
        self.normX1, self.normX2 = nn.LayerNorm(dx, eps=layer_norm_eps), nn.LayerNorm(dx, eps=layer_norm_eps)
        self.linE1, self.linE2 = nn.Linear(de, dim_ffE), nn.Linear(dim_ffE, de)
        self.normE1, self.normE2 = nn.LayerNorm(de, eps=layer_norm_eps), nn.LayerNorm(de, eps=layer_norm_eps)
        if predicts_y:
            self.lin_y1, self.lin_y2 = nn.Linear(dy, dim_ffy), nn.Linear(dim_ffy, dy)
            self.norm_y1, self.norm_y2 = nn.LayerNorm(dy, eps=layer_norm_eps), nn.LayerNorm(dy, eps=layer_norm_eps)
        self.dim_ffy = dim_ffy


class GraphTransformer(nn.Module):
    """GraphTransformer(n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, dropout, dropout_in_and_out)
    -- reference transformer_model.py:257-342.  ``dim_ffy`` of hidden_dims is ignored, as in the reference (:309-317,
    quirk 3): lin_y1 is dy -> 2048."""

    def __init__(self, n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, dropout=0.1,
                 dropout_in_and_out=False):
        super().__init__()
        self.n_layers = n_layers
        self.in_dim_X = _dim(input_dims, "X") + _dim(input_dims, "charges")
        self.in_dim_E = _dim(input_dims, "E")
        self.in_dim_y = _dim(input_dims, "y")
        self.out_dim_X = _dim(output_dims, "X")
        self.out_dim_E = _dim(output_dims, "E")
        self.out_dim_y = _dim(output_dims, "y")
        self.out_dim_charges = _dim(output_dims, "charges")
        self.predicts_final_y = self.out_dim_y > 0
        self.hidden_mlp_dims = dict(hidden_mlp_dims)
        self.hidden_dims = dict(hidden_dims)
        self.dropout = dropout
        self.dropout_in_and_out = dropout_in_and_out
        hm, hd = self.hidden_mlp_dims, self.hidden_dims

        self.mlp_in_X = _two_linears(self.in_dim_X, hm["X"], hd["dx"])
        self.mlp_in_E = _two_linears(self.in_dim_E, hm["E"], hd["de"])
        self.mlp_in_y = _two_linears(self.in_dim_y, hm["y"], hd["dy"])
        self.tf_layers = nn.ModuleList([
            XEyTransformerLayer(dx=hd["dx"], de=hd["de"], dy=hd["dy"], n_head=hd["n_head"], dim_ffX=hd["dim_ffX"],
                                dim_ffE=hd["dim_ffE"], dropout=dropout,
                                predicts_y=(l != n_layers - 1) or self.predicts_final_y)
            for l in range(n_layers)])
        self.mlp_out_X = _two_linears(hd["dx"], hm["X"], self.out_dim_X + self.out_dim_charges)
        self.mlp_out_E = _two_linears(hd["de"], hm["E"], self.out_dim_E)
        if self.predicts_final_y:
            self.mlp_out_y = _two_linears(hd["dy"], hm["y"], self.out_dim_y)

        self._engine = None
        self._engine_key = None
        self._ws = {}

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _config(self):
        hm, hd = self.hidden_mlp_dims, self.hidden_dims
        return _lib.DenoiserConfig(
            n_layers=self.n_layers, in_X=self.in_dim_X, in_E=self.in_dim_E, in_y=self.in_dim_y,
            out_X=self.out_dim_X + self.out_dim_charges, out_E=self.out_dim_E, out_y=self.out_dim_y,
            hid_X=hm["X"], hid_E=hm["E"], hid_y=hm["y"], dx=hd["dx"], de=hd["de"], dy=hd["dy"], n_head=hd["n_head"],
            dim_ffX=hd["dim_ffX"], dim_ffE=hd["dim_ffE"], dim_ffy=self.tf_layers[0].dim_ffy)

    def _release(self):
        if self._engine is not None:
            _lib.lib().scgg_denoiser_destroy(self._engine)
            self._engine = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_engine(self, device):
        L = _lib.lib()
        sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
        key = (str(device), tuple((k, v.data_ptr()) for k, v in sd.items()))
        versions = tuple(v._version for v in sd.values())
        if self._engine is None or key != self._engine_key:
            self._release()
            for k, v in sd.items():
                if v.device != device or v.dtype != torch.float32 or not v.is_contiguous():
                    raise _lib.ScggError(f"parameter {k} must be contiguous fp32 on {device} (got {v.dtype} on {v.device})")
            cfg = self._config()
            h = L.scgg_denoiser_create(C.byref(cfg))
            if not h:
                raise _lib.ScggError("scgg_denoiser_create: " + L.scgg_last_error().decode())
            self._engine = C.c_void_p(h)
            for k, v in sd.items():
                _lib.check(L.scgg_denoiser_set_param(self._engine, k.encode(), _lib.ptr(v.detach()), v.numel()), k)
            _lib.check(L.scgg_denoiser_finalize(self._engine, _lib.stream_ptr()), "finalize")
            self._engine_key, self._versions = key, versions
        elif versions != self._versions:      # weights changed in place (optimizer step / load_state_dict)
            _lib.check(L.scgg_denoiser_refresh(self._engine, _lib.stream_ptr()), "refresh")
            self._versions = versions
        return self._engine

    def _workspace(self, B, n, device):
        k = (B, n, str(device))
        if k not in self._ws:
            nbytes = _lib.lib().scgg_denoiser_workspace_bytes(self._engine, B, n)
            self._ws = {k: torch.empty(nbytes + 256, dtype=torch.uint8, device=device)}   # keep one shape only
        ws = self._ws[k]
        off = (-ws.data_ptr()) % 256
        return C.c_void_p(ws.data_ptr() + off), ws.numel() - 256

    def _inputs(self, model_input):
        X, E, y, m = model_input.X, model_input.E, model_input.y, model_input.node_mask
        if not X.is_cuda:
            raise _lib.ScggError("GraphTransformer.forward runs on the MI355X only; move the inputs to the GPU")
        B, n = X.shape[0], X.shape[1]
        if X.shape[2] != self.in_dim_X or E.shape[3] != self.in_dim_E or y.shape[1] != self.in_dim_y:
            raise _lib.ScggError(f"input widths {X.shape[2]},{E.shape[3]},{y.shape[1]} do not match the model "
                                 f"({self.in_dim_X},{self.in_dim_E},{self.in_dim_y})")
        return (X.detach().float().contiguous(), E.detach().float().contiguous(), y.detach().float().contiguous(),
                m.to(torch.bool).contiguous(), B, n)

    def side_stream(self, device):
        """The engine's second stream as a torch stream: producers of the forward's X / y inputs may run on it (see
        scgg_denoiser_side_stream in include/scgg.h)."""
        eng = self._ensure_engine(device)
        p = _lib.lib().scgg_denoiser_side_stream(eng)
        if not p:
            raise _lib.ScggError("scgg_denoiser_side_stream: " + _lib.lib().scgg_last_error().decode())
        return torch.cuda.ExternalStream(p, device=device)

    # ------------------------------------------------------------------------------------------------- forward
    dropout_seed = None     # training mode: seed of the Philox dropout masks of the next forward (None: no dropout)

    def forward(self, model_input):
        """model_input: object with .X[B,n,dX] .E[B,n,n,dE] .y[B,dy] .node_mask[B,n]  ->  PlaceHolder(X, charges, E,
        y, node_mask), masked (reference transformer_model.py:344-389).  Eval mode: the fused HIP engine, no autograd.
        Training mode: the differentiable forward of training.py (device tensor ops + the dropout kernel)."""
        if self.training:
            from .training import train_forward
            return train_forward(self, model_input.X.float(), model_input.E.float(), model_input.y.float(),
                                 model_input.node_mask, seed=self.dropout_seed)
        return self._forward_eval(model_input)

    @torch.no_grad()
    def _forward_eval(self, model_input):
        X, E, y, m, B, n = self._inputs(model_input)
        eng = self._ensure_engine(X.device)
        ws, ws_bytes = self._workspace(B, n, X.device)
        oXc = self.out_dim_X + self.out_dim_charges
        out_X = torch.empty(B, n, oXc, device=X.device, dtype=torch.float32)
        out_E = torch.empty(B, n, n, self.out_dim_E, device=X.device, dtype=torch.float32)
        out_y = torch.empty(B, self.out_dim_y, device=X.device, dtype=torch.float32) if self.predicts_final_y else None
        _lib.check(_lib.lib().scgg_denoiser_forward(
            eng, _lib.ptr(X), _lib.ptr(E), _lib.ptr(y), _lib.ptr(m.view(torch.uint8)), B, n,
            _lib.ptr(out_X) if oXc > 0 else None, _lib.ptr(out_E), _lib.ptr(out_y), ws, ws_bytes, _lib.stream_ptr()),
            "denoiser_forward")
        return PlaceHolder(X=out_X[..., : self.out_dim_X], charges=out_X[..., self.out_dim_X:], E=out_E, y=out_y,
                           node_mask=model_input.node_mask)

    @torch.no_grad()
    def forward_hidden(self, model_input, n_run_layers):
        """Test tap: hidden (X, E, y) after the input MLPs and ``n_run_layers`` layers (E symmetrised and masked)."""
        X, E, y, m, B, n = self._inputs(model_input)
        eng = self._ensure_engine(X.device)
        ws, ws_bytes = self._workspace(B, n, X.device)
        hd = self.hidden_dims
        hX = torch.empty(B, n, hd["dx"], device=X.device)
        hE = torch.empty(B, n, n, hd["de"], device=X.device)
        hy = torch.empty(B, hd["dy"], device=X.device)
        _lib.check(_lib.lib().scgg_denoiser_forward_hidden(
            eng, _lib.ptr(X), _lib.ptr(E), _lib.ptr(y), _lib.ptr(m.view(torch.uint8)), B, n, n_run_layers,
            _lib.ptr(hX), _lib.ptr(hE), _lib.ptr(hy), ws, ws_bytes, _lib.stream_ptr()), "denoiser_forward_hidden")
        return hX, hE, hy


def _profile(self, every_n_forwards, max_launches=4096):
    """Live HIP-event timing of the fused edge-layer kernel (bench.py's roofline leg)."""
    if self._engine is None:
        raise _lib.ScggError("run one forward before enabling the profile")
    _lib.check(_lib.lib().scgg_denoiser_profile(self._engine, every_n_forwards, max_launches), "profile")


def _profile_read(self):
    n, ms = C.c_int32(0), C.c_double(0.0)
    _lib.check(_lib.lib().scgg_denoiser_profile_read(self._engine, C.byref(n), C.byref(ms)), "profile_read")
    return n.value, ms.value


GraphTransformer.profile_edge_layer = _profile
GraphTransformer.read_edge_layer_profile = _profile_read


def model_input(X, E, y, node_mask):
    """Convenience: the minimal object GraphTransformer.forward needs."""
    return SimpleNamespace(X=X, E=E, y=y, node_mask=node_mask)


def load_lightning_checkpoint(model: "GraphTransformer", checkpoint, prefix: str = "model."):
    """Load the denoiser's weights out of a checkpoint of the reference's Lightning module.

    The reference restores ``Discrete3dDenoisingDiffusion`` with ``model.load_state_dict(checkpoint["state_dict"])``
    (edge_denoiser_efficiency_check.py:198-199; main.py:113-126 through ``ckpt_path``), where the denoiser is the
    attribute ``self.model`` (diffusion_model_discrete_3d.py:111-132): its parameters are the keys ``model.<name>``.
    ``checkpoint``: a path (read with ``torch.load(weights_only=True)``: nothing in the file is executed), the loaded
    dict, or its ``state_dict``.  The keys under ``prefix`` must match this ``GraphTransformer`` exactly (strict); every
    other key (metric buffers, noise-model tables) is returned, untouched, as the list of ignored names."""
    import os
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
    sd = checkpoint.get("state_dict", checkpoint)
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if not own:
        raise KeyError(f"no '{prefix}*' keys in the checkpoint ({len(sd)} keys)")
    model.load_state_dict(own, strict=True)
    return [k for k in sd if not k.startswith(prefix)]
