"""Drop-in for ``src/ConStruct/models/pos_transformer_model.py`` and ``equivariant_pos_transformer_model.py`` of the
reference: ``Graph3DTransformer`` / ``EquivariantGraph3DTransformer`` with the same constructor arguments, the same
``state_dict`` names and shapes (a Lightning checkpoint's ``model.*`` keys load, ``transformer_model.load_lightning_checkpoint``)
and the same ``forward(model_input) -> PlaceHolder`` contract (pos_transformer_model.py:338-493; the driver builds one
of the two at diffusion_model_discrete_3d.py:111-132).

Only the parameter containers live here.  The arithmetic of forward runs in libscgg.so (HIP, gfx950) through
``scgg_denoiser3d_forward``; there is no eager fallback for inference.  In training mode ``forward`` is the differentiable
device-tensor forward of ``training.train_forward_3d`` (dropout through the Philox kernel), as for the 2D class.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .placeholder import PlaceHolder
from .transformer_model import _dim, _two_linears, _Pool


class NodeEdgeBlock(nn.Module):
    """Parameters of the reference's position-aware NodeEdgeBlock (pos_transformer_model.py:36-104)."""

    def __init__(self, dx, de, dy, n_head, predicts_y=False):
        super().__init__()
        assert dx % n_head == 0, f"dx: {dx} -- nhead: {n_head}"
        self.dx, self.de, self.dy, self.n_head, self.df = dx, de, dy, n_head, dx // n_head
        self.predicts_y = predicts_y
        self.in_E = nn.Linear(de, de)
        self.x_e_mul1, self.x_e_mul2 = nn.Linear(dx, de), nn.Linear(dx, de)
        self.lin_dist1 = nn.Linear(2, de)
        self.lin_norm_pos1, self.lin_norm_pos2 = nn.Linear(1, de), nn.Linear(1, de)
        self.dist_add_e, self.dist_mul_e = nn.Linear(de, de), nn.Linear(de, de)
        self.k, self.q, self.v = nn.Linear(dx, dx), nn.Linear(dx, dx), nn.Linear(dx, dx)
        self.a = nn.Linear(dx, n_head, bias=False)
        self.out = nn.Linear(dx * n_head, dx)
        self.e_att_mul, self.pos_att_mul = nn.Linear(de, n_head), nn.Linear(de, n_head)
        self.e_x_mul, self.pos_x_mul = _Pool(de, dx), _Pool(de, dx)          # EtoX (layers.py:55-58)
        self.y_e_mul, self.y_e_add = nn.Linear(dy, de), nn.Linear(dy, de)
        self.pre_softmax = nn.Linear(de, dx)                                  # unused by forward; part of the state_dict (:82)
        self.y_x_mul, self.y_x_add = nn.Linear(dy, dx), nn.Linear(dy, dx)
        if predicts_y:
            self.y_y = nn.Linear(dy, dy)
            self.x_y, self.e_y, self.dist_y = _Pool(dx, dy), _Pool(de, dy), _Pool(de, dy)
        self.e_pos1 = nn.Linear(de, de, bias=False)                           # unused by forward (:97-98)
        self.e_pos2 = nn.Linear(de, 1, bias=False)
        self.x_out, self.e_out = nn.Linear(dx, dx), nn.Linear(de, de)
        if predicts_y:
            self.y_out = nn.Sequential(nn.Linear(dy, dy), nn.Identity(), nn.Linear(dy, dy))


class XEyTransformerLayer(nn.Module):
    """Parameters of the reference's XEyTransformerLayer (pos_transformer_model.py:223-256)."""

    def __init__(self, dx, de, dy, n_head, dim_ffX=2048, dim_ffE=128, dim_ffy=2048, dropout=0.1, layer_norm_eps=1e-5,
                 predicts_y=True, **_):
        super().__init__()
        self.predicts_y = predicts_y
        self.self_attn = NodeEdgeBlock(dx, de, dy, n_head, predicts_y=predicts_y)
        self.linX1, self.linX2 = nn.Linear(dx, dim_ffX), nn.Linear(dim_ffX, dx)
        self.normX1, self.normX2 = nn.LayerNorm(dx, eps=layer_norm_eps), nn.LayerNorm(dx, eps=layer_norm_eps)
        self.linE1, self.linE2 = nn.Linear(de, dim_ffE), nn.Linear(dim_ffE, de)
        self.normE1, self.normE2 = nn.LayerNorm(de, eps=layer_norm_eps), nn.LayerNorm(de, eps=layer_norm_eps)
        if predicts_y:
            self.lin_y1, self.lin_y2 = nn.Linear(dy, dim_ffy), nn.Linear(dim_ffy, dy)
            self.norm_y1, self.norm_y2 = nn.LayerNorm(dy, eps=layer_norm_eps), nn.LayerNorm(dy, eps=layer_norm_eps)
        self.dim_ffy = dim_ffy


class Graph3DTransformer(nn.Module):
    """Graph3DTransformer(n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, dropout, dropout_in_and_out)
    -- reference pos_transformer_model.py:344-428.  ``dim_ffy`` of hidden_dims is not passed to the layers there either
    (:400-407): lin_y1 is dy -> 2048."""

    equivariant = False

    def __init__(self, n_layers, input_dims, hidden_mlp_dims, hidden_dims, output_dims, dropout=0.1,
                 dropout_in_and_out=False):
        super().__init__()
        self.n_layers = n_layers
        self.in_dim_X = _dim(input_dims, "X") + _dim(input_dims, "charges")
        self.in_dim_E, self.in_dim_y = _dim(input_dims, "E"), _dim(input_dims, "y")
        self.out_dim_X, self.out_dim_E = _dim(output_dims, "X"), _dim(output_dims, "E")
        self.out_dim_y, self.out_dim_charges = _dim(output_dims, "y"), _dim(output_dims, "charges")
        self.out_dim_radius = 1
        self.predicts_final_y = self.out_dim_y > 0
        if self.predicts_final_y:
            raise _lib.ScggError("output_dims.y > 0: the reference class has no mlp_out_y (pos_transformer_model.py:479-481)")
        self.hidden_mlp_dims, self.hidden_dims = dict(hidden_mlp_dims), dict(hidden_dims)
        self.dropout, self.dropout_in_and_out = dropout, dropout_in_and_out
        hm, hd = self.hidden_mlp_dims, self.hidden_dims
        self.mlp_in_X = _two_linears(self.in_dim_X, hm["X"], hd["dx"])
        self.mlp_in_E = _two_linears(self.in_dim_E, hm["E"], hd["de"])
        self.mlp_in_y = _two_linears(self.in_dim_y, hm["y"], hd["dy"])
        self.mlp_in_pos_high_dim = _two_linears(3, hm["X"], hd["dx"])
        self.tf_layers = nn.ModuleList([
            XEyTransformerLayer(dx=hd["dx"], de=hd["de"], dy=hd["dy"], n_head=hd["n_head"], dim_ffX=hd["dim_ffX"],
                                dim_ffE=hd["dim_ffE"], predicts_y=(l != n_layers - 1) or self.predicts_final_y)
            for l in range(n_layers)])
        self.mlp_out_X = _two_linears(hd["dx"], hm["X"], self.out_dim_X + self.out_dim_charges)
        self.mlp_out_E = _two_linears(hd["de"], hm["E"], self.out_dim_E)
        self.mlp_out_radius = _two_linears(hd["de"], hm["E"], self.out_dim_radius)
        self._engine, self._engine_key, self._versions, self._ws = None, None, None, {}

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _config(self):
        hm, hd = self.hidden_mlp_dims, self.hidden_dims
        return _lib.Denoiser3DConfig(
            n_layers=self.n_layers, in_X=self.in_dim_X, in_E=self.in_dim_E, in_y=self.in_dim_y,
            out_X=self.out_dim_X + self.out_dim_charges, out_E=self.out_dim_E, hid_X=hm["X"], hid_E=hm["E"], hid_y=hm["y"],
            dx=hd["dx"], de=hd["de"], dy=hd["dy"], n_head=hd["n_head"], dim_ffX=hd["dim_ffX"], dim_ffE=hd["dim_ffE"],
            dim_ffy=self.tf_layers[0].dim_ffy, equivariant=int(self.equivariant))

    def _release(self):
        if self._engine is not None:
            _lib.lib().scgg_denoiser3d_destroy(self._engine)
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
            h = L.scgg_denoiser3d_create(C.byref(cfg))
            if not h:
                raise _lib.ScggError("scgg_denoiser3d_create: " + L.scgg_last_error().decode())
            self._engine = C.c_void_p(h)
            for k, v in sd.items():
                _lib.check(L.scgg_denoiser3d_set_param(self._engine, k.encode(), _lib.ptr(v.detach()), v.numel()), k)
            _lib.check(L.scgg_denoiser3d_finalize(self._engine, _lib.stream_ptr()), "finalize")
            self._engine_key, self._versions = key, versions
        elif versions != self._versions:      # weights changed in place: pack again
            _lib.check(L.scgg_denoiser3d_finalize(self._engine, _lib.stream_ptr()), "finalize")
            self._versions = versions
        return self._engine

    def _workspace(self, B, n, device):
        k = (B, n, str(device))
        if k not in self._ws:
            nbytes = _lib.lib().scgg_denoiser3d_workspace_bytes(self._engine, B, n)
            self._ws = {k: torch.empty(nbytes + 256, dtype=torch.uint8, device=device)}
        ws = self._ws[k]
        off = (-ws.data_ptr()) % 256
        return C.c_void_p(ws.data_ptr() + off), ws.numel() - 256

    def _inputs(self, mi):
        X, E, y, pos, m = mi.X, mi.E, mi.y, mi.pos, mi.node_mask
        if not X.is_cuda:
            raise _lib.ScggError("Graph3DTransformer.forward runs on the MI355X only; move the inputs to the GPU")
        B, n = X.shape[0], X.shape[1]
        if X.shape[2] != self.in_dim_X or E.shape[3] != self.in_dim_E or y.shape[1] != self.in_dim_y or pos.shape != (B, n, 3):
            raise _lib.ScggError(f"input widths {X.shape[2]},{E.shape[3]},{y.shape[1]}, pos {tuple(pos.shape)} do not match the "
                                 f"model ({self.in_dim_X},{self.in_dim_E},{self.in_dim_y}, ({B},{n},3))")
        f = lambda t: t.detach().float().contiguous()
        return f(X), f(E), f(y), f(pos), m.to(torch.bool).contiguous(), B, n

    # ------------------------------------------------------------------------------------------------- forward
    dropout_seed = None     # training mode: seed of the Philox dropout masks of the next forward (None: no dropout)

    def forward(self, model_input):
        """model_input: .X[B,n,dX] .E[B,n,n,dE] .y[B,dy] .pos[B,n,3] .radius (None: no radius head) .node_mask[B,n]
        -> PlaceHolder(pos, X, charges, E, y=None, node_mask, radius), masked as the reference's final ``.mask()`` does
        (pos masked and centred over all n rows, utils.py:160-164).  Eval mode: the fused HIP engine, no autograd.  Training
        mode: the differentiable forward of training.py (device tensor ops + the dropout kernel)."""
        if self.training:
            from .training import train_forward_3d
            return train_forward_3d(self, model_input.X.float(), model_input.E.float(), model_input.y.float(),
                                    model_input.pos.float(), model_input.node_mask,
                                    with_radius=getattr(model_input, "radius", None) is not None, seed=self.dropout_seed)
        return self._forward_eval(model_input)

    @torch.no_grad()
    def _forward_eval(self, model_input):
        X, E, y, pos, m, B, n = self._inputs(model_input)
        eng = self._ensure_engine(X.device)
        ws, ws_bytes = self._workspace(B, n, X.device)
        oXc = self.out_dim_X + self.out_dim_charges
        out_X = torch.empty(B, n, oXc, device=X.device, dtype=torch.float32)
        out_E = torch.empty(B, n, n, self.out_dim_E, device=X.device, dtype=torch.float32)
        with_radius = getattr(model_input, "radius", None) is not None
        out_R = torch.empty(B, n, n, 1, device=X.device, dtype=torch.float32) if with_radius else None
        _lib.check(_lib.lib().scgg_denoiser3d_forward(
            eng, _lib.ptr(X), _lib.ptr(E), _lib.ptr(y), _lib.ptr(pos), _lib.ptr(m.view(torch.uint8)), B, n,
            _lib.ptr(out_X) if oXc > 0 else None, _lib.ptr(out_E), _lib.ptr(out_R), ws, ws_bytes, _lib.stream_ptr()),
            "denoiser3d_forward")
        pm = pos * m[..., None]
        return PlaceHolder(pos=pm - pm.mean(dim=1, keepdim=True), X=out_X[..., : self.out_dim_X], charges=out_X[..., self.out_dim_X:],
                           E=out_E, y=None, node_mask=model_input.node_mask, radius=out_R)

    @torch.no_grad()
    def forward_hidden(self, model_input, n_run_layers):
        """Test tap: hidden (X, E, y) after the input MLPs and ``n_run_layers`` layers (E symmetrised and masked)."""
        X, E, y, pos, m, B, n = self._inputs(model_input)
        eng = self._ensure_engine(X.device)
        ws, ws_bytes = self._workspace(B, n, X.device)
        hd = self.hidden_dims
        hX = torch.empty(B, n, hd["dx"], device=X.device)
        hE = torch.empty(B, n, n, hd["de"], device=X.device)
        hy = torch.empty(B, hd["dy"], device=X.device)
        _lib.check(_lib.lib().scgg_denoiser3d_forward_hidden(
            eng, _lib.ptr(X), _lib.ptr(E), _lib.ptr(y), _lib.ptr(pos), _lib.ptr(m.view(torch.uint8)), B, n, n_run_layers,
            _lib.ptr(hX), _lib.ptr(hE), _lib.ptr(hy), ws, ws_bytes, _lib.stream_ptr()), "denoiser3d_forward_hidden")
        return hX, hE, hy


class EquivariantGraph3DTransformer(Graph3DTransformer):
    """equivariant_pos_transformer_model.py:334-486: same parameters and block; the layers start from mlp_in_X(X) and the
    node head reads the hidden X."""

    equivariant = True
