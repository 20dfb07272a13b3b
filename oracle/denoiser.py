"""Oracle: functional fp32 restatement of the GraphTransformer denoiser.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Works on a plain ``state_dict``
(reference parameter names) so that it needs no nn.Module.

Follows /root/reference/src/ConStruct/models/transformer_model.py
  GraphTransformer.forward        :344-389
  XEyTransformerLayer.forward     :76-112
  NodeEdgeBlock.forward           :165-248
and /root/reference/src/ConStruct/models/layers.py (Xtoy :5-21, Etoy :24-43,
masked_softmax :46-51) and PlaceHolder.mask (src/ConStruct/utils.py:134-167).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class DenoiserDims:
    n_layers: int
    n_head: int
    out_X: int
    out_E: int
    out_y: int
    out_charges: int = 0


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(sd, name, x, eps=1e-5):
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[name + ".bias"], eps)


def mask_xe(X, E, node_mask):
    """PlaceHolder.mask for X and E (utils.py:149-155): X*m, E*m_i*m_j*(1-I)."""
    m = node_mask.to(X.dtype)
    n = node_mask.shape[1]
    offdiag = 1.0 - torch.eye(n, dtype=X.dtype, device=X.device)
    X = X * m[:, :, None]
    E = E * (m[:, :, None] * m[:, None, :] * offdiag[None])[..., None]
    return X, E


def pooled_stats_nodes(X, node_mask):
    """Xtoy pooling (layers.py:11-19): [mean | min | max | biased variance] over valid nodes."""
    m = node_mask.to(X.dtype)[:, :, None]
    cnt = m.sum(dim=1)
    mean = X.sum(dim=1) / cnt
    big = 1e5 * (1.0 - m)
    mn = (X + big).min(dim=1).values
    mx = (X - big).max(dim=1).values
    var = (((X - mean[:, None, :]) ** 2) * m).sum(dim=1) / cnt
    return torch.cat([mean, mn, mx, var], dim=1)


def pooled_stats_edges(E, node_mask):
    """Etoy pooling (layers.py:34-41).  The mask is m_i*m_j WITHOUT removing the diagonal,
    so the (zero) diagonal takes part in all four statistics and the divisor is n_valid^2."""
    m = node_mask.to(E.dtype)
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    cnt = pm.sum(dim=(1, 2))
    mean = E.sum(dim=(1, 2)) / cnt
    big = 1e5 * (1.0 - pm)
    mn = (E + big).amin(dim=(1, 2))
    mx = (E - big).amax(dim=(1, 2))
    var = (((E - mean[:, None, None, :]) ** 2) * pm).sum(dim=(1, 2)) / cnt
    return torch.cat([mean, mn, mx, var], dim=1)


def node_edge_block(sd, p, n_head, X, E, y, node_mask, predicts_y=True):
    """NodeEdgeBlock.forward (transformer_model.py:165-248).  ``p`` is the key prefix."""
    B, n, dx = X.shape
    df = dx // n_head
    m = node_mask.to(X.dtype)
    xm = m[:, :, None]
    pm = (m[:, :, None] * m[:, None, :])[..., None]          # m_i*m_j

    Q = _lin(sd, p + "q", X) * xm
    K = _lin(sd, p + "k", X) * xm
    V = _lin(sd, p + "v", X) * xm
    # Y[b,i,j,c] = Q[b,i,c]*K[b,j,c]/sqrt(df)  (:184-192; follow the code, not the comments)
    Y = (Q[:, :, None, :] * K[:, None, :, :]) / math.sqrt(df)
    E1 = _lin(sd, p + "e_mul", E) * pm
    E2 = _lin(sd, p + "e_add", E) * pm
    Y = Y * (E1 + 1.0) + E2                                   # :202

    ye_add = _lin(sd, p + "y_e_add", y)[:, None, None, :]
    ye_mul = _lin(sd, p + "y_e_mul", y)[:, None, None, :]
    newE = _lin(sd, p + "e_out", ye_add + (ye_mul + 1.0) * Y) * pm   # :205-211

    # per-channel softmax over the key axis j, padded keys excluded (:215-216, layers.py:46-51)
    if bool(node_mask.any()):
        logits = Y.masked_fill(~node_mask[:, None, :, None], float("-inf"))
        attn = torch.softmax(logits, dim=2)
    else:
        attn = Y
    wV = (attn * V[:, None, :, :]).sum(dim=2)                 # :218-227

    yx_add = _lin(sd, p + "y_x_add", y)[:, None, :]
    yx_mul = _lin(sd, p + "y_x_mul", y)[:, None, :]
    newX = _lin(sd, p + "x_out", yx_add + (yx_mul + 1.0) * wV) * xm  # :230-235

    new_y = None
    if predicts_y:
        z = (_lin(sd, p + "y_y", y)
             + _lin(sd, p + "x_y.lin", pooled_stats_nodes(X, node_mask))
             + _lin(sd, p + "e_y.lin", pooled_stats_edges(E, node_mask)))
        new_y = _lin(sd, p + "y_out.2", torch.relu(_lin(sd, p + "y_out.0", z)))  # :239-244
    return newX, newE, new_y


def xey_layer(sd, p, n_head, X, E, y, node_mask, predicts_y=True):
    """XEyTransformerLayer.forward in eval mode (dropout = identity) (:76-112)."""
    newX, newE, new_y = node_edge_block(sd, p + "self_attn.", n_head, X, E, y, node_mask, predicts_y)
    X = _ln(sd, p + "normX1", X + newX)
    X = _ln(sd, p + "normX2", X + _lin(sd, p + "linX2", torch.relu(_lin(sd, p + "linX1", X))))
    E = _ln(sd, p + "normE1", E + newE)
    E = _ln(sd, p + "normE2", E + _lin(sd, p + "linE2", torch.relu(_lin(sd, p + "linE1", E))))
    E = 0.5 * (E + E.transpose(1, 2))
    if predicts_y:
        y = _ln(sd, p + "norm_y1", y + new_y)
        y = _ln(sd, p + "norm_y2", y + _lin(sd, p + "lin_y2", torch.relu(_lin(sd, p + "lin_y1", y))))
    else:
        y = None
    X, E = mask_xe(X, E, node_mask)
    return X, E, y


def _mlp_in(sd, p, x):
    return torch.relu(_lin(sd, p + ".3", torch.relu(_lin(sd, p + ".0", x))))


def _mlp_out(sd, p, x):
    return _lin(sd, p + ".3", torch.relu(_lin(sd, p + ".0", x)))


def graph_transformer_forward(sd, dims: DenoiserDims, X, E, y, node_mask, return_hidden=False):
    """GraphTransformer.forward (transformer_model.py:344-389), eval mode.

    Returns (X_out[B,n,out_X], charges[B,n,out_c], E_out[B,n,n,out_E], y_out or None)."""
    n = X.shape[1]
    X_res = X[..., : dims.out_X + dims.out_charges]
    E_res = E[..., : dims.out_E]
    y_res = y[..., : dims.out_y]

    hE = _mlp_in(sd, "mlp_in_E", E)
    hE = (hE + hE.transpose(1, 2)) / 2                         # :359
    hX = _mlp_in(sd, "mlp_in_X", X)
    hy = _mlp_in(sd, "mlp_in_y", y)
    hX, hE = mask_xe(hX, hE, node_mask)
    hidden = [(hX, hE, hy)]

    for l in range(dims.n_layers):
        predicts_y = (l != dims.n_layers - 1) or dims.out_y > 0   # :316
        hX, hE, hy = xey_layer(sd, f"tf_layers.{l}.", dims.n_head, hX, hE, hy, node_mask, predicts_y)
        hidden.append((hX, hE, hy))

    oX = _mlp_out(sd, "mlp_out_X", hX) + X_res
    offdiag = (1.0 - torch.eye(n, dtype=E.dtype, device=E.device))[None, :, :, None]
    oE = (_mlp_out(sd, "mlp_out_E", hE) + E_res) * offdiag      # :375
    oE = 0.5 * (oE + oE.transpose(1, 2))
    oy = _mlp_out(sd, "mlp_out_y", hy) + y_res if dims.out_y > 0 else None

    oX, oE = mask_xe(oX, oE, node_mask)
    out = (oX[..., : dims.out_X], oX[..., dims.out_X:], oE, oy)
    return (out, hidden) if return_hidden else out
