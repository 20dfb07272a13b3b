"""Oracle: functional fp32 restatement of the position-aware denoiser.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Works on a plain ``state_dict`` with the reference's parameter names.

Follows /root/reference/src/ConStruct/models/pos_transformer_model.py
  NodeEdgeBlock.forward          :106-209
  XEyTransformerLayer.forward    :258-303
  Graph3DTransformer.forward     :433-493
/root/reference/src/ConStruct/models/equivariant_pos_transformer_model.py
  EquivariantGraph3DTransformer.forward :428-486 (same block; node features from mlp_in_X, node head on the hidden X)
and /root/reference/src/ConStruct/models/layers.py (EtoX :55-71, Xtoy :5-21, Etoy :24-43).

Quirks kept on purpose:
* Xtoy is applied to the UNMASKED newX (:192, :202): padded rows hold y_x_add and take part in the mean's numerator
  (layers.py:15 sums every row), not in the divisor, minimum, maximum or variance.
* e_y pools Y, the edge features BEFORE the y FiLM (:201), while e_x_mul pools the FiLMed E (:183).
* PlaceHolder.mask() (utils.py:160-164) masks the positions and then subtracts their mean over ALL n rows, padded ones
  included, every time it runs: once on the model input and once after every layer.  For a graph with padding the mean
  of the masked, centred positions is not zero, so every layer sees positions shifted a little further (distances between
  valid nodes do not change, norms and cosines do).  ``recentre`` restates it.
* the diagonal pair (i, i) of a valid node is a valid pair inside the block (zero input features, distance 0, cosine 1
  or 0 for a node at the origin); PlaceHolder.mask() removes it again after every layer.
"""
from __future__ import annotations

import math

import torch

from oracle.denoiser import DenoiserDims, _lin, _ln, _mlp_out, mask_xe, pooled_stats_edges


def pooled_stats_rows(E, node_mask):
    """EtoX pooling (layers.py:60-69): per query row i over the keys j, mask on j only."""
    m = node_mask.to(E.dtype)[:, None, :, None]                 # e_mask2
    cnt = m.sum(dim=2)
    mean = E.sum(dim=2) / cnt
    big = 1e5 * (1.0 - m)
    mn = (E + big).min(dim=2).values
    mx = (E - big).max(dim=2).values
    var = (((E - mean[:, :, None, :]) ** 2) * m).sum(dim=2) / cnt
    return torch.cat([mean, mn, mx, var], dim=2)


def pooled_stats_nodes_unmasked_sum(X, node_mask):
    """Xtoy (layers.py:11-19) on an X whose padded rows are NOT zero: the sum of the mean runs over every row."""
    m = node_mask.to(X.dtype)[:, :, None]
    cnt = m.sum(dim=1)
    mean = X.sum(dim=1) / cnt
    big = 1e5 * (1.0 - m)
    mn = (X + big).min(dim=1).values
    mx = (X - big).max(dim=1).values
    var = (((X - mean[:, None, :]) ** 2) * m).sum(dim=1) / cnt
    return torch.cat([mean, mn, mx, var], dim=1)


def recentre(pos, node_mask):
    """PlaceHolder.mask on pos (utils.py:160-164): mask, then subtract the mean over every row of the padded batch."""
    pos = pos * node_mask.to(pos.dtype)[:, :, None]
    return pos - pos.mean(dim=1, keepdim=True)


DIST_MODE = "cdist"


def distance_features(sd, p, pos, node_mask):
    """dist1 of NodeEdgeBlock.forward :122-132.

    ``DIST_MODE``: the reference calls torch.cdist (:126), which for more than 25 points switches from differences to the
    expanded form |a|^2 + |b|^2 - 2 a.b evaluated by a matrix product; there the distance of a node to itself (and of
    coincident nodes) is the square root of rounding noise, up to ~2e-3 for |a| ~ 3, and depends on the BLAS at hand.
    "cdist" calls the same library function and is what the reference's CPU fixtures are compared with; "direct" takes
    differences at every size (exact zeros on the diagonal), which is what the HIP kernel computes."""
    m = node_mask.to(pos.dtype)
    pos = pos * m[:, :, None]
    norm = torch.linalg.vector_norm(pos, dim=-1, keepdim=True)
    unit = pos / (norm + 1e-7)
    if DIST_MODE == "cdist":
        dist = torch.cdist(pos, pos).unsqueeze(-1)
    else:
        diff = pos[:, :, None, :] - pos[:, None, :, :]
        dist = torch.sqrt((diff * diff).sum(-1, keepdim=True))
    cos = (unit[:, None, :, :] * unit[:, :, None, :]).sum(-1, keepdim=True)
    info = torch.cat([dist, cos], dim=-1)
    n1 = _lin(sd, p + "lin_norm_pos1", norm)
    n2 = _lin(sd, p + "lin_norm_pos2", norm)
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    return torch.relu(_lin(sd, p + "lin_dist1", info) + n1[:, :, None, :] + n2[:, None, :, :]) * pm


def node_edge_block_3d(sd, p, n_head, X, E, y, pos, node_mask, predicts_y=True, return_parts=False):
    m = node_mask.to(X.dtype)
    xm = m[:, :, None]
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    de = E.shape[-1]
    dist1 = distance_features(sd, p, pos, node_mask)

    Y = _lin(sd, p + "in_E", E)
    xe1 = _lin(sd, p + "x_e_mul1", X) * xm
    xe2 = _lin(sd, p + "x_e_mul2", X) * xm
    Y = Y * xe1[:, None, :, :] * xe2[:, :, None, :] * pm        # :142  (mul1 follows the key j, mul2 the query i)
    Y = (Y + _lin(sd, p + "dist_add_e", dist1) + Y * _lin(sd, p + "dist_mul_e", dist1)) * pm
    E1 = (Y + _lin(sd, p + "y_e_add", y)[:, None, None, :] + Y * _lin(sd, p + "y_e_mul", y)[:, None, None, :]) * pm
    Eout = _lin(sd, p + "e_out", E1) * pm

    Q = _lin(sd, p + "q", X) * xm
    K = _lin(sd, p + "k", X) * xm
    V = _lin(sd, p + "v", X) * xm
    prod = Q[:, :, None, :] * K[:, None, :, :] / math.sqrt(de)  # :161 (Y.size(-1) is de)
    a = torch.nn.functional.linear(prod, sd[p + "a.weight"]) * pm
    a = a + _lin(sd, p + "e_att_mul", E1) * a
    a = a + _lin(sd, p + "pos_att_mul", dist1) * a
    a = a * pm
    if bool(node_mask.any()):                                   # masked_softmax over the keys (layers.py:46-51)
        alpha = torch.softmax(a.masked_fill(~node_mask[:, None, :, None], float("-inf")), dim=2)
    else:
        alpha = a
    wV = (alpha[..., None] * V[:, None, :, None, :]).sum(dim=2)  # [B, n, h, dx]
    wV = _lin(sd, p + "out", wV.flatten(start_dim=2)) * xm
    wV = wV + _lin(sd, p + "e_x_mul.lin", pooled_stats_rows(E1, node_mask)) * wV
    wV = wV + _lin(sd, p + "pos_x_mul.lin", pooled_stats_rows(dist1, node_mask)) * wV
    newX = wV * (_lin(sd, p + "y_x_mul", y)[:, None, :] + 1.0) + _lin(sd, p + "y_x_add", y)[:, None, :]
    Xout = _lin(sd, p + "x_out", newX) * xm

    new_y = None
    if predicts_y:
        z = (_lin(sd, p + "y_y", y) + _lin(sd, p + "x_y.lin", pooled_stats_nodes_unmasked_sum(newX, node_mask))
             + _lin(sd, p + "e_y.lin", pooled_stats_edges(Y, node_mask))
             + _lin(sd, p + "dist_y.lin", pooled_stats_edges(dist1, node_mask)))
        new_y = _lin(sd, p + "y_out.2", torch.relu(_lin(sd, p + "y_out.0", z)))
    if return_parts:
        return Xout, Eout, new_y, dict(dist1=dist1, Y=Y, E1=E1, a=a, newX=newX)
    return Xout, Eout, new_y


def xey_layer_3d(sd, p, n_head, X, E, y, pos, node_mask, predicts_y=True):
    newX, newE, new_y = node_edge_block_3d(sd, p + "self_attn.", n_head, X, E, y, pos, node_mask, predicts_y)
    X = _ln(sd, p + "normX1", X + newX)
    E = _ln(sd, p + "normE1", E + newE)
    X = _ln(sd, p + "normX2", X + _lin(sd, p + "linX2", torch.relu(_lin(sd, p + "linX1", X))))
    E = _ln(sd, p + "normE2", E + _lin(sd, p + "linE2", torch.relu(_lin(sd, p + "linE1", E))))
    E = 0.5 * (E + E.transpose(1, 2))
    if predicts_y:
        y = _ln(sd, p + "norm_y1", y + new_y)
        y = _ln(sd, p + "norm_y2", y + _lin(sd, p + "lin_y2", torch.relu(_lin(sd, p + "lin_y1", y))))
    else:
        y = None
    X, E = mask_xe(X, E, node_mask)
    return X, E, y, recentre(pos, node_mask)


def _seq_relu_relu(sd, p, x):                                   # Linear, ReLU, (Dropout), Linear, ReLU in either order of the
    return torch.relu(_lin(sd, p + ".3", torch.relu(_lin(sd, p + ".0", x))))   # last two modules: identical in eval mode


def graph3d_transformer_forward(sd, dims: DenoiserDims, X, E, y, pos, node_mask, with_radius=True, equivariant=False,
                                return_hidden=False):
    """Graph3DTransformer.forward (:433-493) / EquivariantGraph3DTransformer.forward (:428-486), eval mode.
    Returns (X_out, charges, E_out, radius or None)."""
    assert dims.out_y == 0, "neither class has an mlp_out_y (pos_transformer_model.py:479-481 would raise)"
    n = X.shape[1]
    X_res = X[..., : dims.out_X + dims.out_charges]
    E_res = E[..., : dims.out_E]
    hE = _seq_relu_relu(sd, "mlp_in_E", E)
    hE = (hE + hE.transpose(1, 2)) / 2
    hX = _seq_relu_relu(sd, "mlp_in_X", X) if equivariant else _seq_relu_relu(sd, "mlp_in_pos_high_dim", pos)
    hy = _seq_relu_relu(sd, "mlp_in_y", y)
    hX, hE = mask_xe(hX, hE, node_mask)
    hpos = recentre(pos, node_mask)
    hidden = [(hX, hE, hy, hpos)]
    for l in range(dims.n_layers):
        predicts_y = l != dims.n_layers - 1
        hX, hE, hy, hpos = xey_layer_3d(sd, f"tf_layers.{l}.", dims.n_head, hX, hE, hy, hpos, node_mask, predicts_y)
        hidden.append((hX, hE, hy, hpos))
    if equivariant:
        oX = _mlp_out(sd, "mlp_out_X", hX) + X_res
    else:
        oX = _mlp_out(sd, "mlp_out_X", _seq_relu_relu(sd, "mlp_in_X", X)) + X_res      # :467: the node head never sees the layers
    offdiag = (1.0 - torch.eye(n, dtype=E.dtype))[None, :, :, None]
    oE = (_mlp_out(sd, "mlp_out_E", hE) + E_res) * offdiag
    oE = 0.5 * (oE + oE.transpose(1, 2))
    oX, oE = mask_xe(oX, oE, node_mask)
    rad = None
    if with_radius:
        rad = _mlp_out(sd, "mlp_out_radius", hE) * offdiag
        rad = 0.5 * (rad + rad.transpose(1, 2))
        m = node_mask.to(rad.dtype)
        rad = rad * (m[:, :, None] * m[:, None, :])[..., None]
    out = (oX[..., : dims.out_X], oX[..., dims.out_X:], oE, rad)
    return (out, hidden) if return_hidden else out
