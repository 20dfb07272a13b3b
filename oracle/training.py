"""Oracle: the training step of the reference on the CPU (eager torch autograd over oracle/denoiser.py).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/ConStruct
  diffusion/noise_model.py            apply_noise :187-233
  metrics/train_metrics.py            TrainLoss.forward :25-85 (lambda_train = [X, charges, E, y]; radius term :84)
  metrics/abstract_metrics.py         CrossEntropyMetric :77-93 (sum of CE / number of rows)
  diffusion_model_discrete_3d.py      training_step :215-239, forward :184-201
  models/transformer_model.py         dropout sites :88-105 (layers), :284-340 (in/out MLPs when dropout_in_and_out)

Dropout.  The reference draws its masks from torch's global RNG; a bit-exact comparison needs an explicit stream, so the
mask of dropout site ``s`` is a deterministic function of (seed, site, flat element index): Philox4x32-10 with counter
((index >> 2) low, (index >> 2) high, site, 0xD0), key = seed and output word (index & 3), element kept iff u >= p, scaled by 1/(1-p) -- the same function the
HIP kernel evaluates (csrc/train_kernels.hip).  Sites are numbered in call order (``dropout_sites``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import denoiser as D
from . import noise as N


def apply_noise(tr: N.Transition, X0, E0, node_mask, t_int, ux, u0, pos=None):
    """noise_model.py:187-233 -> dict(X_t one-hot, E_t one-hot (masked), labels_E int64 (-1 = zero row), t, pos)."""
    B, n = node_mask.shape
    qtb_x, qtb_e = tr.Qt_bar(t_int)
    probX = X0 @ qtb_x                                                        # :200
    probE = E0 @ qtb_e[:, None]                                               # :201
    xs = N.sample_nodes(probX, node_mask, ux)
    es, _ = N.sample_edges(probE, node_mask, u0, None)
    pm = (node_mask[:, :, None] & node_mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    X_t = F.one_hot(xs, tr.Cx).float() * node_mask[:, :, None]
    E_t = F.one_hot(es, tr.Ce).float() * pm[..., None]
    out = dict(X_t=X_t, E_t=E_t, labels_E=torch.where(pm, es, torch.full_like(es, -1)), t=t_int.float() / tr.s.T)
    if pos is not None:                                                       # PlaceHolder.mask (utils.py:160-164)
        p = pos * node_mask[..., None]
        out["pos"] = p - p.mean(dim=1, keepdim=True)
    return out


def train_loss(pred_X, pred_E, true_X, true_E, node_mask, lambda_train=(1.0, 2.0, 5.0, 0.0)):
    """TrainLoss.forward (train_metrics.py:25-85) for the CoW / ATM setting (no charges, no y, no radius)."""
    B, n = node_mask.shape
    edge_mask = ~torch.eye(n, dtype=torch.bool)[None] & node_mask[:, :, None] & node_mask[:, None, :]
    pX, tX = pred_X[node_mask], true_X[node_mask]
    pE, tE = pred_E[edge_mask], true_E[edge_mask]
    loss_X = F.cross_entropy(pX, tX.argmax(-1), reduction="sum") / pX.shape[0] if tX.numel() > 0 else 0.0
    loss_E = F.cross_entropy(pE, tE.argmax(-1), reduction="sum") / pE.shape[0] if tE.numel() > 0 else 0.0
    return lambda_train[0] * loss_X + lambda_train[2] * loss_E


# ---------------------------------------------------------------------------------------------- dropout stream
def dropout_keep(seed: int, site: int, numel: int, p: float) -> np.ndarray:
    """keep[numel] bool of dropout site ``site``: Philox4x32-10, counter ((idx >> 2) & 2^32-1, idx >> 34, site, 0xD0), key =
    seed, word (idx & 3) -> u in [0,1) from the top 24 bits, kept iff u >= p (four consecutive elements share one call)."""
    calls = (numel + 3) // 4
    idx = np.arange(calls, dtype=np.uint64)
    ctr = np.zeros((calls, 4), dtype=np.uint32)
    ctr[:, 0] = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (idx >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(site)
    ctr[:, 3] = np.uint32(0xD0)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    u = N.u01(N.philox4x32_10(ctr, key)).reshape(-1)[:numel]
    return u >= np.float32(p)


class Dropper:
    """Numbers the dropout sites in call order and applies the Philox masks (p = 0 or seed None: identity)."""

    def __init__(self, p: float, seed=None):
        self.p, self.seed, self.site = float(p), seed, 0

    def __call__(self, x):
        s = self.site
        self.site += 1
        if self.p <= 0.0 or self.seed is None:
            return x
        keep = torch.from_numpy(dropout_keep(self.seed, s, x.numel(), self.p)).reshape(x.shape)
        return x * keep.to(x.dtype) / (1.0 - self.p)


def xey_layer_train(sd, p, n_head, X, E, y, node_mask, predicts_y, drop):
    """XEyTransformerLayer.forward in train mode (transformer_model.py:76-112): sites in the reference's order
    X1, X2, X3, E1, E2, E3, y1, y2, y3."""
    newX, newE, new_y = D.node_edge_block(sd, p + "self_attn.", n_head, X, E, y, node_mask, predicts_y)
    X = D._ln(sd, p + "normX1", X + drop(newX))
    X = D._ln(sd, p + "normX2", X + drop(D._lin(sd, p + "linX2", drop(torch.relu(D._lin(sd, p + "linX1", X))))))
    E = D._ln(sd, p + "normE1", E + drop(newE))
    E = D._ln(sd, p + "normE2", E + drop(D._lin(sd, p + "linE2", drop(torch.relu(D._lin(sd, p + "linE1", E))))))
    E = 0.5 * (E + E.transpose(1, 2))
    if predicts_y:
        y = D._ln(sd, p + "norm_y1", y + drop(new_y))
        y = D._ln(sd, p + "norm_y2", y + drop(D._lin(sd, p + "lin_y2", drop(torch.relu(D._lin(sd, p + "lin_y1", y))))))
    else:
        y = None
    X, E = D.mask_xe(X, E, node_mask)
    return X, E, y


def graph_transformer_train(sd, dims: D.DenoiserDims, X, E, y, node_mask, p_layer=0.1, p_io=0.0, seed=None):
    """GraphTransformer.forward in train mode.  ``p_layer``: the layers' dropout (the reference hard-wires 0.1: the ctor
    argument never reaches XEyTransformerLayer, transformer_model.py:309-317); ``p_io``: the in/out MLP dropout
    (``dropout`` when ``dropout_in_and_out``).  Site order: mlp_in_E (2), mlp_in_X (2), mlp_in_y (2), layers (9 or 6 each),
    mlp_out_X (1), mlp_out_E (1), mlp_out_y (1)."""
    n = X.shape[1]
    dio, dl = Dropper(p_io, seed), Dropper(p_layer, seed)

    def mlp_in(pfx, x):
        return dio(torch.relu(D._lin(sd, pfx + ".3", dio(torch.relu(D._lin(sd, pfx + ".0", x))))))

    def mlp_out(pfx, x):
        return D._lin(sd, pfx + ".3", dio(torch.relu(D._lin(sd, pfx + ".0", x))))
    X_res, E_res, y_res = X[..., : dims.out_X + dims.out_charges], E[..., : dims.out_E], y[..., : dims.out_y]
    hE = mlp_in("mlp_in_E", E)
    hE = (hE + hE.transpose(1, 2)) / 2
    hX = mlp_in("mlp_in_X", X)
    hy = mlp_in("mlp_in_y", y)
    hX, hE = D.mask_xe(hX, hE, node_mask)
    dl.site = 100                                                              # layer sites start at 100
    for l in range(dims.n_layers):
        predicts_y = (l != dims.n_layers - 1) or dims.out_y > 0
        dl.site = 100 + 16 * l
        hX, hE, hy = xey_layer_train(sd, f"tf_layers.{l}.", dims.n_head, hX, hE, hy, node_mask, predicts_y, dl)
    dio.site = 50
    oX = mlp_out("mlp_out_X", hX) + X_res
    offdiag = (1.0 - torch.eye(n, dtype=E.dtype))[None, :, :, None]
    oE = (mlp_out("mlp_out_E", hE) + E_res) * offdiag
    oE = 0.5 * (oE + oE.transpose(1, 2))
    oy = mlp_out("mlp_out_y", hy) + y_res if dims.out_y > 0 else None
    oX, oE = D.mask_xe(oX, oE, node_mask)
    return oX[..., : dims.out_X], oX[..., dims.out_X:], oE, oy


def training_step(sd, dims, tr, X0, E0, node_mask, t_int, ux, u0, extra=None, p_layer=0.0, p_io=0.0, seed=None,
                  lambda_train=(1.0, 2.0, 5.0, 0.0)):
    """diffusion_model_discrete_3d.py:215-239 -> (loss, grads by parameter name, z_t dict, logits_E).
    ``extra``: optional callable (X_t, E_t, node_mask) -> (X_extra, E_extra, y_extra) (reference forward :186-199)."""
    zt = apply_noise(tr, X0, E0, node_mask, t_int, ux, u0)
    X_in, E_in, y_in = zt["X_t"], zt["E_t"], zt["t"]
    if extra is not None:
        xe, ee, ye = extra(zt["X_t"], zt["E_t"], node_mask)
        X_in, E_in, y_in = torch.cat([X_in, xe], -1), torch.cat([E_in, ee], -1), torch.cat([ye, y_in], 1)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oX, _, oE, _ = graph_transformer_train(leaf, dims, X_in, E_in, y_in, node_mask, p_layer, p_io, seed)
    loss = train_loss(oX, oE, X0, E0, node_mask, lambda_train)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), grads, zt, oE.detach()
