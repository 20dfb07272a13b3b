"""Training step of the reference on the MI355X: drop-in pieces for
``src/ConStruct/diffusion_model_discrete_3d.py`` ``training_step`` (:215-239) and ``forward`` (:184-201),
``src/ConStruct/metrics/train_metrics.py`` ``TrainLoss`` (:25-85) and the data-parallel gradient exchange Lightning's DDP
strategy does for it (src/ConStruct/main.py: ``Trainer(strategy="ddp_find_unused_parameters_true")``).

What runs where (round 2):
  * ``apply_noise``, the extra features, the dropout masks (``scgg_dropout``: Philox per dropout site, re-derived in the
    backward pass instead of stored) and the loss with its gradient (``scgg_masked_cross_entropy``) are libscgg.so kernels;
  * the train-mode forward and backward of the network itself are device tensor ops under autograd (rocBLAS GEMMs) over the
    same parameter tensors the fused inference engine borrows -- NOT the fused HIP kernels: those have no backward yet.
    ``GraphTransformer.forward`` dispatches here when the module is in training mode;
  * ``GradientBuckets`` averages the gradients over the ranks with bucketed ``all_reduce`` (RCCL on the GPUs, gloo in the CPU
    tests), the only collective of the path.

Dropout of the reference (quirk kept): ``GraphTransformer`` never hands its ``dropout`` argument to its layers, so the
layers always drop with p = 0.1 in training mode (transformer_model.py:309-317); the constructor argument only reaches the
in / out MLPs, and only with ``dropout_in_and_out``.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from . import _lib
from .placeholder import PlaceHolder


# ------------------------------------------------------------------------------------------------- kernels under autograd
def _is_device_key(seed):
    """A dropout seed held in device memory: one int64 (or two int32 words, low first) on the GPU.  The kernels read it when
    they RUN, so a captured HIP graph draws new masks on every replay once the caller has written a new seed there."""
    return isinstance(seed, torch.Tensor)


def _check_device_key(seed):
    if not (seed.is_cuda and seed.is_contiguous() and seed.numel() * seed.element_size() == 8):
        raise _lib.ScggError("a device-resident dropout seed is one int64 (or two int32) on the GPU")
    return seed


def _dropout_launch(x, out, p, seed, site, what):
    if _is_device_key(seed):
        _lib.check(_lib.lib().scgg_dropout_devkey(_lib.ptr(x), _lib.ptr(out), x.numel(), p, _check_device_key(seed).data_ptr(), site,
                                                  _lib.stream_ptr()), what)
    else:
        _lib.check(_lib.lib().scgg_dropout(_lib.ptr(x), _lib.ptr(out), x.numel(), p, int(seed), site, _lib.stream_ptr()), what)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, site):
        ctx.p, ctx.seed, ctx.site = p, seed, site
        x = x.contiguous()
        out = torch.empty_like(x)
        _dropout_launch(x, out, p, seed, site, "dropout")
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        _dropout_launch(g, out, ctx.p, ctx.seed, ctx.site, "dropout (backward)")
        return out, None, None, None


class DropoutSites:
    """Numbers the dropout sites in call order, as oracle/training.py does (a site is counted even when p = 0)."""

    def __init__(self, p, seed):
        self.p, self.seed, self.site = float(p), seed, 0

    def __call__(self, x):
        s = self.site
        self.site += 1
        if self.p <= 0.0 or self.seed is None:
            return x
        if not x.is_cuda or x.dtype != torch.float32:
            raise _lib.ScggError("dropout runs on the MI355X in fp32 only")
        return _Dropout.apply(x, self.p, self.seed if _is_device_key(self.seed) else int(self.seed), s)


class _MaskedCE(torch.autograd.Function):
    """sum of cross entropies over the rows whose target is >= 0, / their number (CrossEntropyMetric, abstract_metrics.py:77-93)."""

    @staticmethod
    def forward(ctx, logits, target, count):
        logits = logits.contiguous()
        rows, C_ = logits.numel() // logits.shape[-1], logits.shape[-1]
        nb = int(_lib.lib().scgg_masked_cross_entropy_blocks(rows))
        partial = torch.empty(max(nb, 1), device=logits.device)
        _lib.check(_lib.lib().scgg_masked_cross_entropy(_lib.ptr(logits), _lib.ptr(target), rows, C_, _lib.ptr(partial), None, None,
                                                        _lib.stream_ptr()), "masked_cross_entropy")
        ctx.save_for_backward(logits, target, count)
        return partial[:nb].sum() / count

    @staticmethod
    def backward(ctx, g):
        logits, target, count = ctx.saved_tensors
        rows, C_ = logits.numel() // logits.shape[-1], logits.shape[-1]
        scale = (g / count).reshape(1).float().contiguous()
        grad = torch.empty_like(logits)
        _lib.check(_lib.lib().scgg_masked_cross_entropy(_lib.ptr(logits), _lib.ptr(target), rows, C_, None, _lib.ptr(grad),
                                                        _lib.ptr(scale), _lib.stream_ptr()), "masked_cross_entropy (backward)")
        return grad, None, None


def masked_cross_entropy(logits, target):
    """logits [..., C] fp32 on the device, target int8 [...] (-1 = row masked out).  0 when no row takes part."""
    if not logits.is_cuda:
        raise _lib.ScggError("masked_cross_entropy runs on the MI355X only")
    target = target.to(torch.int8).contiguous()
    count = (target >= 0).sum().clamp(min=1).float()
    return _MaskedCE.apply(logits.float(), target, count)


class TrainLoss(torch.nn.Module):
    """train_metrics.py:25-85 for the discrete node / edge terms (CoW / ATM setting: no charges, no y, no radius term):
    ``lambda_train = [X, charges, E, y]``; ``forward(pred, true_X_labels, true_E_labels, node_mask)`` with integer label
    tensors of the clean data (the reference takes one-hots and argmaxes them, :40-47)."""

    def __init__(self, lambda_train=(1.0, 2.0, 5.0, 0.0)):
        super().__init__()
        self.lambda_train = tuple(float(v) for v in lambda_train)

    def forward(self, pred, true_X_labels, true_E_labels, node_mask, true_radius=None):
        """``true_radius`` [B,n,n,1] (the clean data's edge radii): adds the reference's radius term, the mean squared error
        over the valid off-diagonal pairs scaled like the edge term (train_metrics.py:45-53, :76-83)."""
        n = node_mask.shape[1]
        eye = torch.eye(n, dtype=torch.bool, device=node_mask.device)[None]
        live = node_mask[:, :, None] & node_mask[:, None, :] & ~eye
        tE = torch.where(live, true_E_labels.to(torch.int8), torch.full_like(true_E_labels, -1, dtype=torch.int8))
        tX = torch.where(node_mask, true_X_labels.to(torch.int8), torch.full_like(true_X_labels, -1, dtype=torch.int8))
        loss_X = masked_cross_entropy(pred.X, tX)
        loss_E = masked_cross_entropy(pred.E, tE)
        loss = self.lambda_train[0] * loss_X + self.lambda_train[2] * loss_E
        if true_radius is not None and getattr(pred, "radius", None) is not None and bool(live.any()):
            loss = loss + self.lambda_train[2] * ((pred.radius[live] - true_radius[live]) ** 2).mean()
        return loss


# ------------------------------------------------------------------------------------------------- train-mode forward
def _lin(m, x):
    return F.linear(x, m.weight, m.bias)


def _ln(m, x):
    return F.layer_norm(x, m.normalized_shape, m.weight, m.bias, m.eps)


class _SymMask(torch.autograd.Function):
    """E -> 0.5 (E + E^T) on the pairs of valid nodes i != j, 0 elsewhere (transformer_model.py:108-112 + PlaceHolder.mask) in
    one pass (``scgg_sym_mask``); the map is linear and self-adjoint, so the backward is the same call on the gradient.  The
    same bits as the tensor ops it replaces (a scale by 0.5 is exact, the mask is 0 / 1)."""

    @staticmethod
    def forward(ctx, R, mask_u8):
        ctx.save_for_backward(mask_u8)
        R = R.contiguous()
        out = torch.empty_like(R)
        B, n, _, Cc = R.shape
        _lib.check(_lib.lib().scgg_sym_mask(_lib.ptr(R), _lib.ptr(mask_u8), B, n, Cc, _lib.ptr(out), _lib.stream_ptr()), "sym_mask")
        return out

    @staticmethod
    def backward(ctx, g):
        (mask_u8,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        B, n, _, Cc = g.shape
        _lib.check(_lib.lib().scgg_sym_mask(_lib.ptr(g), _lib.ptr(mask_u8), B, n, Cc, _lib.ptr(out), _lib.stream_ptr()), "sym_mask (backward)")
        return out, None


def _sym_mask(E, mask_u8):
    if E.is_cuda and E.dtype == torch.float32 and os.environ.get("SCGG_TRAIN_MODE", "fused") != "eager":
        return _SymMask.apply(E, mask_u8)
    m = mask_u8.to(E.dtype)
    n = E.shape[1]
    w = m[:, :, None] * m[:, None, :] * (1.0 - torch.eye(n, dtype=E.dtype, device=E.device))[None]
    return (0.5 * (E + E.transpose(1, 2))) * w[..., None]


def _mask_xe(X, E, m, offdiag):
    return X * m[:, :, None], E * (m[:, :, None] * m[:, None, :] * offdiag)[..., None]


class _NodeStats(torch.autograd.Function):
    """Xtoy's pooled statistics (layers.py:12-21) as one kernel each way (``scgg_train_node_stats``)."""

    @staticmethod
    def forward(ctx, X, mask_u8):
        X = X.contiguous()
        B, n, dx = X.shape
        out = torch.empty(B, 4 * dx, device=X.device)
        aux = torch.empty(B, 3 * dx, device=X.device)
        arg = torch.empty(B, 2 * dx, dtype=torch.int32, device=X.device)
        _lib.check(_lib.lib().scgg_train_node_stats(_lib.ptr(X), _lib.ptr(mask_u8), B, n, dx, _lib.ptr(out), _lib.ptr(aux), _lib.ptr(arg),
                                                    _lib.stream_ptr()), "train_node_stats")
        ctx.save_for_backward(X, mask_u8, aux, arg)
        return out

    @staticmethod
    def backward(ctx, g):
        X, mask_u8, aux, arg = ctx.saved_tensors
        B, n, dx = X.shape
        dX = torch.empty_like(X)
        _lib.check(_lib.lib().scgg_train_node_stats_backward(_lib.ptr(X), _lib.ptr(mask_u8), B, n, dx, _lib.ptr(g.float().contiguous()),
                                                             _lib.ptr(aux), _lib.ptr(arg), _lib.ptr(dX), _lib.stream_ptr()),
                   "train_node_stats_backward")
        return dX, None


def _node_stats(X, m, mask_u8=None):
    if mask_u8 is not None and X.is_cuda and X.dtype == torch.float32 and os.environ.get("SCGG_TRAIN_MODE", "fused") != "eager":
        return _NodeStats.apply(X, mask_u8)
    w = m[:, :, None]
    cnt = w.sum(dim=1)
    mean = X.sum(dim=1) / cnt
    big = 1e5 * (1.0 - w)
    var = (((X - mean[:, None, :]) ** 2) * w).sum(dim=1) / cnt
    return torch.cat([mean, (X + big).min(dim=1).values, (X - big).max(dim=1).values, var], dim=1)


def _edge_stats(E, m):
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    cnt = pm.sum(dim=(1, 2))
    mean = E.sum(dim=(1, 2)) / cnt
    big = 1e5 * (1.0 - pm)
    var = (((E - mean[:, None, None, :]) ** 2) * pm).sum(dim=(1, 2)) / cnt
    return torch.cat([mean, (E + big).amin(dim=(1, 2)), (E - big).amax(dim=(1, 2)), var], dim=1)


def _block(a, X, E, y, m, node_mask):
    """NodeEdgeBlock.forward (transformer_model.py:165-248) on the parameter holder ``a``."""
    xm = m[:, :, None]
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    Q, K, V = _lin(a.q, X) * xm, _lin(a.k, X) * xm, _lin(a.v, X) * xm
    Y = (Q[:, :, None, :] * K[:, None, :, :]) / math.sqrt(a.df)
    Y = Y * (_lin(a.e_mul, E) * pm + 1.0) + _lin(a.e_add, E) * pm
    newE = _lin(a.e_out, _lin(a.y_e_add, y)[:, None, None, :] + (_lin(a.y_e_mul, y)[:, None, None, :] + 1.0) * Y) * pm
    attn = torch.softmax(Y.masked_fill(~node_mask[:, None, :, None], float("-inf")), dim=2)
    wV = (attn * V[:, None, :, :]).sum(dim=2)
    newX = _lin(a.x_out, _lin(a.y_x_add, y)[:, None, :] + (_lin(a.y_x_mul, y)[:, None, :] + 1.0) * wV) * xm
    new_y = None
    if a.predicts_y:
        z = _lin(a.y_y, y) + _lin(a.x_y.lin, _node_stats(X, m)) + _lin(a.e_y.lin, _edge_stats(E, m))
        new_y = _lin(a.y_out[2], torch.relu(_lin(a.y_out[0], z)))
    return newX, newE, new_y


# ------------------------------------------------------------------------------------------------- fused edge layer
_EDGE_W = ("e_mul.weight", "e_add.weight", "e_out.weight", "linE1.weight", "linE2.weight")


def fused_edge_supported(dx, de, ffe):
    """Shapes csrc/train_edge.hip is built for; SCGG_TRAIN_MODE=eager keeps the unfused tensor-op path (the checker)."""
    import os
    return os.environ.get("SCGG_TRAIN_MODE", "fused") != "eager" and dx % 16 == 0 and (de, ffe) in ((32, 64), (64, 128))


def _pack(W, transpose):
    W = W.detach().float().contiguous()
    out = torch.empty(W.numel(), device=W.device)
    _lib.check(_lib.lib().scgg_train_edge_pack(_lib.ptr(W), W.shape[0], W.shape[1], int(transpose), _lib.ptr(out), _lib.stream_ptr()),
               "train_edge_pack")
    return out


def _pack_layer(Ws):
    """Fragment images of a layer's five weights, plain and transposed, in one launch: (five forward images, five backward)."""
    L = _lib.lib()
    sizes = [W.numel() for W in Ws]
    flat = torch.empty(2 * sum(sizes), device=Ws[0].device)
    pl = _lib.TrainPackList()
    views, off = [], 0
    for t in range(10):
        W = Ws[t % 5]
        v = flat[off:off + W.numel()]
        off += W.numel()
        pl.W[t], pl.out[t], pl.rows[t], pl.cols[t], pl.transpose[t] = W.data_ptr(), v.data_ptr(), W.shape[0], W.shape[1], int(t >= 5)
        views.append(v)
    pl.count = 10
    _lib.check(L.scgg_train_edge_pack_multi(C.byref(pl), _lib.stream_ptr()), "train_edge_pack_multi")
    return views[:5], views[5:]


def _sum_rows(t, out=None, splits=None):
    """t [A,R,C] contiguous fp32 -> the sum over R (``scgg_train_sum_rows``: the large reductions of the backward's factor
    tensors; a fixed order of summation, so repeated steps agree bit for bit).  Without ``out``: returns [A,C].  With
    ``out`` [splits, B', C] (B' >= A, a slice [:, b0:b0+A] of a larger buffer is fine): fills it and leaves the sum over the
    leading index to the caller (one per layer instead of one per call)."""
    A, R, Cc = t.shape
    L = _lib.lib()
    sp = int(L.scgg_train_sum_rows_splits(A, R, Cc)) if splits is None else int(splits)
    if out is None:
        o = torch.empty(sp, A, Cc, device=t.device)
        _lib.check(L.scgg_train_sum_rows(_lib.ptr(t), A, R, Cc, sp, _lib.ptr(o), 0, _lib.stream_ptr()), "train_sum_rows")
        return o.sum(0) if sp > 1 else o[0]
    assert out.shape[0] == sp and out.shape[1] == A and out.shape[2] == Cc and out.stride(2) == 1 and out.stride(1) == Cc
    _lib.check(L.scgg_train_sum_rows(_lib.ptr(t), A, R, Cc, sp, out.data_ptr(), out.stride(0) if sp > 1 else 0, _lib.stream_ptr()),
               "train_sum_rows")
    return out


WGRAD_SLAB = 256


def _wgrad_into(G, X, partial, row0, col0, slab=WGRAD_SLAB):
    """Partials of (G^T X | G.sum(0)) into rows [row0, row0 + slabs) of ``partial`` [rows, ld] at column col0."""
    S, M = G.shape
    N = X.shape[1]
    ld = partial.shape[1]
    _lib.check(_lib.lib().scgg_train_wgrad(_lib.ptr(G), M, _lib.ptr(X), N, S, slab, partial.data_ptr() + 4 * (row0 * ld + col0), ld,
                                           _lib.stream_ptr()), "train_wgrad")


def _wgrad(G, X, slab=None):
    """(G^T X, G.sum(0)) for the factor tensors of the backward kernel: split over the pair slots in the library's own
    MFMA kernel (HBM-bound; the generic GEMM took 0.6 ms per call at K = 2.6e5), slabs added in a fixed order."""
    S, M = G.shape
    N = X.shape[1]
    L = _lib.lib()
    if slab is None:                       # ~4 workgroups per CU over (slabs x 64x64 tiles), slabs of 256 .. 4096 slots
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        slab = max(256, min(4096, (S * tiles // 1024) // 16 * 16))
    nsl = int(L.scgg_train_wgrad_slabs(S, slab))
    ld = (M * N + M + 3) // 4 * 4
    partial = torch.empty(nsl, ld, device=G.device)
    _wgrad_into(G, X, partial, 0, 0, slab)
    tot = _sum_rows(partial.view(1, nsl, ld))[0] if nsl >= 64 else partial.sum(0)
    return tot[: M * N].view(M, N), tot[M * N: M * N + M]


class _PairLinear(torch.autograd.Function):
    """nn.Linear over the [B,n,n,.] tensors of the in / out MLPs of E in training mode: forward and data gradient are the
    library GEMM (rows = pairs: its good case); the weight gradient, a reduction over ~2.6e5 pairs into a small matrix, is
    ``scgg_train_wgrad`` (operands zero-padded to multiples of 16 columns when the widths are not)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return F.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        M, N = g2.shape[1], x2.shape[1]
        Mp, Np = (M + 15) // 16 * 16, (N + 15) // 16 * 16
        if g2.shape[0] >= 4096:
            gp = g2.contiguous() if Mp == M else F.pad(g2, (0, Mp - M))
            xp = x2.contiguous() if Np == N else F.pad(x2, (0, Np - N))
            dW, db = _wgrad(gp, xp)
            dW, db = dW[:M, :N], db[:M]
        else:
            dW, db = g2.t() @ x2, g2.sum(0)
        return g @ W, dW, db


def _plin(m, x):
    return _PairLinear.apply(x, m.weight, m.bias)


class _FusedEdgeLayer(torch.autograd.Function):
    """The [B,n,n,.] part of one XEyTransformerLayer in training mode as two HIP kernels (csrc/train_edge.hip):
    (E, Q, K, V, yea, yem1) -> (E_out before symmetrisation / masking, wV).  Saves its inputs and (wV, lse) only; the
    backward kernel recomputes the chain per 16-key unit, returns dE and the factors of the weight gradients, and
    ``scgg_train_wgrad`` reduces those factors over the pair slots into the gradients of the five weight matrices."""

    @staticmethod
    def forward(ctx, E, Q, K, V, yea, yem1, mask_u8, inv_sqrt_df, p, seed, sites, with_stats,
                W_mul, b_mul, W_add, b_add, W_out, b_out, g1, b1, W_ff1, b_ff1, W_ff2, b_ff2, g2, b2):
        B, n, _, de = E.shape
        dx, ffe = Q.shape[-1], W_ff1.shape[0]
        nch = (n + 15) // 16
        E, Q, K, V, yea, yem1 = (t.detach().float().contiguous() for t in (E, Q, K, V, yea, yem1))
        vecs = [t.detach().float().contiguous() for t in (b_mul, b_add, b_out, g1, b1, b_ff1, b_ff2, g2, b2)]
        Ws = [W.detach().float().contiguous() for W in (W_mul, W_add, W_out, W_ff1, W_ff2)]
        F_fwd, F_bwd = _pack_layer(Ws)                 # the backward's transposed images too: one launch, ~0.3 MB kept
        a = _lib.TrainEdgeArgs()
        keep = [E, Q, K, V, yea, yem1, mask_u8] + vecs + F_fwd
        for k, t in zip(("E", "Q", "K", "V", "yea", "yem1", "mask"), (E, Q, K, V, yea, yem1, mask_u8)):
            setattr(a, k, t.data_ptr())
        for k, t in zip(("F_mul", "F_add", "F_out", "F_ff1", "F_ff2"), F_fwd):
            setattr(a, k, t.data_ptr())
        for k, t in zip(("b_mul", "b_add", "b_out", "g1", "b1", "b_ff1", "b_ff2", "g2", "b2"), vecs):
            setattr(a, k, t.data_ptr())
        E_out = torch.zeros_like(E)
        part = torch.empty(B * n * nch, 3, dx, device=E.device)
        wV = torch.empty(B, n, dx, device=E.device)
        lse = torch.empty(B, n, dx, device=E.device)
        a.E_out, a.part, a.wV_out, a.lse_out = E_out.data_ptr(), part.data_ptr(), wV.data_ptr(), lse.data_ptr()
        a.B, a.n, a.nch, a.dx, a.de, a.ffe = B, n, nch, dx, de, ffe
        a.inv_sqrt_df, a.p = float(inv_sqrt_df), float(p) if seed is not None else 0.0
        key = _check_device_key(seed) if _is_device_key(seed) else None
        sd = int(seed) if seed is not None and key is None else 0
        a.k0, a.k1 = sd & 0xFFFFFFFF, (sd >> 32) & 0xFFFFFFFF
        a.key_dev = key.data_ptr() if key is not None else None
        a.site1, a.site2, a.site3 = (int(v) for v in sites)
        _lib.check(_lib.lib().scgg_train_edge_forward(C.byref(a), _lib.stream_ptr()), "train_edge_forward")
        del keep
        # Etoy's pooled statistics of the layer INPUT (they feed the y update): one more read of E, merged over the slices
        stats = None
        st_keep = [torch.empty(0, device=E.device)] * 2 + [torch.empty(0, dtype=torch.int32, device=E.device)] * 2
        if not with_stats:
            stats = torch.zeros(B, 4 * de, device=E.device)
        else:
            slices = max(1, min(n, 256 // max(B, 1)))
            so = torch.empty(B, slices, 5, de, device=E.device)
            si = torch.empty(B, slices, 2, de, dtype=torch.int32, device=E.device)
            _lib.check(_lib.lib().scgg_train_edge_stats(_lib.ptr(E), _lib.ptr(mask_u8), B, n, de, slices, _lib.ptr(so), _lib.ptr(si),
                                                        _lib.stream_ptr()), "train_edge_stats")
            stats = torch.empty(B, 4 * de, device=E.device)
            mt = torch.empty(2, B, de, device=E.device)
            ix = torch.empty(2, B, de, dtype=torch.int32, device=E.device)
            _lib.check(_lib.lib().scgg_train_edge_stats_finish(_lib.ptr(so), _lib.ptr(si), B, slices, de, _lib.ptr(stats), _lib.ptr(mt[0]),
                                                               _lib.ptr(mt[1]), _lib.ptr(ix[0]), _lib.ptr(ix[1]), _lib.stream_ptr()),
                       "train_edge_stats_finish")
            st_keep = [mt[0], mt[1], ix[0], ix[1]]
        ctx.with_stats = bool(with_stats)
        ctx.save_for_backward(E, Q, K, V, yea, yem1, mask_u8, wV, lse, *vecs, *Ws, *st_keep, *F_fwd, *F_bwd)
        ctx.cfg = (float(inv_sqrt_df), a.p, sd, tuple(int(v) for v in sites))
        ctx.key = key
        ctx.mark_non_differentiable(lse)
        return E_out, wV, lse, stats

    @staticmethod
    def backward(ctx, dOut, dwV, _dlse, dstats):
        E, Q, K, V, yea, yem1, mask_u8, wV, lse = ctx.saved_tensors[:9]
        vecs = list(ctx.saved_tensors[9:18])
        Ws = list(ctx.saved_tensors[18:23])
        st_mean, st_tot, st_imn, st_imx = ctx.saved_tensors[23:27]
        F_fwd, F_bwd = ctx.saved_tensors[27:32], ctx.saved_tensors[32:37]
        inv_sqrt_df, p, sd, sites = ctx.cfg
        B, n, _, de = E.shape
        dx, ffe = Q.shape[-1], Ws[3].shape[0]
        nch = (n + 15) // 16
        dev = E.device
        dOut = torch.zeros_like(E) if dOut is None else dOut.float().contiguous()
        dwV = torch.zeros_like(wV) if dwV is None else dwV.float().contiguous()
        a = _lib.TrainEdgeArgs()
        for k, t in zip(("F_mul", "F_add", "F_out", "F_ff1", "F_ff2"), F_fwd):
            setattr(a, k, t.data_ptr())
        for k, t in zip(("F_mulT", "F_addT", "F_outT", "F_ff1T", "F_ff2T"), F_bwd):
            setattr(a, k, t.data_ptr())
        for k, t in zip(("b_mul", "b_add", "b_out", "g1", "b1", "b_ff1", "b_ff2", "g2", "b2"), vecs):
            setattr(a, k, t.data_ptr())
        dE = torch.empty_like(E)
        st = None
        if ctx.with_stats and dstats is not None:
            gm, gmi, gma, gv = (t.contiguous() for t in dstats.float().split(de, dim=1))
            st = [st_mean, (gm / st_tot).contiguous(), (2.0 * gv / st_tot).contiguous(), gmi, gma]
        # The factor tensors are the backward's transient memory (4.3 KB per pair slot): the batch is walked in chunks of
        # graphs so that they stay below ~0.6 GB.  What the chunks leave behind is summed ONCE per layer: the partials of the
        # four weight-gradient products of every chunk share one buffer (rows = chunk x slab, columns = product), and so do
        # the per-graph sums; with per-chunk sums and adds the tail of a chunk was ~35 small launches.
        per_graph = n * nch * 16 * (4 * de + 2 * ffe + 5 * dx) * 4
        Bc = max(1, min(B, int(0.6 * 2 ** 30) // max(per_graph, 1)))
        Sc = Bc * n * nch * 16
        G = {k: torch.empty(Sc, w, device=dev) for k, w in (("G_e", de), ("G_da", 2 * dx), ("G_z", dx), ("G_dnewE", de), ("G_h1", de),
                                                           ("G_df", ffe), ("G_f", ffe), ("G_do", de), ("G_dkv", 2 * dx))}
        UW = 2 * dx + 4 * de
        upart = torch.empty(Bc * n * nch, UW, device=dev)
        for k, t in G.items():
            setattr(a, k, t.data_ptr())
        a.upart = upart.data_ptr()
        a.n, a.nch, a.dx, a.de, a.ffe = n, nch, dx, de, ffe
        a.inv_sqrt_df, a.p = inv_sqrt_df, p
        a.k0, a.k1 = sd & 0xFFFFFFFF, (sd >> 32) & 0xFFFFFFFF
        a.key_dev = ctx.key.data_ptr() if ctx.key is not None else None
        a.site1, a.site2, a.site3 = sites
        L = _lib.lib()
        prods = (("G_da", "G_e"), ("G_dnewE", "G_z"), ("G_df", "G_h1"), ("G_do", "G_f"))     # [2dx,de] = d e_mul | d e_add weights, ...
        widths = [G[g].shape[1] * G[x].shape[1] + G[g].shape[1] for g, x in prods]
        cols = [sum(widths[:i]) for i in range(5)]
        chunks = list(range(0, B, Bc))
        nsl = int(L.scgg_train_wgrad_slabs(Sc, WGRAD_SLAB))
        P = torch.zeros(len(chunks) * nsl, cols[4], device=dev) if B % Bc else torch.empty(len(chunks) * nsl, cols[4], device=dev)
        sp_u = int(L.scgg_train_sum_rows_splits(Bc, n, UW))
        sp_e = int(L.scgg_train_sum_rows_splits(Bc, n * nch * 16, de))
        U2 = torch.empty(sp_u, B, UW, device=dev)                   # per graph: . | d yem1 | LayerNorm parameter gradients
        Sd = torch.empty(sp_e, B, de, device=dev)                   # per graph: sum of dnewE over the pairs
        dQ = torch.empty_like(Q)
        direct_kv = nch * 16 == n and int(L.scgg_train_sum_rows_splits(Bc, n, n * 2 * dx)) == 1
        dKV = torch.empty(B, n, 2 * dx, device=dev)
        for ci, b0 in enumerate(chunks):
            nb = min(Bc, B - b0)
            for k, t in (("E", E), ("Q", Q), ("K", K), ("V", V), ("yea", yea), ("yem1", yem1), ("mask", mask_u8), ("dOut", dOut),
                         ("dwV", dwV), ("wV", wV), ("lse", lse), ("dE", dE)):
                setattr(a, k, t[b0:].data_ptr())
            if st is not None:
                a.st_mean, a.st_cm, a.st_cv, a.st_gmi, a.st_gma = (t[b0:].data_ptr() for t in st)
                a.st_imn, a.st_imx = st_imn[b0:].data_ptr(), st_imx[b0:].data_ptr()
            a.B, a.b0 = nb, b0
            _lib.check(L.scgg_train_edge_backward(C.byref(a), _lib.stream_ptr()), "train_edge_backward")
            Sn = nb * n * nch * 16
            for (g, x), c0 in zip(prods, cols):
                _wgrad_into(G[g][:Sn], G[x][:Sn], P, ci * nsl, c0)
            U1 = _sum_rows(upart[: nb * n * nch].view(nb * n, nch, UW), splits=1).view(nb, n, UW)     # over the key chunks
            torch.mul(U1[..., :dx], inv_sqrt_df, out=dQ[b0:b0 + nb])
            _sum_rows(U1, out=U2[:, b0:b0 + nb], splits=sp_u)                                          # over the query rows
            kv = G["G_dkv"][:Sn].view(nb, n, nch * 16 * 2 * dx)                                        # dK | dV: over the query rows
            if direct_kv:
                _sum_rows(kv, out=dKV[b0:b0 + nb].view(1, nb, n * 2 * dx), splits=1)
            else:
                dKV[b0:b0 + nb] = _sum_rows(kv).view(nb, nch * 16, 2 * dx)[:, :n]
            _sum_rows(G["G_dnewE"][:Sn].view(nb, n * nch * 16, de), out=Sd[:, b0:b0 + nb], splits=sp_e)
        rows = P.shape[0]
        tot = _sum_rows(P.view(1, rows, cols[4]))[0] if rows >= 64 else P.sum(0)
        dWs = [(tot[c0:c0 + G[g].shape[1] * G[x].shape[1]].view(G[g].shape[1], G[x].shape[1]), tot[c0 + w - G[g].shape[1]:c0 + w])
               for (g, x), c0, w in zip(prods, cols, widths)]
        (dW_ma, db_ma), (dW_out, db_out), (dW_ff1, db_ff1), (dW_ff2, db_ff2) = dWs
        U = U2.sum(0) if sp_u > 1 else U2[0]
        dyem1 = U[:, dx:2 * dx]
        ln = U[:, 2 * dx:].sum(0)
        dyea = (Sd.sum(0) if sp_e > 1 else Sd[0]) @ Ws[2]           # sum of dz = W_out^T sum of dnewE, per graph
        dg2, db2, dg1, db1 = ln[:de], ln[de:2 * de], ln[2 * de:3 * de], ln[3 * de:]
        return (dE, dQ, dKV[..., :dx], dKV[..., dx:], dyea, dyem1, None, None, None, None, None, None,
                dW_ma[:dx], db_ma[:dx], dW_ma[dx:], db_ma[dx:], dW_out, db_out, dg1, db1, dW_ff1, db_ff1, dW_ff2, db_ff2, dg2, db2)


class _YRound(torch.autograd.Function):
    """The y path of one XEyTransformerLayer in training mode (transformer_model.py:229-246 + :100-105) as three kernels
    (csrc/train_y.hip): (hy, Xtoy statistics, Etoy statistics) -> the layer's new y.  Saves its inputs and the chain's
    intermediates (a few vectors per graph); one more kernel forms all 18 parameter gradients."""

    NAMES = ("W_yy", "b_yy", "W_xy", "b_xy", "W_ey", "b_ey", "W_o0", "b_o0", "W_o2", "b_o2", "g1", "be1", "W_1", "b_1", "W_2", "b_2", "g2", "be2")

    @staticmethod
    def _args(hy, xs, es, params, p, seed, sites):
        a = _lib.TrainYArgs()
        a.hy, a.xstats, a.estats = hy.data_ptr(), xs.data_ptr(), es.data_ptr()
        for k, t in zip(_YRound.NAMES, params):
            setattr(a, k, t.data_ptr())
        a.B, a.dy, a.dx4, a.de4, a.ffy = hy.shape[0], hy.shape[1], xs.shape[1], es.shape[1], params[12].shape[0]
        key = _check_device_key(seed) if _is_device_key(seed) else None
        sd = int(seed) if seed is not None and key is None else 0
        a.p = float(p) if seed is not None else 0.0
        a.k0, a.k1 = sd & 0xFFFFFFFF, (sd >> 32) & 0xFFFFFFFF
        a.key_dev = key.data_ptr() if key is not None else None
        a.site1, a.site2, a.site3 = (int(v) for v in sites)
        return a

    @staticmethod
    def forward(ctx, hy, xs, es, p, seed, sites, *params):
        hy, xs, es = (t.detach().float().contiguous() for t in (hy, xs, es))
        params = [t.detach().float().contiguous() for t in params]
        a = _YRound._args(hy, xs, es, params, p, seed, sites)
        out = torch.empty_like(hy)
        keep = torch.empty(hy.shape[0], int(_lib.lib().scgg_train_y_keep_width(a.dy, a.ffy)), device=hy.device)
        a.out, a.fsave = out.data_ptr(), keep.data_ptr()
        _lib.check(_lib.lib().scgg_train_y_forward(C.byref(a), _lib.stream_ptr()), "train_y_forward")
        ctx.save_for_backward(hy, xs, es, keep, *params)
        ctx.cfg = (p, seed, tuple(int(v) for v in sites))
        return out

    @staticmethod
    def backward(ctx, dout):
        hy, xs, es, keep = ctx.saved_tensors[:4]
        params = list(ctx.saved_tensors[4:])
        p, seed, sites = ctx.cfg
        L = _lib.lib()
        a = _YRound._args(hy, xs, es, params, p, seed, sites)
        a.fsave = keep.data_ptr()
        dout = dout.float().contiguous()
        dhy, dxs, des = torch.empty_like(hy), torch.empty_like(xs), torch.empty_like(es)
        save = torch.empty(hy.shape[0], int(L.scgg_train_y_save_width(a.dy, a.ffy)), device=hy.device)
        a.dout, a.dhy, a.dxstats, a.destats, a.save = dout.data_ptr(), dhy.data_ptr(), dxs.data_ptr(), des.data_ptr(), save.data_ptr()
        _lib.check(L.scgg_train_y_backward(C.byref(a), _lib.stream_ptr()), "train_y_backward")
        flat = torch.empty(sum(t.numel() for t in params), device=hy.device)          # one allocation for the 18 gradients
        grads, off = [], 0
        for t in params:
            grads.append(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
        ptrs = (C.c_void_p * 18)(*[g.data_ptr() for g in grads])
        _lib.check(L.scgg_train_y_wgrad(C.byref(a), ptrs, _lib.stream_ptr()), "train_y_wgrad")
        return (dhy, dxs, des, None, None, None, *grads)


def y_round_supported(dy):
    return dy <= 256 and os.environ.get("SCGG_TRAIN_MODE", "fused") != "eager" and os.environ.get("SCGG_TRAIN_Y", "fused") != "ops"


def _layer_fused(layer, hX, hE, hy, m, node_mask, mask_u8, offdiag, dl, base):
    """One XEyTransformerLayer in training mode with the [B,n,n,.] part in the fused kernels; the X / y paths and the pooled
    statistics stay tensor ops (O(n dx^2) and O(n^2 de) work).  Dropout sites as in train_forward."""
    a = layer.self_attn
    xm = m[:, :, None]
    Q, K, V = _lin(a.q, hX) * xm, _lin(a.k, hX) * xm, _lin(a.v, hX) * xm
    E_raw, wV, _, e_stats = _FusedEdgeLayer.apply(
        hE, Q, K, V, _lin(a.y_e_add, hy), _lin(a.y_e_mul, hy) + 1.0, mask_u8, 1.0 / math.sqrt(a.df), dl.p, dl.seed,
        (base + 3, base + 4, base + 5), bool(a.predicts_y), a.e_mul.weight, a.e_mul.bias, a.e_add.weight, a.e_add.bias, a.e_out.weight, a.e_out.bias,
        layer.normE1.weight, layer.normE1.bias, layer.linE1.weight, layer.linE1.bias, layer.linE2.weight, layer.linE2.bias,
        layer.normE2.weight, layer.normE2.bias)
    newX = _lin(a.x_out, _lin(a.y_x_add, hy)[:, None, :] + (_lin(a.y_x_mul, hy)[:, None, :] + 1.0) * wV) * xm
    new_y = hy_next = None
    y_fused = bool(a.predicts_y and layer.predicts_y and hy.is_cuda and y_round_supported(hy.shape[-1]))
    if y_fused:
        ps = dl.seed if dl.p > 0.0 else None
        hy_next = _YRound.apply(hy, _node_stats(hX, m, mask_u8), e_stats, dl.p, ps, (base + 6, base + 7, base + 8),
                                a.y_y.weight, a.y_y.bias, a.x_y.lin.weight, a.x_y.lin.bias, a.e_y.lin.weight, a.e_y.lin.bias,
                                a.y_out[0].weight, a.y_out[0].bias, a.y_out[2].weight, a.y_out[2].bias, layer.norm_y1.weight, layer.norm_y1.bias,
                                layer.lin_y1.weight, layer.lin_y1.bias, layer.lin_y2.weight, layer.lin_y2.bias, layer.norm_y2.weight,
                                layer.norm_y2.bias)
    elif a.predicts_y:
        z = _lin(a.y_y, hy) + _lin(a.x_y.lin, _node_stats(hX, m, mask_u8)) + _lin(a.e_y.lin, e_stats)
        new_y = _lin(a.y_out[2], torch.relu(_lin(a.y_out[0], z)))
    dl.site = base
    hX = _ln(layer.normX1, hX + dl(newX))
    hX = _ln(layer.normX2, hX + dl(_lin(layer.linX2, dl(torch.relu(_lin(layer.linX1, hX))))))
    if y_fused:
        hy = hy_next
    elif layer.predicts_y:
        dl.site = base + 6
        hy = _ln(layer.norm_y1, hy + dl(new_y))
        hy = _ln(layer.norm_y2, hy + dl(_lin(layer.lin_y2, dl(torch.relu(_lin(layer.lin_y1, hy))))))
    else:
        hy = None
    return hX * xm, _sym_mask(E_raw, mask_u8), hy


def train_forward(model, X, E, y, node_mask, seed=None, p_layer=None):
    """GraphTransformer.forward in training mode (transformer_model.py:344-389 with the dropout sites of :88-105 and
    :284-340), differentiable.  ``seed`` None: no dropout (the fixtures' setting); ``p_layer`` defaults to the reference's
    hard-wired 0.1.  Dropout site numbers: in-MLPs 0..5 (E, X, y; two each), layer l: 100 + 16 l + (0..8) in the order
    X1, X2-inner, X2-outer, E1, E2-inner, E2-outer, y1, y2-inner, y2-outer; out-MLPs 50, 51, 52 (X, E, y)."""
    if not X.is_cuda:
        raise _lib.ScggError("the training forward runs on the MI355X only; move the inputs to the GPU")
    p_layer = 0.1 if p_layer is None else p_layer
    p_io = model.dropout if model.dropout_in_and_out else 0.0
    dio, dl = DropoutSites(p_io, seed), DropoutSites(p_layer, seed)
    node_mask = node_mask.bool()
    m = node_mask.to(X.dtype)
    n = X.shape[1]
    offdiag = (1.0 - torch.eye(n, dtype=X.dtype, device=X.device))[None]
    oXc = model.out_dim_X + model.out_dim_charges

    import os
    plin = _plin if os.environ.get("SCGG_TRAIN_MODE", "fused") != "eager" else _lin

    def mlp_in(seq, x, lin=_lin):
        return dio(torch.relu(lin(seq[3], dio(torch.relu(lin(seq[0], x))))))

    def mlp_out(seq, x, lin=_lin):
        return lin(seq[3], dio(torch.relu(lin(seq[0], x))))
    mask_u8 = node_mask.contiguous().view(torch.uint8)
    hE = _sym_mask(mlp_in(model.mlp_in_E, E, plin), mask_u8)
    hX = mlp_in(model.mlp_in_X, X) * m[:, :, None]
    hy = mlp_in(model.mlp_in_y, y)
    for l, layer in enumerate(model.tf_layers):
        dl.site = 100 + 16 * l
        if fused_edge_supported(hX.shape[-1], hE.shape[-1], layer.linE1.weight.shape[0]):
            hX, hE, hy = _layer_fused(layer, hX, hE, hy, m, node_mask, mask_u8, offdiag, dl, 100 + 16 * l)
            continue
        newX, newE, new_y = _block(layer.self_attn, hX, hE, hy, m, node_mask)
        hX = _ln(layer.normX1, hX + dl(newX))
        hX = _ln(layer.normX2, hX + dl(_lin(layer.linX2, dl(torch.relu(_lin(layer.linX1, hX))))))
        hE = _ln(layer.normE1, hE + dl(newE))
        hE = _ln(layer.normE2, hE + dl(_lin(layer.linE2, dl(torch.relu(_lin(layer.linE1, hE))))))
        hE = 0.5 * (hE + hE.transpose(1, 2))
        if layer.predicts_y:
            hy = _ln(layer.norm_y1, hy + dl(new_y))
            hy = _ln(layer.norm_y2, hy + dl(_lin(layer.lin_y2, dl(torch.relu(_lin(layer.lin_y1, hy))))))
        else:
            hy = None
        hX, hE = _mask_xe(hX, hE, m, offdiag)
    dio.site = 50
    oX = mlp_out(model.mlp_out_X, hX) + X[..., :oXc]
    oE = _sym_mask(mlp_out(model.mlp_out_E, hE, plin) + E[..., : model.out_dim_E], mask_u8)    # :386-389: off-diagonal, symmetric, masked
    oy = mlp_out(model.mlp_out_y, hy) + y[..., : model.out_dim_y] if model.predicts_final_y else None
    oX = oX * m[:, :, None]
    return PlaceHolder(X=oX[..., : model.out_dim_X], charges=oX[..., model.out_dim_X:], E=oE, y=oy, node_mask=node_mask)


def _row_stats(E, m):
    """EtoX pooling (layers.py:60-69): per query row over the keys, mask on the key only."""
    w = m[:, None, :, None]
    cnt = w.sum(dim=2)
    mean = E.sum(dim=2) / cnt
    big = 1e5 * (1.0 - w)
    var = (((E - mean[:, :, None, :]) ** 2) * w).sum(dim=2) / cnt
    return torch.cat([mean, (E + big).min(dim=2).values, (E - big).max(dim=2).values, var], dim=2)


def _recentre(pos, m):
    pos = pos * m[:, :, None]
    return pos - pos.mean(dim=1, keepdim=True)


def _block3d(a, X, E, y, pos, m, node_mask):
    """The position-aware NodeEdgeBlock.forward (pos_transformer_model.py:106-209) on the parameter holder ``a``."""
    xm = m[:, :, None]
    pm = (m[:, :, None] * m[:, None, :])[..., None]
    pos = pos * xm
    norm = torch.linalg.vector_norm(pos, dim=-1, keepdim=True)
    unit = pos / (norm + 1e-7)
    diff = pos[:, :, None, :] - pos[:, None, :, :]
    info = torch.cat([torch.sqrt((diff * diff).sum(-1, keepdim=True)), (unit[:, None] * unit[:, :, None]).sum(-1, keepdim=True)], dim=-1)
    dist1 = torch.relu(_lin(a.lin_dist1, info) + _lin(a.lin_norm_pos1, norm)[:, :, None, :] + _lin(a.lin_norm_pos2, norm)[:, None, :, :]) * pm
    Y = _lin(a.in_E, E) * (_lin(a.x_e_mul1, X) * xm)[:, None, :, :] * (_lin(a.x_e_mul2, X) * xm)[:, :, None, :] * pm
    Y = (Y + _lin(a.dist_add_e, dist1) + Y * _lin(a.dist_mul_e, dist1)) * pm
    E1 = (Y + _lin(a.y_e_add, y)[:, None, None, :] + Y * _lin(a.y_e_mul, y)[:, None, None, :]) * pm
    Eout = _lin(a.e_out, E1) * pm
    Q, K, V = _lin(a.q, X) * xm, _lin(a.k, X) * xm, _lin(a.v, X) * xm
    att = F.linear(Q[:, :, None, :] * K[:, None, :, :] / math.sqrt(a.de), a.a.weight) * pm
    att = att + _lin(a.e_att_mul, E1) * att
    att = att + _lin(a.pos_att_mul, dist1) * att
    att = att * pm
    alpha = torch.softmax(att.masked_fill(~node_mask[:, None, :, None], float("-inf")), dim=2)
    wV = torch.einsum("bijh,bjd->bihd", alpha, V).flatten(start_dim=2)
    wV = _lin(a.out, wV) * xm
    wV = wV + _lin(a.e_x_mul.lin, _row_stats(E1, m)) * wV
    wV = wV + _lin(a.pos_x_mul.lin, _row_stats(dist1, m)) * wV
    newX = wV * (_lin(a.y_x_mul, y)[:, None, :] + 1.0) + _lin(a.y_x_add, y)[:, None, :]
    Xout = _lin(a.x_out, newX) * xm
    new_y = None
    if a.predicts_y:
        z = (_lin(a.y_y, y) + _lin(a.x_y.lin, _node_stats(newX, m)) + _lin(a.e_y.lin, _edge_stats(Y, m))
             + _lin(a.dist_y.lin, _edge_stats(dist1, m)))          # _node_stats sums every row: the unmasked-newX quirk (:202)
        new_y = _lin(a.y_out[2], torch.relu(_lin(a.y_out[0], z)))
    return Xout, Eout, new_y


def train_forward_3d(model, X, E, y, pos, node_mask, with_radius=True, seed=None, p_layer=None):
    """Graph3DTransformer.forward / EquivariantGraph3DTransformer.forward in training mode (pos_transformer_model.py:433-493
    with the dropout sites of :275-297 and :365-428), differentiable: device tensor ops under autograd over the parameter
    tensors the fused inference engine borrows (as ``train_forward`` for the 2D class).  The layers drop with the
    reference's hard-wired 0.1 (the classes do not pass ``dropout`` on, :400-407).  Dropout sites: in-MLPs 0..7 (E, X or pos,
    y; two each, then the node head's mlp_in_X), layer l: 100 + 16 l + (0..8), out-MLPs 50.. (X, E, radius)."""
    if not X.is_cuda:
        raise _lib.ScggError("the training forward runs on the MI355X only; move the inputs to the GPU")
    p_layer = 0.1 if p_layer is None else p_layer
    p_io = model.dropout if model.dropout_in_and_out else 0.0
    dio, dl = DropoutSites(p_io, seed), DropoutSites(p_layer, seed)
    node_mask = node_mask.bool()
    m = node_mask.to(X.dtype)
    n = X.shape[1]
    offdiag = (1.0 - torch.eye(n, dtype=X.dtype, device=X.device))[None]
    oXc = model.out_dim_X + model.out_dim_charges
    pos = pos.detach()

    def mlp_in(seq, x):        # Linear, ReLU, Dropout, Linear, then ReLU / Dropout in either order (same result for one mask)
        return dio(torch.relu(_lin(seq[3], dio(torch.relu(_lin(seq[0], x))))))

    def mlp_out(seq, x):
        return _lin(seq[3], dio(torch.relu(_lin(seq[0], x))))
    hE = mlp_in(model.mlp_in_E, E)
    hE = (hE + hE.transpose(1, 2)) / 2
    hX = mlp_in(model.mlp_in_X, X) if model.equivariant else mlp_in(model.mlp_in_pos_high_dim, pos)
    hy = mlp_in(model.mlp_in_y, y)
    hX, hE = _mask_xe(hX, hE, m, offdiag)
    hpos = _recentre(pos, m)
    for l, layer in enumerate(model.tf_layers):
        dl.site = 100 + 16 * l
        newX, newE, new_y = _block3d(layer.self_attn, hX, hE, hy, hpos, m, node_mask)
        hX = _ln(layer.normX1, hX + dl(newX))
        hE = _ln(layer.normE1, hE + dl(newE))
        hX = _ln(layer.normX2, hX + dl(_lin(layer.linX2, dl(torch.relu(_lin(layer.linX1, hX))))))
        hE = _ln(layer.normE2, hE + dl(_lin(layer.linE2, dl(torch.relu(_lin(layer.linE1, hE))))))
        hE = 0.5 * (hE + hE.transpose(1, 2))
        if layer.predicts_y:
            hy = _ln(layer.norm_y1, hy + dl(new_y))
            hy = _ln(layer.norm_y2, hy + dl(_lin(layer.lin_y2, dl(torch.relu(_lin(layer.lin_y1, hy))))))
        hX, hE = _mask_xe(hX, hE, m, offdiag)
        hpos = _recentre(hpos, m)
    dio.site = 50
    oX = mlp_out(model.mlp_out_X, hX if model.equivariant else mlp_in(model.mlp_in_X, X)) + X[..., :oXc]
    oE = (mlp_out(model.mlp_out_E, hE) + E[..., : model.out_dim_E]) * offdiag[..., None]
    oE = 0.5 * (oE + oE.transpose(1, 2))
    oX, oE = _mask_xe(oX, oE, m, offdiag)
    rad = None
    if with_radius:
        rad = mlp_out(model.mlp_out_radius, hE) * offdiag[..., None]
        rad = 0.5 * (rad + rad.transpose(1, 2)) * (m[:, :, None] * m[:, None, :])[..., None]
    return PlaceHolder(pos=_recentre(pos, m), X=oX[..., : model.out_dim_X], charges=oX[..., model.out_dim_X:], E=oE, y=None,
                       node_mask=node_mask, radius=rad)


# ------------------------------------------------------------------------------------------------- the step
def training_step(model, noise_model, extra_features, dense_data, true_E_labels, loss_fn=None, t_int=None, rng=None,
                  dropout_seed=None):
    """diffusion_model_discrete_3d.py:215-239: z_t = apply_noise(x_0), extra features, train-mode forward, TrainLoss.
    Returns (loss, pred, z_t); the caller does ``loss.backward()`` and the optimizer step (Lightning does in the reference).
    ``dense_data``: PlaceHolder of the clean batch (one-hot X / E, node_mask, optional pos); ``true_E_labels`` its integer
    edge classes [B,n,n]."""
    loss_fn = loss_fn or TrainLoss()
    z_t = noise_model.apply_noise(dense_data, t_int=t_int, rng=rng)
    extra = extra_features(z_t)
    X = torch.cat((z_t.X, z_t.charges, extra.X), dim=2).float()
    E = torch.cat((z_t.E, extra.E), dim=3).float()
    y = torch.hstack((z_t.y, extra.y, z_t.t)).float()
    true_X = dense_data.X.argmax(dim=-1)
    if hasattr(model, "mlp_in_pos_high_dim"):      # Graph3DTransformer / EquivariantGraph3DTransformer: the classes the
        # reference's driver builds (diffusion_model_discrete_3d.py:111-132); their loss carries the radius term (:223)
        true_radius = getattr(dense_data, "radius", None)
        pred = train_forward_3d(model, X, E, y, z_t.pos, z_t.node_mask, with_radius=true_radius is not None, seed=dropout_seed)
        loss = loss_fn(pred, true_X, true_E_labels, z_t.node_mask, true_radius=true_radius)
    else:
        pred = train_forward(model, X, E, y, z_t.node_mask, seed=dropout_seed)
        loss = loss_fn(pred, true_X, true_E_labels, z_t.node_mask)
    return loss, pred, z_t


# ------------------------------------------------------------------------------------------------- data-parallel gradients
class GradientBuckets:
    """Average the gradients of ``params`` over the ranks of ``group``: parameters in reverse registration order (the order
    backward produces them) are laid out in flat buckets of about ``bucket_mb`` and every bucket is one ``all_reduce``
    (RCCL over xGMI on the GPUs; the ring is per-link bound, so few large messages -- the default bucket holds the whole
    4.6 M-parameter ConStruct model, 18 MB).

    The flat buffers are PERSISTENT and every ``p.grad`` is a view into its bucket (``attach()``, done lazily): backward
    accumulates straight into the buffer, the all_reduce runs in place, nothing is packed or unpacked per step.  Use
    ``zero_grad()`` of this object (or ``optimizer.zero_grad(set_to_none=False)``) between steps; a ``set_to_none`` clears
    the views and they are re-attached (with one copy) on the next ``finish()``.

    Parameters that receive no gradient on ANY rank (the reference runs DDP with find_unused_parameters: ``pre_softmax``,
    the last layer's y modules, ...) keep ``grad = None`` after ``finish()``, as under the reference's DDP, so AdamW gives them
    neither weight decay nor state: a per-parameter "touched" flag is reduced with the last bucket.  ``hook=True`` starts a
    bucket's all_reduce from the autograd hook of its last-arriving parameter, overlapped with the rest of backward;
    buckets are always LAUNCHED in index order (a bucket whose turn has not come waits for its predecessors), so every rank
    issues the same sequence of collectives even if their sets of unused parameters differ."""

    def __init__(self, params, group=None, bucket_mb=25.0, hook=False):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.params = [p for p in params if p.requires_grad][::-1]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_mb * 2 ** 20:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._views = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._touched = None                     # [n_params] flags, reduced with MAX behind the last bucket
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._where = {id(p): (bi, k) for bi, b in enumerate(self.buckets) for k, p in enumerate(b)}
        self._seen = set()                       # parameters whose gradient arrived in this backward
        self._hook, self._finishing = bool(hook), False
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._arrived)

    def world(self):
        return self.dist.get_world_size(self.group) if self.dist.is_initialized() else 1

    def attach(self):
        """Allocate the flat buffers (once) and make every existing ``p.grad`` a view into them."""
        for bi, b in enumerate(self.buckets):
            if self._flat[bi] is None:
                self._flat[bi] = torch.zeros(sum(p.numel() for p in b), device=b[0].device)
                off, views = 0, []
                for p in b:
                    views.append(self._flat[bi][off: off + p.numel()].view_as(p))
                    off += p.numel()
                self._views[bi] = views
            for p, v in zip(b, self._views[bi]):
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v
        if self._touched is None:
            self._touched = torch.zeros(len(self.params), device=self.params[0].device)

    def zero_grad(self):
        """Zero the buffers and point every parameter's ``grad`` at its view: backward then accumulates in place."""
        self.attach()
        for bi, b in enumerate(self.buckets):
            self._flat[bi].zero_()
            for p, v in zip(b, self._views[bi]):
                p.grad = v
        self._seen.clear()

    def _launch_in_order(self):
        for bi in range(len(self.buckets)):
            if self._launched[bi]:
                continue
            if self._pending[bi] > 0 and not self._finishing:
                return
            self._launched[bi] = True
            if self.world() > 1:
                self._work[bi] = self.dist.all_reduce(self._flat[bi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _arrived(self, p):
        self._seen.add(id(p))
        bi, k = self._where[id(p)]
        if not self._hook or self._flat[bi] is None or p.grad.data_ptr() != self._views[bi][k].data_ptr():
            return                   # not overlapping, or not attached yet (first step / after set_to_none): finish() does it
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch_in_order()

    def finish(self):
        """Call after ``loss.backward()``: attaches / launches what the hooks have not, waits, scales by 1 / world, and
        leaves ``grad = None`` on the parameters no rank produced a gradient for."""
        w = self.world()
        had = [id(p) in self._seen for p in self.params]
        self.attach()
        for bi, b in enumerate(self.buckets):          # no gradient on this rank: zeros in the sum
            for p, v in zip(b, self._views[bi]):
                if p.grad is None:
                    v.zero_()
        self._touched.copy_(torch.tensor([1.0 if h else 0.0 for h in had]))
        self._finishing = True
        self._launch_in_order()
        self._finishing = False
        twork = None
        if w > 1:
            twork = self.dist.all_reduce(self._touched, op=self.dist.ReduceOp.MAX, group=self.group, async_op=True)
        for bi in range(len(self.buckets)):
            if self._work[bi] is not None:
                self._work[bi].wait()
            if w > 1:
                self._flat[bi].div_(w)
        if twork is not None:
            twork.wait()
        touched = self._touched.cpu().tolist()
        for bi, b in enumerate(self.buckets):
            for p, v in zip(b, self._views[bi]):
                p.grad = v if touched[self._index[id(p)]] > 0 else None
            self._work[bi] = None
            self._launched[bi] = False
        self._pending = [len(b) for b in self.buckets]
        self._seen.clear()
