"""Oracle: noise schedules, transition matrices, reverse-step posterior and categorical sampling.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/ConStruct/diffusion/diffusion_utils.py
  cosine_beta_schedule_discrete :56-87, linear_beta_schedule :90-98,
  sample_discrete_features :200-285, compute_batched_over0_posterior_distribution :315-343
and /root/reference/src/ConStruct/diffusion/noise_model.py
  NoiseModel.__init__ :35-57, get_Qt :101-125, get_Qt_bar :127-154,
  sample_zs_from_zt_and_pred :342-447, AbsorbingEdgesTransition :496-552.

Sampling is defined as a deterministic function of (probabilities, uniforms): the reference's
``torch.multinomial`` offers no uniform input, so the golden harness replaces it by
``inverse_cdf_draws`` below and everything around it is the reference's code.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MAPPING = ["x", "c", "e", "y", "e_small", "e_mid", "e_large"]


# ----------------------------------------------------------------------------- schedules
def cosine_betas(T: int, nu, s: float = 0.008) -> np.ndarray:
    """diffusion_utils.py:56-87 -> betas[T+1, len(nu)] (float64)."""
    steps = T + 2
    x = np.linspace(0, steps, steps)[None, :]
    nu = np.asarray(nu, dtype=np.float64)[:, None]
    ac = np.cos(0.5 * np.pi * (((x / steps) ** nu) + s) / (1 + s)) ** 2
    ac = ac / ac[:, :1]
    alphas = ac[:, 1:] / ac[:, :-1]
    return (1.0 - alphas).T.copy()


def linear_betas(T: int, ncol: int) -> np.ndarray:
    """diffusion_utils.py:90-98 -> beta_t = 1/(T - t + 1), t = 0..T, tiled over ncol columns."""
    t = np.linspace(0, T, T + 1)
    b = 1.0 / (T - t + 1.0)
    return np.tile(b[:, None], (1, ncol))


class Schedule:
    """Tables of NoiseModel.__init__ (:38-57) and AbsorbingEdgesTransition.__init__ (:504-512)."""

    def __init__(self, T: int, transition: str, nu: dict):
        self.T = T
        self.transition = transition
        nu_arr = [nu[k] for k in MAPPING]
        if transition in ("uniform", "marginal", "planar", "absorbing_edges", "adding_edges"):
            betas = cosine_betas(T, nu_arr)
        elif transition == "absorbing":
            betas = linear_betas(T, len(nu_arr))
        else:
            raise NotImplementedError(transition)
        self.betas = torch.from_numpy(betas)                                   # float64 [T+1,7]
        alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(alphas), dim=0))
        if transition == "absorbing_edges":
            ba = linear_betas(T, 7)
            ba[:, 4] = ba[:, 4] ** 0.5
            ba[:, 5] = ba[:, 5] ** 0.75
            self.betas_abs = torch.from_numpy(ba)                              # unclamped (quirk 10)
            self.alphas_abs_bar = torch.cumprod(1 - torch.clamp(self.betas_abs, min=0, max=0.9999), dim=0)

    def beta(self, t_int, key):
        return self.betas[t_int.long().cpu()][..., MAPPING.index(key)].float()

    def alpha_bar(self, t_int, key):
        return self.alphas_bar[t_int.long().cpu()][..., MAPPING.index(key)].float()

    def beta_abs(self, t_int, key):
        return self.betas_abs[t_int.long().cpu()][..., MAPPING.index(key)].float()

    def alpha_abs_bar(self, t_int, key):
        return self.alphas_abs_bar[t_int.long().cpu()][..., MAPPING.index(key)].float()


class Transition:
    """Qt / Qt_bar for the X and E branches (charges are zero-width for CoW/ATM)."""

    def __init__(self, sched: Schedule, x_marginals, e_marginals):
        self.s = sched
        self.xm = torch.as_tensor(x_marginals, dtype=torch.float32)
        em = torch.as_tensor(e_marginals, dtype=torch.float32)
        if sched.transition == "absorbing_edges":                              # :500-501
            em = torch.zeros_like(em)
            em[0] = 1
        self.em = em
        self.Cx, self.Ce = self.xm.numel(), self.em.numel()
        self.Px = self.xm[None, None, :].expand(1, self.Cx, self.Cx)
        self.Pe = self.em[None, None, :].expand(1, self.Ce, self.Ce)

    def Qt(self, t_int):
        """one-step matrices t-1 -> t, [bs,C,C] (:101-125, :524-537)."""
        bx = self.s.beta(t_int, "x")[:, :, None]
        qx = bx * self.Px + (1 - bx) * torch.eye(self.Cx)[None]
        if self.s.transition == "absorbing_edges":
            be = self.s.beta_abs(t_int, "e")[:, :, None]
        else:
            be = self.s.beta(t_int, "e")[:, :, None]
        qe = be * self.Pe + (1 - be) * torch.eye(self.Ce)[None]
        return qx, qe

    def Qt_bar(self, t_int):
        """t-step matrices 0 -> t (:127-154, :539-552)."""
        ax = self.s.alpha_bar(t_int, "x")[:, :, None]
        qx = ax * torch.eye(self.Cx)[None] + (1 - ax) * self.Px
        if self.s.transition == "absorbing_edges":
            ae = self.s.alpha_abs_bar(t_int, "e")[:, :, None]
        else:
            ae = self.s.alpha_bar(t_int, "e")[:, :, None]
        qe = ae * torch.eye(self.Ce)[None] + (1 - ae) * self.Pe
        return qx, qe


# ----------------------------------------------------------------------------- posterior
def posterior_over_x0(z_t, Qt, Qsb, Qtb):
    """diffusion_utils.py:315-343: out[b,N,x0,s] = (z_t Qt^T)[s] * Qsb[x0,s] / (Qtb z_t^T)[x0],
    zero denominators replaced by 1e-6."""
    zt = z_t.flatten(start_dim=1, end_dim=-2).float()
    left = zt @ Qt.transpose(-1, -2)
    num = left[:, :, None, :] * Qsb[:, None, :, :]
    den = (Qtb @ zt.transpose(-1, -2)).transpose(-1, -2)[..., None].clone()
    den[den == 0] = 1e-6
    return num / den


def _mix(pred_prob, post):
    """noise_model.py:376-399: weight by the x0 prediction, sum over x0, all-zero rows -> 1e-5, normalise."""
    un = (pred_prob[..., None] * post).sum(dim=-2)
    un[un.sum(dim=-1) == 0] = 1e-5
    return un / un.sum(dim=-1, keepdim=True)


def reverse_step_probs(tr: Transition, zt_X, zt_E, pred_X, pred_E, t_int, s_int):
    """prob_X[B,n,Cx], prob_E[B,n,n,Ce] of noise_model.py:349-399 (before sampling)."""
    B, n = zt_X.shape[:2]
    qtb_x, qtb_e = tr.Qt_bar(t_int)
    qsb_x, qsb_e = tr.Qt_bar(s_int)
    qt_x, qt_e = tr.Qt(t_int)
    pX = F.softmax(pred_X, dim=-1)
    pE = F.softmax(pred_E, dim=-1).reshape(B, n * n, -1)
    prob_X = _mix(pX, posterior_over_x0(zt_X, qt_x, qsb_x, qtb_x))
    prob_E = _mix(pE, posterior_over_x0(zt_E, qt_e, qsb_e, qtb_e)).reshape(B, n, n, -1)
    return prob_X, prob_E


# ----------------------------------------------------------------------------- sampling
def inverse_cdf_draws(prob: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Sequential inverse-CDF draws WITHOUT replacement, fp32, class order 0..C-1.

    prob[R,C] float32 (need not be normalised), u[R,k] float32 in [0,1) -> idx[R,k] int64.
    For draw r: total = p_0+...+p_{C-1} (left-to-right fp32 adds), thr = u_r*total,
    pick the first c with thr < p_0+...+p_c; if none (rounding) pick the last class with
    remaining mass; if no mass remains pick the lowest-index unpicked class.  The picked
    class's mass is then set to 0.  (SURVEY.md section 8c "Top-k rule".)
    """
    p = np.array(prob, dtype=np.float32, copy=True)
    u = np.asarray(u, dtype=np.float32)
    R, C = p.shape
    k = u.shape[1]
    picked = np.zeros((R, C), dtype=bool)
    out = np.zeros((R, k), dtype=np.int64)
    rows = np.arange(R)
    for r in range(k):
        total = np.zeros(R, dtype=np.float32)
        for c in range(C):
            total = (total + p[:, c]).astype(np.float32)
        thr = (u[:, r] * total).astype(np.float32)
        acc = np.zeros(R, dtype=np.float32)
        sel = np.full(R, -1, dtype=np.int64)
        last_pos = np.full(R, -1, dtype=np.int64)
        for c in range(C):
            acc = (acc + p[:, c]).astype(np.float32)
            hit = (sel < 0) & (thr < acc)
            sel[hit] = c
            last_pos[p[:, c] > 0] = c
        need = sel < 0
        sel[need] = last_pos[need]
        need = sel < 0
        if need.any():
            first_unpicked = np.argmax(~picked, axis=1)
            sel[need] = first_unpicked[need]
        out[:, r] = sel
        p[rows, sel] = 0
        picked[rows, sel] = True
    return out


def sample_edges(prob_E: torch.Tensor, node_mask: torch.Tensor, u0, uk=None):
    """E branch of sample_discrete_features (diffusion_utils.py:228-276).

    prob_E[B,n,n,C]; u0[B,n,n] uniforms of the single draw; uk[B,n,n,top_k] uniforms of the
    top-k draw (None -> no top_k).  Returns (E[B,n,n] int64, extra_info[B,n,n,top_k] or None),
    BEFORE the final PlaceHolder.mask()."""
    B, n = node_mask.shape
    C = prob_E.shape[-1]
    p = prob_E.clone()
    inv = ~(node_mask[:, None, :] & node_mask[:, :, None])
    p[inv] = 1.0 / C                                                          # :233
    p[torch.eye(n, dtype=torch.bool)[None].expand(B, -1, -1)] = 1.0 / C       # :234
    p = p.reshape(B * n * n, C).numpy()
    e = inverse_cdf_draws(p, np.asarray(u0, np.float32).reshape(-1, 1)).reshape(B, n, n)
    e = torch.from_numpy(e)
    if uk is not None:
        k = uk.shape[-1]
        multi = torch.from_numpy(inverse_cdf_draws(p, np.asarray(uk, np.float32).reshape(-1, k)).reshape(B, n, n, k))
        multi[..., 0] = e                                                     # :249
        multi = multi.permute(0, 3, 1, 2)
        up = torch.triu(multi, diagonal=1)
        multi = (up + up.transpose(-1, -2)).permute(0, 2, 3, 1)               # :263-269
        return multi[..., 0].contiguous(), multi.contiguous()
    up = torch.triu(e, diagonal=1)
    return up + up.transpose(1, 2), None


def sample_nodes(prob_X: torch.Tensor, node_mask: torch.Tensor, ux):
    """X branch of sample_discrete_features (:207-217)."""
    B, n = node_mask.shape
    p = prob_X.clone()
    p[~node_mask] = 1.0 / p.shape[-1]
    x = inverse_cdf_draws(p.reshape(B * n, -1).numpy(), np.asarray(ux, np.float32).reshape(-1, 1))
    return torch.from_numpy(x.reshape(B, n))


def reverse_step(tr: Transition, zt_X, zt_E, pred_X, pred_E, node_mask, t_int, s_int, ux, u0, uk):
    """Full sample_zs_from_zt_and_pred for the X/E branches (noise_model.py:342-447).

    Returns dict with prob_X, prob_E, X_s (one-hot float, masked), E_s (one-hot float, masked),
    labels_E (int64, -1 where the one-hot row is the zero vector), extra_info (masked, int64)."""
    B, n = node_mask.shape
    prob_X, prob_E = reverse_step_probs(tr, zt_X, zt_E, pred_X, pred_E, t_int, s_int)
    xs = sample_nodes(prob_X, node_mask, ux)
    es, extra = sample_edges(prob_E, node_mask, u0, uk)
    m = node_mask
    pm = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    X_s = F.one_hot(xs, tr.Cx).float() * m[:, :, None]
    E_s = F.one_hot(es, tr.Ce).float() * pm[..., None]
    labels = torch.where(pm, es, torch.full_like(es, -1))
    if extra is not None:
        extra = extra * pm[..., None]
    return dict(prob_X=prob_X, prob_E=prob_E, X_s=X_s, E_s=E_s, labels_E=labels, extra_info=extra)


# ----------------------------------------------------------------------------- Philox4x32-10
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Counter-based Philox4x32-10 (Salmon et al. 2011).  ctr[...,4] uint32, key[2] uint32 -> [...,4] uint32."""
    c = [ctr[..., i].astype(np.uint32) for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * _M0
            p1 = c[2].astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return np.stack(c, axis=-1)


def u01(x: np.ndarray) -> np.ndarray:
    """uint32 -> float32 uniform in [0,1) from the top 24 bits."""
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


STREAM_E, STREAM_X = 0, 1


def philox_edge_uniforms(seed: int, step: int, graph_ids, n: int, top_k: int):
    """The uniform stream the HIP reverse-step kernel consumes ('RNG').

    counter = ((i<<16)|j, graph_id, step, stream*16 + block); key = (seed & 2^32-1, seed >> 32).
    block 0 -> [u0, uk0, uk1, uk2], block 1 -> [uk3, ...].  Returns u0[B,n,n], uk[B,n,n,top_k]."""
    g = np.asarray(graph_ids, dtype=np.uint32)
    B = g.shape[0]
    i, j = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32), indexing="ij")
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    nblk = (1 + top_k + 3) // 4
    outs = []
    for blk in range(nblk):
        ctr = np.zeros((B, n, n, 4), dtype=np.uint32)
        ctr[..., 0] = ((i << np.uint32(16)) | j)[None]
        ctr[..., 1] = g[:, None, None]
        ctr[..., 2] = np.uint32(step)
        ctr[..., 3] = np.uint32(STREAM_E * 16 + blk)
        outs.append(u01(philox4x32_10(ctr, key)))
    u = np.concatenate(outs, axis=-1)
    return u[..., 0], u[..., 1:1 + top_k]
