"""Drop-in for ``src/ConStruct/diffusion/noise_model.py`` of the reference (NoiseModel :11, DiscreteUniformTransition
:450, MarginalTransition :464, AbsorbingTransition :478, AbsorbingEdgesTransition :496): same attributes
(``T, X_classes, E_classes, ...``) and methods (``get_Qt, get_Qt_bar, apply_noise, get_limit_dist,
sample_limit_dist, sample_zs_from_zt_and_pred``).

Host side here: the beta / alpha_bar tables (float64, built once) and tiny C x C matrices.  Device side: the
O(n^2) posterior + sampling runs in libscgg.so (``scgg_reverse_step_edges`` / ``scgg_sample_edges_from_probs``).
Randomness is explicit: every sampling call takes ``rng=(seed, step, graph_id_base)`` (Philox counter stream) or
explicit uniform tensors; nothing consumes the global torch RNG on the edge path.

Differences from the reference that a caller can see:
  * the node-coordinate DDPM the reference hard-loads in ``__init__`` (:62-79, checkpoint not in its repo) is an
    optional ``node_coord_model`` callable; the default draws N(0,1) positions;
  * the returned PlaceHolder also carries the int8 state (``_labels``, ``_cands``) so the next step skips the
    one-hot round trip.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .placeholder import PlaceHolder

MAPPING = ("x", "c", "e", "y", "e_small", "e_mid", "e_large")
POS_START_STEP = 0x70000000      # Philox step word of the start coordinates: no chain or SiT iteration uses it


def _cfg_get(cfg, *path, default=None):
    cur = cfg
    for p in path:
        if cur is None:
            return default
        cur = cur.get(p, None) if isinstance(cur, dict) else getattr(cur, p, None)
    return default if cur is None else cur


def _cosine_betas(T, nu, s=0.008):
    steps = T + 2
    u = np.linspace(0, steps, steps)[None, :] / steps
    a = np.cos(0.5 * np.pi * ((u ** np.asarray(nu, np.float64)[:, None]) + s) / (1 + s)) ** 2
    a = a / a[:, :1]
    return (1.0 - a[:, 1:] / a[:, :-1]).T.copy()


def _linear_betas(T, ncol):
    t = np.linspace(0, T, T + 1)
    return np.tile((1.0 / (T - t + 1.0))[:, None], (1, ncol))


class Rng:
    """Counter-stream handle for the sampling kernels: Philox key = seed, counter = (i, j, graph id, step)."""

    def __init__(self, seed=0, step=0, graph_id_base=0, u0=None, uk=None):
        self.seed, self.step, self.graph_id_base, self.u0, self.uk = int(seed), int(step), int(graph_id_base), u0, uk

    def c_struct(self):
        return _lib.Rng(_lib.ptr(self.u0).value if self.u0 is not None else None,
                        _lib.ptr(self.uk).value if self.uk is not None else None,
                        self.seed & 0xFFFFFFFFFFFFFFFF, self.step & 0xFFFFFFFF, self.graph_id_base & 0xFFFFFFFF)


class NoiseModel:
    def __init__(self, cfg, node_coord_model=None):
        self.mapping = list(MAPPING)
        self.inverse_mapping = {m: i for i, m in enumerate(self.mapping)}
        self.cfg = cfg
        nu = _cfg_get(cfg, "model", "nu")
        nu = nu if isinstance(nu, dict) else vars(nu)
        self.nu_arr = [nu[m] for m in self.mapping]
        self.transition = _cfg_get(cfg, "model", "transition")
        self.T = self.timesteps = int(_cfg_get(cfg, "model", "diffusion_steps"))
        self.top_k = _cfg_get(cfg, "model", "top_k")
        if self.transition in ("uniform", "marginal", "planar", "absorbing_edges", "adding_edges"):
            betas = _cosine_betas(self.T, self.nu_arr)
        elif self.transition == "absorbing":
            betas = _linear_betas(self.T, len(self.nu_arr))
        else:
            raise NotImplementedError(self.transition)
        self._betas = torch.from_numpy(betas)
        self._alphas = 1 - torch.clamp(self._betas, min=0, max=0.9999)
        self._log_alpha_bar = torch.cumsum(torch.log(self._alphas), dim=0)
        self._alphas_bar = torch.exp(self._log_alpha_bar)
        self.node_coord_model = node_coord_model
        self.X_classes = self.E_classes = self.charges_classes = self.y_classes = None
        self.X_marginals = self.E_marginals = self.charges_marginals = self.y_marginals = None
        self._dev_tables = {}
        self._dev_cache = {}

    def _on(self, t, device):
        """A host table on ``device``, copied once per (table, device).  A copy from pageable host memory makes the host wait
        for everything queued on the stream: six of them per apply_noise kept the training loop from ever running ahead of
        the device."""
        k = (id(t), str(device))
        hit = self._dev_cache.get(k)
        if hit is None or hit[0] is not t:
            hit = self._dev_cache[k] = (t, t.to(device))
        return hit[1]

    # ---- tables ------------------------------------------------------------------------------------------
    def _get(self, series, t_normalized=None, t_int=None, key=None):
        assert (t_normalized is None) != (t_int is None)
        if t_int is None:
            t_int = torch.round(t_normalized * self.T)
        att = self._on(series, t_int.device)[t_int.long()]
        return att.float() if key is None else att[..., self.inverse_mapping[key]].float()

    def get_beta(self, t_normalized=None, t_int=None, key=None):
        return self._get(self._betas, t_normalized, t_int, key)

    def get_alpha_bar(self, t_normalized=None, t_int=None, key=None):
        return self._get(self._alphas_bar, t_normalized, t_int, key)

    def _edge_tables(self):
        """fp32 [T+1] beta / alpha_bar of the E branch (what get_Qt / get_Qt_bar index with t_int)."""
        return self._betas[:, 2].float(), self._alphas_bar[:, 2].float()

    def _device_tables(self, device):
        k = str(device)
        if k not in self._dev_tables:
            beta, abar = self._edge_tables()
            self._dev_tables[k] = (beta.to(device).contiguous(), abar.to(device).contiguous(),
                                   self.E_marginals.float().to(device).contiguous())
        return self._dev_tables[k]

    def _P(self, marg, device):
        c = marg.numel()
        return self._on(marg, device).float()[None, None, :].expand(1, c, c)

    def get_Qt(self, t_int):
        """one-step transition matrices (reference :101-125): beta_t * 1 m^T + (1 - beta_t) * I."""
        dev = t_int.device
        out = {}
        for name, key, marg in (("X", "x", self.X_marginals), ("charges", "c", self.charges_marginals),
                                ("E", "e", self.E_marginals), ("y", "y", self.y_marginals)):
            b = self._one_step_beta(t_int, key)[:, :, None] if name == "E" else self.get_beta(t_int=t_int, key=key)[:, :, None]
            c = marg.numel()
            out[name] = b * self._P(marg, dev) + (1 - b) * torch.eye(c, device=dev)[None]
        return PlaceHolder(X=out["X"], charges=out["charges"], E=out["E"], y=out["y"])

    def get_Qt_bar(self, t_int):
        """t-step transition matrices (reference :127-154): abar_t * I + (1 - abar_t) * 1 m^T."""
        dev = t_int.device
        out = {}
        for name, key, marg in (("X", "x", self.X_marginals), ("charges", "c", self.charges_marginals),
                                ("E", "e", self.E_marginals), ("y", "y", self.y_marginals)):
            a = self._edge_alpha_bar(t_int)[:, :, None] if name == "E" else self.get_alpha_bar(t_int=t_int, key=key)[:, :, None]
            c = marg.numel()
            out[name] = a * torch.eye(c, device=dev)[None] + (1 - a) * self._P(marg, dev)
        return PlaceHolder(X=out["X"], charges=out["charges"], E=out["E"], y=out["y"])

    def _one_step_beta(self, t_int, key):
        return self.get_beta(t_int=t_int, key=key)

    def _edge_alpha_bar(self, t_int):
        return self.get_alpha_bar(t_int=t_int, key="e")

    def get_limit_dist(self):
        """reference :235-246."""
        def norm(v):
            v = v + 1e-7
            return v / v.sum()
        return PlaceHolder(X=norm(self.X_marginals), E=norm(self.E_marginals), charges=norm(self.charges_marginals), y=None)

    # ---- sampling helpers ------------------------------------------------------------------------------------
    @staticmethod
    def _labels_of(ph, C_e):
        """int8 edge state of a PlaceHolder: cached by our own sampler, else derived from the one-hot E."""
        lab = getattr(ph, "_labels", None)
        if lab is not None:
            return lab
        E = ph.E.float().contiguous()
        P = E.numel() // C_e
        lab = torch.empty(E.shape[:-1], dtype=torch.int8, device=E.device)
        _lib.check(_lib.lib().scgg_onehot_to_labels(_lib.ptr(E), P, C_e, _lib.ptr(lab), _lib.stream_ptr()), "onehot_to_labels")
        return lab

    def _to_placeholder(self, labels, cands, node_mask, t_int, pos, y_rows, radius=None, with_onehot=True):
        B, n = node_mask.shape
        dev = node_mask.device
        C_e = self.E_classes
        top_k = cands.shape[-1] if cands is not None else 0
        E = extra = None
        if with_onehot:
            E = torch.empty(B, n, n, C_e, device=dev)
            extra = torch.empty(B, n, n, top_k, dtype=torch.int64, device=dev) if top_k else None
            _lib.check(_lib.lib().scgg_labels_to_onehot(_lib.ptr(labels), _lib.ptr(cands), B * n * n, C_e, top_k, _lib.ptr(E),
                                                        _lib.ptr(extra), _lib.stream_ptr()), "labels_to_onehot")
        X = node_mask[..., None].float() if self.X_classes == 1 else None
        ph = PlaceHolder(X=X, charges=torch.zeros(B, n, 0, device=dev), E=E, y=torch.zeros(y_rows, 0, device=dev),
                         t_int=t_int, t=t_int / self.T, node_mask=node_mask, pos=pos, radius=radius, extra_info=extra)
        ph._labels, ph._cands = labels, cands
        return ph

    @staticmethod
    def _fresh_rng(step=0):
        """A stream nobody has used yet, seeded from torch's global generator (which the reference consumes directly):
        two calls without an explicit ``rng`` never draw the same uniforms, and ``torch.manual_seed`` reproduces them."""
        return Rng(seed=int(torch.randint(0, 2 ** 31 - 1, (1,))), step=step)

    def sample_limit_dist(self, node_mask, generator=None, with_onehot=True, rng: Rng = None):
        """reference :248-340.  For absorbing / absorbing_edges the limit is the all-"no edge" graph (:253-261); for the
        other transitions z_T is drawn from the marginals with ``rng`` (seed, graph_id_base; the step word is T + 1, which
        no reverse step uses), default: a fresh stream from torch's global generator."""
        B, n = node_mask.shape
        dev = node_mask.device
        if self.X_classes != 1 or self.charges_classes != 0:
            raise NotImplementedError("node / charge classes beyond the CoW / ATM setting (1 / 0) are not built yet")
        top_k = self.top_k if self.top_k is not None else 1
        live = (node_mask[:, :, None] & node_mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool, device=dev)[None]
        if bool((self.E_marginals[1:] == 0).all()):
            labels = torch.where(live, 0, -1).to(torch.int8).contiguous()
            cands = torch.zeros(B, n, n, top_k, dtype=torch.int8, device=dev) if top_k > 1 else None
        else:
            prob = self._on(self.E_marginals, dev).float().expand(B, n, n, -1).contiguous()
            zr = self._fresh_rng() if rng is None else rng
            zr = Rng(seed=zr.seed, step=self.T + 1, graph_id_base=zr.graph_id_base)
            labels, cands = self.sample_edges_from_probs(prob, node_mask, top_k if top_k > 1 else 0, zr)
        t_int = torch.full((B, 1), self.T, dtype=torch.long, device=dev)
        # Start coordinates (:313-325).  With ``rng`` they are Philox normals keyed by (seed, GLOBAL graph id, node): graph g
        # starts from the same point whatever shard or batch slot samples it (sharding.py's invariant), and the coordinate
        # model's per-step normals are keyed the same way.  Without ``rng``: torch.randn from ``generator`` / the global one.
        if rng is not None:
            pos = torch.empty(B, n, 3, device=dev)
            _lib.check(_lib.lib().scgg_philox_normal_rows(_lib.ptr(pos), B, n, rng.seed, rng.graph_id_base, POS_START_STEP,
                                                          _lib.stream_ptr()), "philox_normal_rows")
        else:
            pos = torch.randn(B, n, 3, device=dev, generator=generator)
        if self.node_coord_model is not None:      # reference :323-325: node_coord_model.sample(pos=start_pos_shape, node_mask=...)
            sample = getattr(self.node_coord_model, "sample", self.node_coord_model)
            if rng is not None and getattr(self.node_coord_model, "takes_rng", False):
                pos = sample(pos, node_mask, seed=rng.seed, graph_id_base=rng.graph_id_base)
            else:
                pos = sample(pos, node_mask)
        raw = pos * node_mask[..., None]
        radius = torch.zeros(B, n, n, 1, device=dev)
        ph = self._to_placeholder(labels, cands, node_mask, t_int, self.pos_after_masks(raw, node_mask, 1), B, radius,
                                  with_onehot)
        ph.t = torch.ones(B, 1, device=dev)
        ph._pos_raw = raw
        return ph

    @staticmethod
    def pos_after_masks(raw, node_mask, t: int):
        """``pos`` after ``t`` applications of PlaceHolder.mask (utils.py:160-164: zero the padded slots, subtract the mean
        over ALL n slots), in closed form.  The reference applies it once in sample_limit_dist (:340) and once per reverse
        step (:446), so the frame of a graph with k < n nodes keeps drifting: with S = sum of the valid rows and
        r = 1 - k/n the valid rows are x_i - (S/k)(1 - r^t) and the padded slots -(S/n) r^(t-1); k = n: centred at once.
        ``raw``: positions with the padded slots already zero."""
        k = node_mask.sum(dim=1, keepdim=True).to(raw.dtype)[..., None]            # [B,1,1]
        n = float(node_mask.shape[1])
        S = raw.sum(dim=1, keepdim=True)
        r = 1.0 - k / n
        ck = torch.where(k > 0, S / k.clamp(min=1.0), torch.zeros_like(S)) * (1.0 - r ** t)
        pad = -(S / n) * r ** (t - 1)
        return torch.where(node_mask[..., None], raw - ck, pad.expand_as(raw))

    def sample_edges_from_probs(self, prob_E, node_mask, top_k, rng: Rng):
        """E branch of sample_discrete_features (diffusion_utils.py:228-276) on the GPU -> (labels int8, cands int8)."""
        B, n, _, C_e = prob_E.shape
        dev = prob_E.device
        labels = torch.empty(B, n, n, dtype=torch.int8, device=dev)
        cands = torch.empty(B, n, n, top_k, dtype=torch.int8, device=dev) if top_k else None
        r = rng.c_struct()
        _lib.check(_lib.lib().scgg_sample_edges_from_probs(
            _lib.ptr(prob_E.float().contiguous()), _lib.ptr(node_mask.contiguous().view(torch.uint8)), B, n, C_e, top_k,
            C.byref(r), _lib.ptr(labels), _lib.ptr(cands), _lib.stream_ptr()), "sample_edges_from_probs")
        return labels, cands

    def apply_noise(self, dense_data, t_int=None, rng: Rng = None):
        """reference :187-233: z_t ~ q(z_t | x_0) for a per-graph t ~ U{1..T}."""
        dev = dense_data.X.device
        B = dense_data.X.size(0)
        if t_int is None:
            t_int = torch.randint(1, self.T + 1, size=(B, 1), device=dev)
        if self.X_classes != 1 or self.charges_classes != 0:
            raise NotImplementedError("node / charge classes beyond the CoW / ATM setting (1 / 0) are not built yet")
        Qtb = self.get_Qt_bar(t_int)
        probE = dense_data.E.float() @ Qtb.E.unsqueeze(1)
        rng = rng or self._fresh_rng()
        labels, _ = self.sample_edges_from_probs(probE, dense_data.node_mask, 0, rng)
        ph = self._to_placeholder(labels, None, dense_data.node_mask, t_int, dense_data.pos, B, dense_data.radius)
        ph.y = dense_data.y
        ph.t = t_int.float() / self.T
        if ph.pos is not None:
            ph.pos = ph.pos * dense_data.node_mask[..., None]
            ph.pos = ph.pos - ph.pos.mean(dim=1, keepdim=True)
        return ph

    def reverse_step_state(self, logits_E, labels_t, node_mask, t_int32, s_int32, rng: Rng, labels_out=None,
                           cands_out=None, prob_out=None):
        """The device call behind sample_zs_from_zt_and_pred on the int8 state (no PlaceHolder, no allocation when the
        output buffers are given): labels_t [B,n,n] int8 -> (labels_s, cands_s)."""
        B, n = node_mask.shape
        dev = node_mask.device
        top_k = self.top_k or 0
        beta, abar, marg = self._device_tables(dev)
        if labels_out is None:
            labels_out = torch.empty(B, n, n, dtype=torch.int8, device=dev)
        if cands_out is None and top_k:
            cands_out = torch.empty(B, n, n, top_k, dtype=torch.int8, device=dev)
        r = rng.c_struct()
        _lib.check(_lib.lib().scgg_reverse_step_edges(
            _lib.ptr(logits_E), _lib.ptr(labels_t), _lib.ptr(node_mask.view(torch.uint8)), _lib.ptr(t_int32), _lib.ptr(s_int32),
            _lib.ptr(beta), _lib.ptr(abar), _lib.ptr(marg), B, n, self.E_classes, top_k, C.byref(r), _lib.ptr(labels_out),
            _lib.ptr(cands_out), _lib.ptr(prob_out), _lib.stream_ptr()), "reverse_step_edges")
        return labels_out, cands_out

    def sample_zs_from_zt_and_pred(self, z_t, pred, s_int, rng: Rng = None, with_onehot=True):
        """reference :342-447: z_s ~ p(z_s | z_t) given the denoiser output ``pred``."""
        if self.X_classes != 1 or self.charges_classes != 0:
            raise NotImplementedError("node / charge classes beyond the CoW / ATM setting (1 / 0) are not built yet")
        if not self.top_k:
            raise _lib.ScggError("sample_zs_from_zt_and_pred needs cfg.model.top_k (the reference asserts on extra_info, :433)")
        node_mask = z_t.node_mask.contiguous()
        B, n = node_mask.shape
        if rng is None:      # the reference draws from the global torch RNG here: a fresh stream per call
            rng = self._fresh_rng()
        labels_t = self._labels_of(z_t, self.E_classes)
        t32 = z_t.t_int.reshape(B).to(torch.int32).contiguous()
        s32 = s_int.reshape(B).to(torch.int32).contiguous()
        labels_s, cands_s = self.reverse_step_state(pred.E.float().contiguous(), labels_t, node_mask, t32, s32, rng)
        radius = None
        if getattr(pred, "radius", None) is not None:                      # :415-420
            radius = ((labels_s > 0) * pred.radius.squeeze(-1)).unsqueeze(-1)
        pos = z_t.pos
        if pos is not None:                                                # .mask() of the new PlaceHolder (:446)
            pos = pos * node_mask[..., None]
            pos = pos - pos.mean(dim=1, keepdim=True)
        z_s = self._to_placeholder(labels_s, cands_s, node_mask, s_int, pos, z_t.y.shape[0], radius, with_onehot)
        return z_s


class DiscreteUniformTransition(NoiseModel):
    def __init__(self, cfg, output_dims, node_coord_model=None):
        super().__init__(cfg, node_coord_model)
        self.X_classes, self.charges_classes = output_dims.X, output_dims.charges
        self.E_classes, self.y_classes = output_dims.E, output_dims.y
        self.X_marginals = torch.ones(self.X_classes) / self.X_classes
        self.charges_marginals = torch.ones(self.charges_classes) / max(self.charges_classes, 1)
        self.E_marginals = torch.ones(self.E_classes) / self.E_classes
        self.y_marginals = torch.ones(self.y_classes) / max(self.y_classes, 1)


class MarginalTransition(NoiseModel):
    def __init__(self, cfg, x_marginals, e_marginals, charges_marginals, y_classes, node_coord_model=None):
        super().__init__(cfg, node_coord_model)
        self.X_classes, self.E_classes = len(x_marginals), len(e_marginals)
        self.charges_classes, self.y_classes = len(charges_marginals), y_classes
        self.X_marginals = torch.as_tensor(x_marginals, dtype=torch.float32)
        self.E_marginals = torch.as_tensor(e_marginals, dtype=torch.float32)
        self.charges_marginals = torch.as_tensor(charges_marginals, dtype=torch.float32)
        self.y_marginals = torch.ones(y_classes) / max(y_classes, 1)


class AbsorbingTransition(NoiseModel):
    def __init__(self, cfg, output_dims, node_coord_model=None):
        super().__init__(cfg, node_coord_model)
        self.X_classes, self.charges_classes = output_dims.X, output_dims.charges
        self.E_classes, self.y_classes = output_dims.E, output_dims.y
        self.X_marginals = F.one_hot(torch.tensor(0), self.X_classes).float()
        self.E_marginals = F.one_hot(torch.tensor(0), self.E_classes).float()
        self.charges_marginals = torch.zeros(self.charges_classes)
        if self.charges_classes > 0:
            self.charges_marginals[0] = 1
        self.y_marginals = torch.ones(self.y_classes) / max(self.y_classes, 1)


class AbsorbingEdgesTransition(MarginalTransition):
    """Edge-deletion noise (reference :496-552): edges are absorbed into class 0 on the schedule beta_t = 1/(T-t+1);
    nodes keep the marginal transition.  get_beta_abs returns the UNclamped beta (quirk 10)."""

    def __init__(self, cfg, x_marginals, e_marginals, charges_marginals, y_classes, node_coord_model=None):
        super().__init__(cfg, x_marginals, e_marginals, charges_marginals, y_classes, node_coord_model)
        self.E_marginals = F.one_hot(torch.tensor(0), self.E_classes).float()
        ba = _linear_betas(self.T, 7)
        ba[:, 4] = ba[:, 4] ** 0.5
        ba[:, 5] = ba[:, 5] ** 0.75
        self._betas_abs = torch.from_numpy(ba)
        self._alphas_abs = 1 - torch.clamp(self._betas_abs, min=0, max=0.9999)
        self._alphas_abs_bar = torch.cumprod(self._alphas_abs, dim=0)

    def get_beta_abs(self, t_normalized=None, t_int=None, key=None):
        return self._get(self._betas_abs, t_normalized, t_int, key)

    def get_alpha_abs_bar(self, t_normalized=None, t_int=None, key=None):
        return self._get(self._alphas_abs_bar, t_normalized, t_int, key)

    def _edge_tables(self):
        return self._betas_abs[:, 2].float(), self._alphas_abs_bar[:, 2].float()

    def _one_step_beta(self, t_int, key):
        return self.get_beta_abs(t_int=t_int, key=key) if key == "e" else self.get_beta(t_int=t_int, key=key)

    def _edge_alpha_bar(self, t_int):
        return self.get_alpha_abs_bar(t_int=t_int, key="e")
