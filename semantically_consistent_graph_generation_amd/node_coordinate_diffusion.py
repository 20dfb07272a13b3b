"""Drop-in for ``src/bootstrap/node_coordinate_diffusion`` of the reference: the ``SiT`` denoiser
(sit_model.py:157-247; ``SiT_models['SiT-T']``) as a parameter container with the reference's ``state_dict`` names, and
``NodeCoordsDiffusionModel`` (cow_ / atm_diff_nodes_with_3d_checks.py:22-81) with ``.model``, ``.to_cuda_device`` and
``.sample(pos, node_mask)``, which ``NoiseModel.sample_limit_dist`` calls once per sampled batch (noise_model.py:323-325;
here: ``NoiseModel(cfg, node_coord_model=NodeCoordsDiffusionModel(...).sample)``-style hook, see noise_model.py).

The arithmetic runs in libscgg.so (``scgg_sit_forward`` / ``scgg_sit_sample``, csrc/sit.hip); there is no eager fallback.
The reference's checkpoints (``cow_best_coord_model.pth``) are not in its repository: weights are whatever the caller loads.
Parity: the network is checked against outputs of the reference's own module (fixtures h1_sit_*, made with a stand-in for
the two ``timm`` constructors it imports); the sampler loop against ``NodeCoordsDiffusionModel.sample`` run from the
reference's real module with its normal draws recorded (fixtures h2_sit_sample_*)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class _Attention(nn.Module):          # timm.models.vision_transformer.Attention's parameters (qkv with bias, proj)
    def __init__(self, dim):
        super().__init__()
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.attn, self.mlp = _Attention(dim), _Mlp(dim, hidden)
        self.adaLN_modulation = nn.Sequential(nn.Identity(), nn.Linear(dim, 6 * dim))


class _TEmb(nn.Module):
    def __init__(self, dim, freq=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(freq, dim), nn.Identity(), nn.Linear(dim, dim))


class _Final(nn.Module):
    def __init__(self, dim, out):
        super().__init__()
        self.linear = nn.Linear(dim, out)
        self.adaLN_modulation = nn.Sequential(nn.Identity(), nn.Linear(dim, 2 * dim))


class SiT(nn.Module):
    """SiT(in_channels, hidden_size, num_transformer_blocks, num_heads, mlp_ratio) -- sit_model.py:161-187."""

    def __init__(self, in_channels=3, hidden_size=192, num_transformer_blocks=6, num_heads=6, mlp_ratio=4.0, noise_steps=1000,
                 beta_start=1e-4, beta_end=0.02):
        super().__init__()
        self.in_channels, self.hidden_size, self.num_heads = in_channels, hidden_size, num_heads
        self.mlp_hidden = int(hidden_size * mlp_ratio)
        self.noise_steps, self.beta_start, self.beta_end = noise_steps, beta_start, beta_end
        self.pos_embedder = nn.Linear(in_channels, hidden_size)
        self.t_embedder = _TEmb(hidden_size)
        self.blocks = nn.ModuleList([_Block(hidden_size, self.mlp_hidden) for _ in range(num_transformer_blocks)])
        self.final_layer = _Final(hidden_size, in_channels)
        self._engine, self._key, self._versions, self._ws = None, None, None, {}

    def _release(self):
        if self._engine is not None:
            _lib.lib().scgg_sit_destroy(self._engine)
            self._engine = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_engine(self, device):
        L = _lib.lib()
        sd = dict(self.state_dict(keep_vars=True))
        key = (str(device), tuple((k, v.data_ptr()) for k, v in sd.items()))
        versions = tuple(v._version for v in sd.values())
        if self._engine is None or key != self._key:
            self._release()
            for k, v in sd.items():
                if v.device != device or v.dtype != torch.float32 or not v.is_contiguous():
                    raise _lib.ScggError(f"parameter {k} must be contiguous fp32 on {device}")
            cfg = _lib.SitConfig(self.in_channels, self.hidden_size, len(self.blocks), self.num_heads, self.mlp_hidden,
                                 self.noise_steps, self.beta_start, self.beta_end)
            h = L.scgg_sit_create(C.byref(cfg))
            if not h:
                raise _lib.ScggError("scgg_sit_create: " + L.scgg_last_error().decode())
            self._engine = C.c_void_p(h)
            for k, v in sd.items():
                _lib.check(L.scgg_sit_set_param(self._engine, k.encode(), _lib.ptr(v.detach()), v.numel()), k)
            _lib.check(L.scgg_sit_finalize(self._engine, _lib.stream_ptr()), "sit finalize")
            self._key, self._versions = key, versions
        elif versions != self._versions:
            _lib.check(L.scgg_sit_finalize(self._engine, _lib.stream_ptr()), "sit finalize")
            self._versions = versions
        return self._engine

    def _workspace(self, B, n, device):
        k = (B, n, str(device))
        if k not in self._ws:
            nbytes = _lib.lib().scgg_sit_workspace_bytes(self._engine, B, n)
            self._ws = {k: torch.empty(nbytes + 256, dtype=torch.uint8, device=device)}
        ws = self._ws[k]
        off = (-ws.data_ptr()) % 256
        return C.c_void_p(ws.data_ptr() + off), ws.numel() - 256

    @torch.no_grad()
    def forward(self, x, t, padding_mask):
        """x [B,N,3], t [B] timesteps, padding_mask [B,N] bool -> predicted noise [B,N,3] (valid rows; 0 on padded ones)."""
        if not x.is_cuda:
            raise _lib.ScggError("SiT.forward runs on the MI355X only; move the inputs to the GPU")
        B, n = x.shape[0], x.shape[1]
        eng = self._ensure_engine(x.device)
        ws, nb = self._workspace(B, n, x.device)
        x = x.detach().float().contiguous()
        t32 = t.to(torch.int32).contiguous()
        m = padding_mask.to(torch.bool).contiguous()
        out = torch.empty(B, n, 3, device=x.device)
        _lib.check(_lib.lib().scgg_sit_forward(eng, _lib.ptr(x), _lib.ptr(t32), _lib.ptr(m.view(torch.uint8)), B, n, _lib.ptr(out),
                                               ws, nb, _lib.stream_ptr()), "sit_forward")
        return out

    @torch.no_grad()
    def sample(self, x_T, node_mask, steps=-1, noise=None, seed=0, graph_id_base=0):
        B, n = x_T.shape[0], x_T.shape[1]
        eng = self._ensure_engine(x_T.device)
        ws, nb = self._workspace(B, n, x_T.device)
        x_T = x_T.detach().float().contiguous()
        m = node_mask.to(torch.bool).contiguous()
        out = torch.empty(B, n, 3, device=x_T.device)
        nz = None if noise is None else noise.detach().float().contiguous()
        _lib.check(_lib.lib().scgg_sit_sample(eng, _lib.ptr(x_T), _lib.ptr(m.view(torch.uint8)), B, n, steps, _lib.ptr(nz), seed,
                                              graph_id_base, _lib.ptr(out), ws, nb, _lib.stream_ptr()), "sit_sample")
        return out


SiT_models = {"SiT-T": lambda **kw: SiT(num_transformer_blocks=6, hidden_size=192, num_heads=6, **kw),
              "SiT-VT": lambda **kw: SiT(num_transformer_blocks=4, hidden_size=192, num_heads=4, **kw),
              "SiT-S": lambda **kw: SiT(num_transformer_blocks=12, hidden_size=384, num_heads=6, **kw)}


class NodeCoordsDiffusionModel:
    """cow_diff_nodes_with_3d_checks.py:22-81 (sampling side): ``.model`` (SiT-T), ``.to_cuda_device``, ``.sample``."""

    takes_rng = True     # sample() accepts (seed, graph_id_base): NoiseModel.sample_limit_dist hands its Philox key on

    def __init__(self, noise_steps=1000, beta_start=1e-4, beta_end=0.02, coordinate_size=3, device="cuda"):
        self.model = SiT_models["SiT-T"](in_channels=coordinate_size, noise_steps=noise_steps, beta_start=beta_start,
                                         beta_end=beta_end).to(device)
        self.noise_steps, self.beta_start, self.beta_end, self.device = noise_steps, beta_start, beta_end, device

    def to_cuda_device(self, device):
        if not next(self.model.parameters()).is_cuda:
            self.model.to(device)

    def sample(self, pos, node_mask, seed=None, graph_id_base=0):
        """``pos``: standard-normal start coordinates [B,n,3].  The reference ignores the values it is handed and draws
        torch.randn of the same shape again (:66); here the given draw IS x_T (same distribution, and the caller's
        generator decides it).  The per-step normals come from Philox keyed by (``seed``, ``graph_id_base`` + b, node, i);
        ``NoiseModel.sample_limit_dist`` passes the chain's own key, so a graph's coordinates depend on its global id only,
        not on the shard.  Called bare (the reference's signature), the seed is drawn from torch's global generator, which
        the reference consumes here too: ``torch.manual_seed`` reproduces a run and two calls never share a stream."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)))
        return self.model.sample(pos, node_mask, seed=seed, graph_id_base=graph_id_base)
