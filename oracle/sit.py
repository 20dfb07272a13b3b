"""Oracle: functional fp32 restatement of the node-coordinate DDPM of the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Works on a plain ``state_dict`` with the reference's parameter names.

Follows /root/reference/src/bootstrap/node_coordinate_diffusion/sit_model.py
  TimestepEmbedder        :41-77     SiTBlock (adaLN-zero)   :112-134    FinalLayer :137-154    SiT.forward :228-247
  AttentionWithPads.forward :85-103 (the reference's own code; its constructor and ``Mlp`` come from ``timm``, pinned nowhere
  in the reference's requirements and absent from this image: ``timm.models.vision_transformer.Attention`` = qkv Linear with
  bias, ``num_heads`` heads of dim / num_heads, scale head_dim ** -0.5, output Linear ``proj``; ``Mlp`` = fc1, activation, fc2)
and cow_diff_nodes_with_3d_checks.py :22-81 (linear beta schedule 1e-4 .. 0.02 over 1000 steps, ancestral sampler ``sample``).

Pinning: ``tests/golden/h1_sit_*`` hold outputs of the reference's ``SiT`` module run with a stand-in for the two timm
constructors (oracle/gen_golden.py:gen_sit says exactly what the stand-in is); the sampler loop cannot be imported at all
(open3d, matplotlib, timm.optim, torch_geometric at module level) and is restated from its source: PARITY UNPINNED for the
loop itself, pinned for the network up to the timm stand-in.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embed(sd, t):
    return _lin(sd, "t_embedder.mlp.2", F.silu(_lin(sd, "t_embedder.mlp.0", timestep_embedding(t))))


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def attention_with_pads(sd, p, x, mask, n_heads):
    B, N, C = x.shape
    hd = C // n_heads
    qkv = _lin(sd, p + "qkv", x).reshape(B, N, 3, n_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5
    pair = (mask[:, None, :] & mask[:, :, None])[:, None]
    att = att.masked_fill(~pair, -1e9).softmax(dim=-1)
    return _lin(sd, p + "proj", (att @ v).transpose(1, 2).reshape(B, N, C))


def sit_forward(sd, x, t, mask, n_blocks=6, n_heads=6):
    """SiT.forward (:228-247): x [B,N,3], t [B], mask [B,N] bool -> predicted noise [B,N,3]."""
    h = _lin(sd, "pos_embedder", x)
    c = t_embed(sd, t)
    for i in range(n_blocks):
        p = f"blocks.{i}."
        sh1, sc1, g1, sh2, sc2, g2 = _lin(sd, p + "adaLN_modulation.1", F.silu(c)).chunk(6, dim=1)
        u = _ln(h) * (1 + sc1[:, None]) + sh1[:, None]
        h = h + g1[:, None] * attention_with_pads(sd, p + "attn.", u, mask, n_heads)
        u = _ln(h) * (1 + sc2[:, None]) + sh2[:, None]
        h = h + g2[:, None] * _lin(sd, p + "mlp.fc2", F.gelu(_lin(sd, p + "mlp.fc1", u), approximate="tanh"))
    sh, sc = _lin(sd, "final_layer.adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)
    return _lin(sd, "final_layer.linear", _ln(h) * (1 + sc[:, None]) + sh[:, None])


def schedule(noise_steps=1000, beta_start=1e-4, beta_end=0.02):
    beta = torch.linspace(beta_start, beta_end, noise_steps)
    alpha = 1.0 - beta
    return beta, alpha, torch.cumprod(alpha, dim=0)


def sample(sd, x_T, mask, noises, steps=None, noise_steps=1000):
    """NodeCoordsDiffusionModel.sample (cow_diff_nodes_with_3d_checks.py:63-81) with the normal draws given:
    ``noises[k]`` is the draw of the k-th loop iteration (i = noise_steps-1, ...; unused at i = 1).  ``steps`` limits the
    number of iterations (tests)."""
    beta, alpha, alpha_hat = schedule(noise_steps)
    x = x_T.clone()
    its = list(reversed(range(1, noise_steps)))
    if steps is not None:
        its = its[:steps]
    for k, i in enumerate(its):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps = sit_forward(sd, x, t, mask)
        noise = noises[k] if i > 1 else torch.zeros_like(x)
        x = 1 / torch.sqrt(alpha[i]) * (x - ((1 - alpha[i]) / torch.sqrt(1 - alpha_hat[i])) * eps) + torch.sqrt(beta[i]) * noise
    return x
