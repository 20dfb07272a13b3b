"""The bf16x3 ("x3") kernels against the plain fp32-MFMA kernels they replaced (SCGG_EDGE_MODE / SCGG_NODE_MODE = f32,
read once per process, hence the subprocesses): same network, same inputs, the two results must agree to fp32
round-off, which is what "fp32 accuracy on the bf16 matrix pipe"  promises.  Both also stay
within the oracle tolerance through the ordinary parity tests; this one pins the two HIP paths to each other."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
torch.manual_seed(5)
dims = fx.CONSTRUCT_DIMS
ind, od, B, n = (10, 25, 13), (1, 14, 0), 6, 32
model = GraphTransformer(3, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval().cuda()
X = torch.randn(B, n, ind[0]); E = torch.randn(B, n, n, ind[1]); E = 0.5 * (E + E.transpose(1, 2)); y = torch.randn(B, ind[2])
mask = fx.ragged_mask(torch.tensor([32, 12, 27, 16, 5, 21]), n)
out = model(model_input(X.cuda(), E.cuda(), y.cuda(), mask.cuda()))
torch.cuda.synchronize()
np.savez(sys.argv[1], X=out.X.cpu().numpy(), E=out.E.cpu().numpy())
"""


def _run(tmp_path, tag, env_extra):
    out = tmp_path / f"{tag}.npz"
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT), str(out)], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.gpu
def test_x3_kernels_agree_with_fp32_mfma_kernels(tmp_path):
    a = _run(tmp_path, "x3", {})
    b = _run(tmp_path, "f32", {"SCGG_EDGE_MODE": "f32", "SCGG_NODE_MODE": "f32"})
    for k in ("X", "E"):
        scale = float(np.abs(b[k]).max())
        err = float(np.abs(a[k] - b[k]).max()) / scale
        assert err < 2e-5, f"{k}: x3 vs fp32 MFMA differ by {err:.2e} of the output scale"   # fp32 round-off over 3 layers


@pytest.mark.gpu
def test_staggered_x3s_kernel_agrees_with_the_default(tmp_path):
    """SCGG_EDGE_MODE=x3s: the staggered variant of the bf16x3 edge kernel (three LDS weight buffers, wave groups one
    sub-step apart, LDS-DMA issued from inline asm with hand-counted waits).  Same arithmetic per unit, so
    the two agree to fp32 round-off (the accumulation order of the FFN halves differs); a missed DMA wait or a
    mis-ordered buffer would show here as a gross error."""
    a = _run(tmp_path, "x3", {})
    b = _run(tmp_path, "x3s", {"SCGG_EDGE_MODE": "x3s"})
    for k in ("X", "E"):
        scale = float(np.abs(a[k]).max())
        err = float(np.abs(a[k] - b[k]).max()) / scale
        assert err < 2e-5, f"{k}: x3s vs x3 differ by {err:.2e} of the output scale"


SCRIPT_3D = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
torch.manual_seed(6)
dims = fx.VESSEL_DIMS
B, n = 6, 32
model = Graph3DTransformer(3, fx.dims_ns(1, 14, 1), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, 14, 0)).eval()
model.load_state_dict(fx.make_state_dict({{k: tuple(v.shape) for k, v in model.state_dict().items()}}, seed=62))
model = model.cuda()
mask = fx.ragged_mask(torch.tensor([32, 12, 27, 16, 5, 21]), n)
E = fx.one_hot_edges(fx.random_labels(mask, 14, 0.15, 3), mask, 14)
inp = PlaceHolder(X=mask[..., None].float().cuda(), E=E.cuda(), y=torch.rand(B, 1).cuda(), pos=torch.randn(B, n, 3).cuda(),
                  node_mask=mask.cuda(), radius=torch.zeros(B, n, n, 1).cuda())
out = model(inp)
torch.cuda.synchronize()
np.savez(sys.argv[1], X=out.radius.cpu().numpy(), E=out.E.cpu().numpy())
"""


@pytest.mark.gpu
def test_3d_edge_layer_x3_pair_agrees_with_the_one_kernel_fp32_form(tmp_path):
    """SCGG_B3D_MODE=f32: the position-aware layer as ONE fp32-MFMA kernel (all weights in LDS as fp32) against the default
    pair of bf16x3 kernels (block part, then LN-FFN-LN in place): DESIGN.md 4.9.  Logits and radius agree to fp32 round-off."""
    def run(tag, env_extra):
        out = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", SCRIPT_3D.format(root=ROOT), str(out)], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.load(out)
    a, b = run("x3", {}), run("f32", {"SCGG_B3D_MODE": "f32"})
    for k in ("X", "E"):
        scale = float(np.abs(b[k]).max())
        err = float(np.abs(a[k] - b[k]).max()) / scale
        assert err < 2e-5, f"{k}: x3 pair vs fp32 kernel differ by {err:.2e} of the output scale"
    c = run("ypool0", {"SCGG_Y_POOL": "0"})      # graph merge as its own kernel instead of inside the y round: same arithmetic
    for k in ("X", "E"):
        assert np.array_equal(a[k], c[k]), f"SCGG_Y_POOL=0: {k} differs by {float(np.abs(a[k] - c[k]).max()):.2e}"


@pytest.mark.gpu
def test_one_stream_schedule_switches_agree_bit_for_bit(tmp_path):
    """SCGG_Y_POOL=0 (the pooled statistics as separate kernels in front of the y round instead of inside it), SCGG_LATE_JOIN=0
    (side stream joined at the start of the forward) and SCGG_TWO_STREAMS=y (y round on the second stream): the same
    kernels' arithmetic in another launch arrangement (pool_stats.h is shared), so the results are identical."""
    a = _run(tmp_path, "default", {})
    for tag, env in (("ypool0", {"SCGG_Y_POOL": "0"}), ("join0", {"SCGG_LATE_JOIN": "0"}), ("ybeside", {"SCGG_TWO_STREAMS": "y"}),
                     ("xcd0", {"SCGG_EDGE_XCD": "0"})):      # plain instead of XCD-major unit order in the edge kernel
        b = _run(tmp_path, tag, env)
        for k in ("X", "E"):
            assert np.array_equal(a[k], b[k]), f"{tag}: {k} differs by {float(np.abs(a[k] - b[k]).max()):.2e}"


SCRIPT_SIT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.node_coordinate_diffusion import SiT_models
model = SiT_models["SiT-T"](in_channels=3)
model.load_state_dict(fx.make_state_dict({{k: tuple(v.shape) for k, v in model.state_dict().items()}}, seed=611))
model = model.cuda()
m = fx.ragged_mask([29, 7, 48, 1, 16])
B, n = m.shape
g = torch.Generator().manual_seed(4)
x = torch.randn(B, n, 3, generator=g)
t = torch.tensor([999, 3, 500, 1, 77])
eps = model(x.cuda(), t.cuda(), m.cuda())
noise = torch.randn(6, B, n, 3, generator=g)
xs = model.sample(x.cuda(), m.cuda(), steps=6, noise=noise.cuda())
torch.cuda.synchronize()
np.savez(sys.argv[1], X=eps.cpu().numpy(), E=xs.cpu().numpy())
"""


@pytest.mark.gpu
def test_sit_fused_block_and_mfma_attention_agree_with_the_separate_kernels(tmp_path):
    """SCGG_SIT_FUSED=0 (row-list Linears with the LayerNorm-modulate prologue and GELU / accumulate epilogues instead of
    sit_block_x3_kernel) and SCGG_SIT_ATTN=valu (LDS / vector-unit attention instead of the matrix-pipe one): the same network
    through other kernels, equal to fp32 round-off over 6 blocks (forward) and 6 ancestral steps (sampler)."""
    def run(tag, env_extra):
        out = tmp_path / f"{tag}.npz"
        r = subprocess.run([sys.executable, "-c", SCRIPT_SIT.format(root=ROOT), str(out)], env=dict(os.environ, **env_extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.load(out)
    a = run("default", {})
    for tag, env in (("unfused", {"SCGG_SIT_FUSED": "0"}), ("valu", {"SCGG_SIT_ATTN": "valu"}),
                     ("gemm", {"SCGG_SIT_FUSED": "0", "SCGG_NODE_MODE": "f32"})):
        b = run(tag, env)
        for k in ("X", "E"):
            scale = float(np.abs(a[k]).max())
            err = float(np.abs(a[k] - b[k]).max()) / scale
            assert err < 3e-5, f"{tag}: {k} differs by {err:.2e} of the output scale"
