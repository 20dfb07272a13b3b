"""CPU-side checks of the product's host logic (no GPU, no compute calls): schedule tables and transition matrices
against the reference's golden vectors, PlaceHolder semantics, the drop-in state_dict layout, input-dim bookkeeping,
and that libscgg.so loads and exports every symbol include/scgg.h declares."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_shapes
from oracle import fixtures as fx

PKG = "semantically_consistent_graph_generation_amd"


def cfg(transition, T=500, top_k=4):
    return SimpleNamespace(model=SimpleNamespace(nu=dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1),
                                                 transition=transition, diffusion_steps=T, top_k=top_k),
                           dataset=SimpleNamespace(name="cow"))


def make_noise_model(tr, C=5, e_marg=None, T=500, top_k=4):
    from semantically_consistent_graph_generation_amd import noise_model as NM
    xm, cm = torch.ones(1), torch.zeros(0)
    em = torch.ones(C) / C if e_marg is None else torch.as_tensor(e_marg)
    if tr == "absorbing_edges":
        return NM.AbsorbingEdgesTransition(cfg(tr, T, top_k), xm, em, cm, 0)
    if tr == "marginal":
        return NM.MarginalTransition(cfg(tr, T, top_k), xm, em, cm, 0)
    if tr == "uniform":
        return NM.DiscreteUniformTransition(cfg(tr, T, top_k), fx.dims_ns(1, C, 0, 0))
    return NM.AbsorbingTransition(cfg(tr, T, top_k), fx.dims_ns(1, C, 0, 0))


@pytest.mark.parametrize("tr", ["absorbing_edges", "marginal", "uniform", "absorbing"])
def test_noise_model_tables_and_Q(tr):
    g = load_golden("f7_schedules")
    em = g["e_marginals"] if tr in ("marginal", "absorbing_edges") else None
    nm = make_noise_model(tr, 5, em)
    np.testing.assert_allclose(nm._betas.numpy(), g[f"{tr}/betas"], rtol=1e-12)
    np.testing.assert_allclose(nm._alphas_bar.numpy(), g[f"{tr}/alphas_bar"], rtol=1e-12)
    if tr == "absorbing_edges":
        np.testing.assert_allclose(nm._betas_abs.numpy(), g[f"{tr}/betas_abs"], rtol=1e-12)
        np.testing.assert_allclose(nm._alphas_abs_bar.numpy(), g[f"{tr}/alphas_abs_bar"], rtol=1e-12)
    t = torch.from_numpy(g["t"])
    qt, qtb = nm.get_Qt(t), nm.get_Qt_bar(t)
    np.testing.assert_allclose(qt.E.numpy(), g[f"{tr}/Qt_E"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(qtb.E.numpy(), g[f"{tr}/Qtb_E"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(qt.X.numpy(), g[f"{tr}/Qt_X"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(qtb.X.numpy(), g[f"{tr}/Qtb_X"], rtol=1e-6, atol=1e-7)
    # the fp32 tables handed to the kernel are exactly what get_Qt / get_Qt_bar index
    beta, abar = nm._edge_tables()
    k = 1 if tr != "absorbing" else 1
    np.testing.assert_allclose(qt.E[:, k, 0].numpy() / max(float(nm.E_marginals[0]), 1e-30) if nm.E_marginals[0] > 0 else beta[t[:, 0]].numpy(),
                               beta[t[:, 0]].numpy(), rtol=1e-6)
    assert nm.T == 500 and nm.E_classes == 5 and nm.X_classes == 1


def test_placeholder_mask_and_collapse():
    from semantically_consistent_graph_generation_amd import PlaceHolder
    g = load_golden("f11_placeholder")
    m = torch.from_numpy(g["node_mask"])
    ph = PlaceHolder(X=torch.from_numpy(g["X"]), E=torch.from_numpy(g["E"]), y=torch.zeros(2, 0),
                     charges=torch.zeros(2, 5, 0), node_mask=m, pos=torch.from_numpy(g["pos"])).mask(check_symmetry=True)
    np.testing.assert_array_equal(ph.X.numpy(), g["mX"])
    np.testing.assert_array_equal(ph.E.numpy(), g["mE"])
    np.testing.assert_allclose(ph.pos.numpy(), g["mpos"], atol=1e-7)
    col = ph.collapse(None)
    np.testing.assert_array_equal(col.X.numpy(), g["cX"])
    np.testing.assert_array_equal(col.E.numpy(), g["cE"])
    parts = ph.split()
    assert [p.X.shape[0] for p in parts] == [5, 3] and parts[1].E.shape == (3, 3, 4)


@pytest.mark.parametrize("name,dims", [("f3_gt_construct_cow", fx.CONSTRUCT_DIMS), ("f3_gt_vessel_atm", fx.VESSEL_DIMS),
                                       ("f3_gt_tiny", fx.TINY_DIMS), ("f3_gt_tiny_noy", fx.TINY_DIMS)])
def test_state_dict_layout_matches_reference(name, dims):
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    g = load_golden(name)
    model = GraphTransformer(int(g["n_layers"]), fx.dims_ns(*[int(v) for v in g["in_dims"]]), dims["hidden_mlp_dims"],
                             dims["hidden_dims"], fx.dims_ns(*[int(v) for v in g["out_dims"]]))
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == golden_shapes(g)
    # dim_ffy of the yaml is ignored: lin_y1 is dy -> 2048 (quirk 3)
    assert model.tf_layers[0].lin_y1.out_features == 2048


def test_forward_refuses_to_run_without_gpu_tensors():
    from semantically_consistent_graph_generation_amd import _lib
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
    model = GraphTransformer(2, fx.dims_ns(5, 9, 6), fx.TINY_DIMS["hidden_mlp_dims"], fx.TINY_DIMS["hidden_dims"],
                             fx.dims_ns(3, 4, 2)).eval()
    with pytest.raises(_lib.ScggError):
        model(model_input(torch.zeros(1, 3, 5), torch.zeros(1, 3, 3, 9), torch.zeros(1, 6), torch.ones(1, 3, dtype=torch.bool)))


def test_update_input_dims_and_dummy():
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures, DummyExtraFeatures
    c = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5,
                                              num_eigenvalues=9, eigenfeatures=True))
    ef = ExtraFeatures(c, SimpleNamespace(max_n_nodes=32))
    d = ef.update_input_dims(fx.dims_ns(1, 14, 1))
    assert (d.X, d.E, d.y) == (10, 25, 43)                       # CoW, SURVEY.md section 8
    d = ef.update_input_dims(fx.dims_ns(1, 5, 1))
    assert (d.X, d.E, d.y) == (10, 16, 34)                       # ATM
    out = DummyExtraFeatures()(SimpleNamespace(X=torch.zeros(2, 3, 1), E=torch.zeros(2, 3, 3, 5), y=torch.zeros(2, 0)))
    assert out.X.shape == (2, 3, 0) and out.E.shape == (2, 3, 3, 0) and out.y.shape == (2, 0)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "scgg.h")).read()
    declared = sorted(set(re.findall(r"\b(scgg_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    so = os.path.join(ROOT, PKG, "libscgg.so")
    assert os.path.exists(so), "build the library first (__graft_entry__.build())"
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/scgg.h but not exported"
    from semantically_consistent_graph_generation_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared
    assert _lib.lib().scgg_version() >= 1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, PKG)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_sample_plan_matches_reference_batching_rule():
    """plan_sample_batches against the batches the REAL sample_n_graphs asked for (tests/golden/f13_sample_plan.npz,
    oracle/gen_golden.py gen_sample_plan): int(1.8*bs) in the first part, the float 1.8*bs in the remainder, the
    adaptive rule with sorting."""
    from semantically_consistent_graph_generation_amd.sampling import effective_batch_size, plan_sample_batches
    g = np.load(os.path.join(ROOT, "tests", "golden", "f13_sample_plan.npz"))
    for name in g["names"]:
        n_gen, n_chain, n_save, bs, adaptive, ref_bs = g[f"{name}/args"].tolist()
        plan = plan_sample_batches(g[f"{name}/first"], g[f"{name}/rest"], batch_size=bs, adaptive_loader=bool(adaptive),
                                   reference_batch_size=ref_bs or None)
        assert [len(p) for p in plan] == g[f"{name}/batch_sizes"].tolist(), name
        assert [v for p in plan for v in p] == g[f"{name}/batches_flat"].tolist(), name
        assert sum(len(p) for p in plan) == n_gen
    assert effective_batch_size(20, 10) == 10 and effective_batch_size(40, 10, sampling=True) == 4
    assert plan_sample_batches([], []) == []


def test_load_lightning_checkpoint_key_mapping(tmp_path):
    """`model.` prefix of the reference's Lightning module (diffusion_model_discrete_3d.py:111-132), strict match, other
    keys reported; the file is read with weights_only=True."""
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, load_lightning_checkpoint
    dims = fx.TINY_DIMS
    make = lambda: GraphTransformer(2, fx.dims_ns(4, 11, 7), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, 6, 0))
    src, dst = make(), make()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in src.state_dict().items()}, seed=3)
    ckpt = {"state_dict": {**{"model." + k: v for k, v in sd.items()}, "train_loss.lambda_train": torch.ones(4),
                           "noise_model._betas": torch.zeros(3)}, "epoch": 7}
    path = tmp_path / "last.ckpt"
    torch.save(ckpt, path)
    ignored = load_lightning_checkpoint(dst, str(path))
    assert sorted(ignored) == ["noise_model._betas", "train_loss.lambda_train"]
    for k, v in dst.state_dict().items():
        assert torch.equal(v, sd[k]), k
    bad = {"state_dict": {"model." + k: v for k, v in list(sd.items())[1:]}}
    with pytest.raises(RuntimeError):
        load_lightning_checkpoint(make(), bad)
    with pytest.raises(KeyError):
        load_lightning_checkpoint(make(), {"state_dict": {"other.w": torch.ones(1)}})


def test_pos_follows_the_reference_recentring_per_mask_call():
    """utils.py:160-164 applied once in sample_limit_dist (:340) and once per reverse step (:446): the closed form used by
    the sampling loop equals t iterated PlaceHolder.mask calls (pos was centred only once)."""
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    torch.manual_seed(0)
    B, n = 5, 9
    m = fx.ragged_mask(torch.tensor([9, 4, 1, 6, 0]), n)
    x = torch.randn(B, n, 3)
    ph = PlaceHolder(X=None, E=None, y=None, pos=x.clone(), node_mask=m)
    raw = x * m[..., None]
    for t in range(1, 40):
        ph.mask()
        np.testing.assert_allclose(NoiseModel.pos_after_masks(raw, m, t).numpy(), ph.pos.numpy(), atol=2e-6)
    g = load_golden("f11_placeholder")
    gm = torch.from_numpy(g["node_mask"])
    np.testing.assert_allclose(NoiseModel.pos_after_masks(torch.from_numpy(g["pos"]) * gm[..., None], gm, 1).numpy(), g["mpos"],
                               atol=1e-6)


def test_package_creates_no_streams_of_its_own():
    """the only extra stream is the engine's (scgg_denoiser_side_stream, wrapped as an ExternalStream); a
    third stream created from Python hung the GPU box in round 1."""
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parents[1] / "semantically_consistent_graph_generation_amd"
    offenders = [str(p) for p in root.rglob("*.py") if re.search(r"cuda\.Stream\(|hipStreamCreate", p.read_text())]
    assert offenders == []
    csrc = [p for p in (root / "csrc").glob("*.hip") if "hipStreamCreate" in p.read_text()]
    assert [p.name for p in csrc] == ["denoiser.hip"]


def test_vtp_export_and_checkpoint_keys(tmp_path):
    """Row f4: save_graphs (utils.py:272-315) and the Lightning ``model.`` prefix.  The file is parsed back: points, both
    orientations of every edge in nonzero() order, edge_type per line; an edgeless graph is skipped without using an index."""
    import xml.etree.ElementTree as ET
    from semantically_consistent_graph_generation_amd.export import save_graphs
    from semantically_consistent_graph_generation_amd.transformer_model import load_lightning_checkpoint
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    from oracle import fixtures as fx
    m = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 0], [1, 1, 1, 1]], dtype=torch.bool)
    E = torch.full((3, 4, 4), -1, dtype=torch.long)
    E[0, :3, :3] = torch.tensor([[0, 2, 0], [2, 0, 5], [0, 5, 0]])
    E[1, :2, :2] = 0                                         # no edges: skipped
    E[2] = torch.tensor([[0, 0, 0, 1], [0, 0, 3, 0], [0, 3, 0, 0], [1, 0, 0, 0]])
    pos = torch.arange(36, dtype=torch.float32).reshape(3, 4, 3) / 7
    ph = PlaceHolder(X=torch.where(m, 0, -1), charges=torch.where(m, 0, 1000), E=E, y=torch.zeros(3, 0), node_mask=m, pos=pos)
    files = save_graphs([ph], str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["graph_0.vtp", "graph_1.vtp"]
    for f, b, nb in ((files[0], 0, 3), (files[1], 2, 4)):
        piece = ET.parse(f).getroot().find("PolyData/Piece")
        pts = np.array(piece.find("Points/DataArray").text.split(), dtype=np.float32).reshape(-1, 3)
        np.testing.assert_allclose(pts, pos[b, :nb].numpy(), rtol=1e-7)
        arrs = {a.get("Name"): np.array(a.text.split(), dtype=np.int64) for a in piece.find("Lines")}
        want = torch.nonzero(E[b, :nb, :nb]).numpy()
        np.testing.assert_array_equal(arrs["connectivity"].reshape(-1, 2), want)
        np.testing.assert_array_equal(arrs["offsets"], 2 * (np.arange(len(want)) + 1))
        et = np.array(piece.find("CellData/DataArray").text.split(), dtype=np.int64)
        np.testing.assert_array_equal(et, E[b, :nb, :nb][want[:, 0], want[:, 1]].numpy())
        assert int(piece.get("NumberOfLines")) == len(want)
    model = GraphTransformer(1, fx.dims_ns(1, 5, 1), fx.TINY_DIMS["hidden_mlp_dims"], fx.TINY_DIMS["hidden_dims"], fx.dims_ns(1, 5, 0))
    ckpt = {"state_dict": {**{"model." + k: v.clone() + 1 for k, v in model.state_dict().items()},
                           "train_loss.node_loss.total_ce": torch.zeros(1), "noise_model._betas": torch.zeros(3)}, "epoch": 3}
    before = {k: v.clone() for k, v in model.state_dict().items()}
    ignored = load_lightning_checkpoint(model, ckpt)
    assert sorted(ignored) == ["noise_model._betas", "train_loss.node_loss.total_ce"]
    assert all(torch.equal(v, before[k] + 1) for k, v in model.state_dict().items())
    with pytest.raises(KeyError):
        load_lightning_checkpoint(model, {"state_dict": {"other.weight": torch.zeros(1)}})


@pytest.mark.parametrize("name,dims,eq", [("g3_g3d_vessel_cow", fx.VESSEL_DIMS, False), ("g3_g3d_tiny_equivariant", fx.TINY_DIMS, True)])
def test_graph3d_state_dict_layout_matches_reference(name, dims, eq):
    """Row f1: the parameter containers of Graph3DTransformer / EquivariantGraph3DTransformer carry the reference's
    state_dict names and shapes (recorded from the reference's own modules in the golden files), and refuse to run on CPU."""
    from semantically_consistent_graph_generation_amd import pos_transformer_model as P3
    from semantically_consistent_graph_generation_amd import _lib
    g = load_golden(name)
    cls = P3.EquivariantGraph3DTransformer if eq else P3.Graph3DTransformer
    m = cls(int(g["n_layers"]), fx.dims_ns(*[int(v) for v in g["in_dims"]]), dims["hidden_mlp_dims"], dims["hidden_dims"],
            fx.dims_ns(*[int(v) for v in g["out_dims"]])).eval()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == golden_shapes(g)
    T = torch.from_numpy
    with pytest.raises(_lib.ScggError):
        m(SimpleNamespace(X=T(g["X"]), E=T(g["E"]), y=T(g["y"]), pos=T(g["pos"]), node_mask=T(g["node_mask"]), radius=None))
    with pytest.raises(_lib.ScggError):
        cls(2, fx.dims_ns(1, 5, 1), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, 5, 2))   # no mlp_out_y


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference only exists in the build container")
def test_golden_recipe_leaves_no_stand_ins_behind():
    """``python -m oracle.gen_golden`` crashed half way because gen_sample_plan left its stand-in modules in
    sys.modules.  Runs that generator in a child process (output to a scratch directory) and checks that the modules it
    shadowed are gone afterwards and that the real position-aware model still imports."""
    import subprocess, sys, tempfile
    code = ("import sys\n"
            "from oracle import gen_golden as G\n"
            "R = G.import_reference()\n"
            "G.gen_sample_plan(R)\n"
            "for k in ('pytorch_lightning', 'src.ConStruct.models.pos_transformer_model', 'src.ConStruct.metrics.train_metrics',\n"
            "          'src.ConStruct.diffusion_model_discrete_3d', 'torch_geometric.loader'):\n"
            "    assert k not in sys.modules, k\n"
            "from src.ConStruct.models import pos_transformer_model as P3\n"
            "assert hasattr(P3, 'NodeEdgeBlock') and hasattr(P3, 'Graph3DTransformer')\n")
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, SCGG_GOLDEN_OUT=d)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
