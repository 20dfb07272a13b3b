"""GPU parity of the position-aware denoiser (Graph3DTransformer / EquivariantGraph3DTransformer through
scgg_denoiser3d_forward) against (a) the golden outputs of the real reference and (b) the CPU oracle layer by layer.

Tolerance: 1e-4 relative to the tensor's scale.  Distances: the reference calls torch.cdist, which above 25 points
evaluates |a|^2 + |b|^2 - 2 a.b by a matrix product; there the distance of a node to itself is the square root of
rounding noise (up to ~2e-3) and depends on the BLAS.  The HIP kernel takes coordinate differences at every size, so
* for graphs of at most 25 nodes the reference's CPU goldens are met at 1e-4;
* for larger graphs the comparison at 1e-4 is against the oracle in its "direct" mode (the oracle's "cdist" mode is what
  tests/test_oracle_golden.py pins against the same goldens), and the goldens themselves are met at the looser bound
  GOLDEN_LARGE_REL, which is the reference's own noise, not ours."""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_sd, golden_shapes
from oracle import denoiser as D
from oracle import denoiser3d as D3
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu

REL = 1e-4
GOLDEN_LARGE_REL = 2e-2
CASES = {   # name: (dims, equivariant, every graph <= 25 nodes)
    "g3_g3d_tiny": (fx.TINY_DIMS, False, True), "g3_g3d_tiny_equivariant": (fx.TINY_DIMS, True, True),
    "g3_g3d_vessel_cow": (fx.VESSEL_DIMS, False, True), "g3_g3d_vessel_n40": (fx.VESSEL_DIMS, False, False),
    "g3_g3d_vessel_equivariant": (fx.VESSEL_DIMS, True, False),
}


def rel_err(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def build(name):
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer, EquivariantGraph3DTransformer
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    g = load_golden(name)
    dims, eq, small = CASES[name]
    ind, od = [int(v) for v in g["in_dims"]], [int(v) for v in g["out_dims"]]
    sd = golden_sd(g) or fx.make_state_dict(golden_shapes(g), int(g["seed"]))
    cls = EquivariantGraph3DTransformer if eq else Graph3DTransformer
    model = cls(int(g["n_layers"]), fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    dev = torch.device("cuda")
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    radius = g["out_radius"].size > 0
    B, n = g["node_mask"].shape
    inp = PlaceHolder(X=T("X"), E=T("E"), y=T("y"), pos=T("pos"), node_mask=T("node_mask"),
                      radius=torch.zeros(B, n, n, 1, device=dev) if radius else None)
    return g, sd, dims, eq, small, od, model, inp


def oracle_forward(g, sd, dims, eq, od, mode, hidden=False):
    T = torch.from_numpy
    dd = D.DenoiserDims(int(g["n_layers"]), dims["hidden_dims"]["n_head"], od[0], od[1], od[2], od[3])
    old = D3.DIST_MODE
    D3.DIST_MODE = mode
    try:
        return D3.graph3d_transformer_forward(sd, dd, T(g["X"]), T(g["E"]), T(g["y"]), T(g["pos"]), T(g["node_mask"]),
                                              with_radius=g["out_radius"].size > 0, equivariant=eq, return_hidden=hidden)
    finally:
        D3.DIST_MODE = old


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_golden(name):
    g, sd, dims, eq, small, od, model, inp = build(name)
    out = model(inp)
    torch.cuda.synchronize()
    bound = REL if small else GOLDEN_LARGE_REL
    assert rel_err(out.E, g["outE"]) < bound
    assert rel_err(out.X, g["outX"]) < bound
    assert rel_err(out.charges, g["out_charges"]) < bound
    assert rel_err(out.pos, g["out_pos"]) < 1e-6
    if inp.radius is not None:
        assert rel_err(out.radius, g["out_radius"]) < bound
    else:
        assert out.radius is None
    assert out.y is None
    # against the oracle on differences at every size: the 1e-4 bar
    oX, oc, oE, orad = oracle_forward(g, sd, dims, eq, od, "direct")
    assert rel_err(out.E, oE) < REL
    assert rel_err(out.X, oX) < REL
    if orad is not None:
        assert rel_err(out.radius, orad) < REL
    # structural invariants the reference asserts (utils.py:155-159)
    E, m = out.E.cpu(), torch.from_numpy(g["node_mask"])
    assert torch.equal(E, E.transpose(1, 2))
    pm = (m[:, :, None] & m[:, None, :]) & ~torch.eye(m.shape[1], dtype=torch.bool)[None]
    assert (E[~pm] == 0).all() and (out.X.cpu()[~m] == 0).all()
    if out.radius is not None:
        R = out.radius.cpu()
        assert torch.equal(R, R.transpose(1, 2)) and (R[~pm] == 0).all()


@pytest.mark.parametrize("name", ["g3_g3d_tiny", "g3_g3d_tiny_equivariant", "g3_g3d_vessel_n40"])
def test_hidden_states_match_oracle_layer_by_layer(name):
    g, sd, dims, eq, small, od, model, inp = build(name)
    _, hidden = oracle_forward(g, sd, dims, eq, od, "direct", hidden=True)
    for l, (hX, hE, hy, _) in enumerate(hidden):
        X, E, y = model.forward_hidden(inp, l)
        torch.cuda.synchronize()
        assert rel_err(E, hE) < REL, f"E after {l} layers"
        assert rel_err(X, hX) < REL, f"X after {l} layers"
        if hy is not None:
            assert rel_err(y, hy) < REL, f"y after {l} layers"


def test_scattered_mask_and_repeatability():
    """Valid nodes anywhere in the row (not a prefix), a node at the origin, two coincident nodes; the same forward twice
    is bit-identical (one stream, no atomics)."""
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    torch.manual_seed(5)
    dims, ind, od, L = fx.VESSEL_DIMS, (1, 5, 1, 0), (1, 5, 0, 0), 3
    model = Graph3DTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=51)
    model.load_state_dict(sd)
    model = model.cuda()
    B, n = 5, 23
    mask = torch.rand(B, n) < 0.6
    mask[0] = True
    mask[1] = False
    mask[1, 7] = True                       # a single node
    mask[2, :16] = False                    # an empty first chunk
    mask[2, 16:] = True
    lab = fx.random_labels(mask, 5, 0.2, 9)
    E = fx.one_hot_edges(lab, mask, 5)
    X = mask[..., None].float()
    y = torch.rand(B, 1)
    pos = torch.randn(B, n, 3)
    pos[0, 3] = 0.0                         # |p| = 0: cosine through the 1e-7 guard
    pos[0, 5] = pos[0, 4]                   # coincident nodes: distance exactly 0 off the diagonal
    dev = torch.device("cuda")
    inp = PlaceHolder(X=X.to(dev), E=E.to(dev), y=y.to(dev), pos=pos.to(dev), node_mask=mask.to(dev),
                      radius=torch.zeros(B, n, n, 1, device=dev))
    out = model(inp)
    E1, R1 = out.E.clone(), out.radius.clone()
    out2 = model(inp)
    torch.cuda.synchronize()
    assert torch.equal(E1, out2.E) and torch.equal(R1, out2.radius)
    dd = D.DenoiserDims(L, dims["hidden_dims"]["n_head"], 1, 5, 0, 0)
    old = D3.DIST_MODE
    D3.DIST_MODE = "direct"
    try:
        oX, oc, oE, orad = D3.graph3d_transformer_forward(sd, dd, X, E, y, pos, mask)
    finally:
        D3.DIST_MODE = old
    assert rel_err(E1, oE) < REL
    assert rel_err(R1, orad) < REL
    assert rel_err(out.X, oX) < REL


def test_vessel_batch_512_slice_and_invariants():
    """The shipped vessel configuration at sampling size (B = 512, n <= 32, CoW classes): a slice of graphs against the
    oracle, structural invariants on the whole batch."""
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    dims, ind, od, L = fx.VESSEL_DIMS, (1, 14, 1, 0), (1, 14, 0, 0), 5
    model = Graph3DTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=52)
    model.load_state_dict(sd)
    model = model.cuda()
    g = torch.Generator().manual_seed(3)
    B, n = 512, 32
    nn_ = torch.randint(12, 33, (B,), generator=g)
    mask = fx.ragged_mask(nn_, n)
    E = fx.one_hot_edges(fx.random_labels(mask, 14, 0.1, 4), mask, 14)
    X = mask[..., None].float()
    y = torch.rand(B, 1, generator=g)
    pos = torch.randn(B, n, 3, generator=g)
    dev = torch.device("cuda")
    inp = PlaceHolder(X=X.to(dev), E=E.to(dev), y=y.to(dev), pos=pos.to(dev), node_mask=mask.to(dev), radius=None)
    out = model(inp)
    torch.cuda.synchronize()
    Eo = out.E.cpu()
    assert torch.isfinite(Eo).all() and torch.equal(Eo, Eo.transpose(1, 2))
    pm = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    assert (Eo[~pm] == 0).all()
    sl = torch.tensor([0, 1, 137, 300, 511])
    dd = D.DenoiserDims(L, dims["hidden_dims"]["n_head"], 1, 14, 0, 0)
    old = D3.DIST_MODE
    D3.DIST_MODE = "direct"
    try:
        oX, oc, oE, _ = D3.graph3d_transformer_forward(sd, dd, X[sl], E[sl], y[sl], pos[sl], mask[sl], with_radius=False)
    finally:
        D3.DIST_MODE = old
    assert rel_err(Eo[sl], oE) < REL
    assert rel_err(out.X.cpu()[sl], oX) < REL


def test_chain_with_3d_denoiser_matches_cpu_oracle_bit_exactly():
    """T = 12 absorbing-edge chain driven by Graph3DTransformer (tiny dims): labels and candidates of the device chain equal
    a CPU chain built from the oracle forward (differences mode), the oracle reverse step and the same Philox stream; the
    positions every step sees are the reference's drifting frame (noise_model.py:446 -> utils.py:160-164), and the radius
    handed back is the last step's prediction on the sampled edges (noise_model.py:415-420)."""
    from oracle import noise as N
    from test_host_logic import make_noise_model
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)
    C, T, top_k, seed, base = 6, 12, 4, 77, 16
    dims = fx.TINY_DIMS
    model = Graph3DTransformer(2, fx.dims_ns(1, C, 1), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0)).eval()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=61)
    model.load_state_dict(sd)
    model = model.cuda()
    nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
    sampler = DiscreteSampler(model, nm)
    n_nodes = [9, 4, 7, 1]
    labels, cands, mask, pos_end = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, return_state=True)
    radius = sampler.last_radius
    torch.cuda.synchronize()
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    raw = nm.sample_limit_dist(mask, with_onehot=False, rng=Rng(seed=seed, graph_id_base=base))._pos_raw.cpu()
    m = fx.ragged_mask(n_nodes)
    B, n = m.shape
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tr = N.Transition(N.Schedule(T, "absorbing_edges", NU), [1.0], np.ones(C, np.float32) / C)
    dd = D.DenoiserDims(2, dims["hidden_dims"]["n_head"], 1, C, 0)
    lab = torch.where(live, 0, -1)
    X = m[..., None].float()
    pos = D3.recentre(raw, m)
    old = D3.DIST_MODE
    D3.DIST_MODE = "direct"
    try:
        for s in reversed(range(T)):
            t = s + 1
            E = fx.one_hot_edges(lab.clamp(min=0), m, C)
            y = torch.full((B, 1), t / T)
            _, _, logits, rad = D3.graph3d_transformer_forward(sd, dd, X, E, y, pos, m)
            u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
            r = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), t), torch.full((B, 1), s),
                               np.zeros((B, n), np.float32), u0, uk)
            lab = r["labels_E"]
            pos = D3.recentre(pos, m)
    finally:
        D3.DIST_MODE = old
    assert int((labels.cpu().long() != lab).sum()) == 0
    assert int(((cands.cpu().long() * live[..., None]) != r["extra_info"]).sum()) == 0
    assert int((lab > 0).sum()) > 0
    assert rel_err(pos_end, pos) < 1e-5
    assert rel_err(radius, (lab > 0) * rad.squeeze(-1)) < REL


def test_vessel_graphs_of_256_nodes_match_oracle():
    """ATM-size graphs (n_max = 256: 16 key chunks per row, 16 units merged per row, one full and one ragged graph), vessel
    dims, 2 layers, both heads: against the oracle (differences mode) at 1e-4, plus the structural invariants."""
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    dims, ind, od, L = fx.VESSEL_DIMS, (1, 5, 1, 0), (1, 5, 0, 0), 2
    model = Graph3DTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=53)
    model.load_state_dict(sd)
    model = model.cuda()
    g = torch.Generator().manual_seed(8)
    n_nodes = [256, 203]
    mask = fx.ragged_mask(n_nodes)
    B, n = mask.shape
    E = fx.one_hot_edges(fx.random_labels(mask, 5, 0.01, 6), mask, 5)
    X = mask[..., None].float()
    y = torch.tensor([[0.37], [0.81]])
    pos = torch.randn(B, n, 3, generator=g) * 2.0
    dev = torch.device("cuda")
    inp = PlaceHolder(X=X.to(dev), E=E.to(dev), y=y.to(dev), pos=pos.to(dev), node_mask=mask.to(dev),
                      radius=torch.zeros(B, n, n, 1, device=dev))
    out = model(inp)
    torch.cuda.synchronize()
    dd = D.DenoiserDims(L, dims["hidden_dims"]["n_head"], 1, 5, 0, 0)
    old = D3.DIST_MODE
    D3.DIST_MODE = "direct"
    try:
        oX, oc, oE, orad = D3.graph3d_transformer_forward(sd, dd, X, E, y, pos, mask)
    finally:
        D3.DIST_MODE = old
    assert rel_err(out.E, oE) < REL
    assert rel_err(out.radius, orad) < REL
    assert rel_err(out.X, oX) < REL
    Eo = out.E.cpu()
    pm = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    assert torch.equal(Eo, Eo.transpose(1, 2)) and (Eo[~pm] == 0).all()
