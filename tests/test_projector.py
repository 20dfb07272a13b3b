"""Host-side validity projectors against trajectories recorded from the REAL reference
(oracle/gen_golden.py gen_projector -> tests/golden/f12_projector_*.npz): same proposals, same candidates, same
np.random seed -> the same projected labels at every step, the same blocked-edge table and correction counter."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd import projector as PJ
from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("cow", PJ.CoWProjector), ("cow_vanilla", PJ.CoWVanillaProjector), ("atm", PJ.ATMTreeProjector),
         ("atm_vanilla", PJ.ATMVanillaTreeProjector), ("tree", PJ.TreeProjector)]


def _holder(lab, cd, mask, C):
    B, n = mask.shape
    return PlaceHolder(X=mask[..., None].float(), E=fx.one_hot_edges(lab, mask, C), y=torch.zeros(B, 0), node_mask=mask,
                       pos=torch.zeros(B, n, 3), extra_info=cd)


def _blocked(proj, shape):
    out = np.zeros(shape, dtype=bool)
    for b, d in proj.blocked_edges.items():
        for (i, j), v in d.items():
            out[b, i, j] = v
    return out


@pytest.mark.parametrize("engine", ["native", "python"])
@pytest.mark.parametrize("name,cls", CASES)
def test_projector_matches_reference_trajectory(name, cls, engine):
    g = np.load(os.path.join(GOLD, f"f12_projector_{name}.npz"))
    mask, C = torch.from_numpy(g["node_mask"]), int(g["C"])
    props, cands, want = g["props"], g["cands"], g["out_labels"]
    B, n = mask.shape
    np.random.seed(int(g["seed"]))
    z = torch.zeros(B, n, n, dtype=torch.long)
    proj = cls(_holder(z, None, mask, C), torch.device("cpu"), engine=engine)
    for s in range(props.shape[0]):
        raw = torch.where(z == 0, torch.from_numpy(props[s]).long(), z)
        z_s = _holder(raw, torch.from_numpy(cands[s]).long(), mask, C)
        proj.project(z_s)
        z = torch.argmax(z_s.E, dim=3) * (z_s.E.sum(-1) > 0)
        assert np.array_equal(z.numpy(), want[s]), f"{name}: step {s} differs"
        # the reference's own consistency checks (:361-372): stored graph == 0/1 adjacency of z_s
        for b, gr in enumerate(proj.nx_graphs_list):
            k = gr.number_of_nodes()
            assert np.array_equal((z[b, :k, :k] != 0).numpy(), gr.to_numpy_array() != 0)
    assert np.array_equal(_blocked(proj, g["blocked"].shape), g["blocked"])
    fixer = getattr(proj, "top_cow_validity_check_and_fix_obj", None)
    if fixer is not None:
        assert fixer.counter == int(g["counter"])


@pytest.mark.parametrize("engine", ["native", "python"])
@pytest.mark.parametrize("name,cls", CASES)
def test_state_route_equals_placeholder_route(name, cls, engine):
    """project_labels (int8 state, -1 on padding and diagonal) == project (one-hot) on the same trajectory (CPU tensors)."""
    g = np.load(os.path.join(GOLD, f"f12_projector_{name}.npz"))
    mask = torch.from_numpy(g["node_mask"])
    props, cands, want = g["props"], g["cands"], g["out_labels"]
    B, n = mask.shape
    pair = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    np.random.seed(int(g["seed"]))
    lab = torch.where(pair, 0, -1).to(torch.int8)
    z_T = PlaceHolder(X=mask[..., None].float(), E=None, y=torch.zeros(B, 0), node_mask=mask, pos=torch.zeros(B, n, 3))
    z_T._labels = lab
    proj = cls(z_T, torch.device("cpu"), engine=engine)
    for s in range(props.shape[0]):
        lab = torch.where(lab == 0, torch.from_numpy(props[s]), lab)
        proj.project_labels(lab, torch.from_numpy(cands[s]))
        assert np.array_equal(lab.clamp(min=0).numpy(), want[s]), f"{name}: step {s} differs"
        assert bool((lab[~pair] == -1).all())


def test_line_graph_rule_and_forest_rule():
    allowed = PJ._allowed_masks(PJ.COW_ADJACENCY)
    g = PJ.LabelGraph(5)
    g.add_edge(0, 1, 1)
    g.add_edge(1, 2, 2)           # 1-2 adjacent in the table
    assert PJ.line_graph_valid(g, allowed)
    g.add_edge(1, 3, 1)           # same class twice at a node: the diagonal is cleared (:457)
    assert PJ.line_graph_valid(g, allowed)
    g.add_edge(2, 4, 13)          # 2-13 not in the table
    assert not PJ.line_graph_valid(g, allowed)
    g.add_edge(2, 4, 8)           # relabel in place: 2-8 allowed
    assert PJ.line_graph_valid(g, allowed) and g.number_of_edges() == 4
    assert PJ.has_no_cycles(g)
    g.add_edge(0, 2, 3)
    assert not PJ.has_no_cycles(g)           # triangle 0-1-2
    h = g.copy()
    h.remove_edge(0, 2)
    assert PJ.has_no_cycles(h) and not PJ.has_no_cycles(g)


def test_projector_for_names():
    assert PJ.projector_for(None) is None and PJ.projector_for("") is None
    assert PJ.projector_for("cow_proj") is PJ.CoWProjector
    assert PJ.projector_for("atm_proj_vanilla") is PJ.ATMVanillaTreeProjector
    assert PJ.projector_for("planar") is PJ.BaseValidProjector


def test_native_projector_rejects_bad_input_without_side_effects():
    import ctypes as C
    from semantically_consistent_graph_generation_amd import _lib
    L = _lib.lib()
    nn_ = np.asarray([5, 4], dtype=np.int32)
    h = C.c_void_p()
    assert L.scgg_projector_create(99, 2, nn_.ctypes.data, C.byref(h)) != 0
    assert L.scgg_projector_create(0, 2, nn_.ctypes.data, C.byref(h)) == 0
    fbg, fbc, nfb = np.zeros(2, np.int32), np.zeros(2, np.int32), C.c_int32(0)

    def begin(rows, k=4):
        rec = np.asarray(rows, dtype=np.int32).reshape(-1, 4 + k)
        return L.scgg_projector_begin(h, rec.shape[0], rec.ctypes.data, k, fbg.ctypes.data, fbc.ctypes.data, C.byref(nfb))
    good = [0, 0, 1, 1, 1, 2, 3, 0]
    assert begin([good, [0, 0, 1, 2, 1, 2, 3, 0]]) != 0 and b"sorted" in L.scgg_last_error()        # repeated pair
    assert begin([[0, 1, 0, 1, 1, 2, 3, 0]]) != 0                                                    # i >= j
    assert begin([[1, 0, 4, 1, 1, 2, 3, 0]]) != 0                                                    # node outside the graph
    assert begin([[0, 0, 1, 14, 1, 2, 3, 0]]) != 0 and b"validity table" in L.scgg_last_error()     # class 14
    assert begin([[0, 0, 1, 1]], k=0) != 0                                                           # repairs need candidates
    assert begin([good, [1, 0, 1, 1, 1, 2, 3, 0], [1, 0, 2, 13, 0, 0, 0, 0]]) == 0
    assert nfb.value == 2 and fbg.tolist() == [0, 1] and fbc.tolist() == [1, 2]      # 1 edge -> one by one; 1-13 clash, no repair
    assert begin([good]) != 0 and b"not finished" in L.scgg_last_error()
    out, ne = np.zeros((3, 4), np.int32), C.c_int64(0)
    bad = np.asarray([0, 1, 1], dtype=np.int32)
    assert L.scgg_projector_finish(h, bad.ctypes.data, 3, out.ctypes.data, 3, C.byref(ne)) != 0       # not a permutation
    assert L.scgg_projector_finish(h, bad.ctypes.data, 2, out.ctypes.data, 3, C.byref(ne)) != 0       # wrong length
    perm = np.asarray([0, 0, 1], dtype=np.int32)           # identity for both graphs
    assert L.scgg_projector_finish(h, perm.ctypes.data, 3, out.ctypes.data, 3, C.byref(ne)) == 0
    got = {tuple(r) for r in out[:ne.value].tolist()}
    # graph 0 keeps (0,1)=1; graph 1: (0,1)=1 enters first, then (0,2)=13 clashes with it, no candidate -> deleted and blocked
    assert got == {(0, 0, 1, 1), (1, 0, 1, 1), (1, 0, 2, 0)}
    blk = np.zeros((4, 4), np.uint8)
    lab = np.zeros((4, 4), np.int8)
    assert L.scgg_projector_export(h, 1, lab.ctypes.data, blk.ctypes.data) == 0
    assert blk[0, 2] == 1 and blk.sum() == 1 and lab[0, 1] == 1 and lab[1, 0] == 1 and lab.astype(bool).sum() == 2
    L.scgg_projector_destroy(h)


@pytest.mark.parametrize("name,cls", CASES)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_engine_equals_python_engine_on_random_trajectories(name, cls, seed):
    """Beyond the recorded trajectories: denser proposals, more steps, other sizes; the C++ engine and the readable
    restatement must agree on every step, on the blocked tables and on the repair counter."""
    rs = np.random.RandomState(100 + seed)
    C = 5 if "atm" in name or name == "tree" else 14
    n_nodes = [int(v) for v in rs.randint(1, 12, size=7)]
    n = max(n_nodes)
    mask = torch.arange(n)[None, :] < torch.tensor(n_nodes)[:, None]
    B = len(n_nodes)
    pair = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    outs = {}
    for engine in ("native", "python"):
        r2 = np.random.RandomState(200 + seed)
        shuffle_rng = np.random.RandomState(300 + seed)
        lab = torch.where(pair, 0, -1).to(torch.int8)
        z_T = PlaceHolder(X=mask[..., None].float(), E=None, y=torch.zeros(B, 0), node_mask=mask, pos=torch.zeros(B, n, 3))
        z_T._labels = lab
        proj = cls(z_T, torch.device("cpu"), engine=engine, rng=shuffle_rng)
        steps = []
        for s in range(16):
            prop = np.triu(r2.randint(1, C, (B, n, n)) * (r2.rand(B, n, n) < 0.15), 1)
            prop = torch.from_numpy(prop + prop.transpose(0, 2, 1)).to(torch.int8)
            cd = torch.from_numpy(r2.randint(0, C, (B, n, n, 3))).to(torch.int8)
            cd = torch.where(torch.triu(torch.ones(n, n, dtype=torch.bool), 1)[None, :, :, None], cd, cd.transpose(1, 2))
            lab = torch.where((lab == 0) & pair, prop, lab).contiguous()
            proj.project_labels(lab, cd.contiguous())
            steps.append(lab.clone())
        fixer = getattr(proj, "top_cow_validity_check_and_fix_obj", None)
        outs[engine] = (steps, proj.blocked_edges, fixer.counter if fixer is not None else 0)
    for a, b in zip(outs["native"][0], outs["python"][0]):
        assert torch.equal(a, b)
    assert outs["native"][1] == outs["python"][1] and outs["native"][2] == outs["python"][2]
    assert int((outs["native"][0][-1] > 0).sum()) > 0
