"""The denoiser forward and the whole reverse-step input assembly are pure functions of their inputs: the same call repeated
must return the same bits.  (Round 2: the two-stream schedule of the forward did not -- a few per cent of the forwards
differed by up to 1e-2 in the logits, DESIGN.md "Two-stream schedule" -- which no tolerance test against the oracle catches
reliably; this one does.)"""
from types import SimpleNamespace

import pytest
import torch

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims_name,ind,C,n_nodes,L", [
    ("construct", (10, 25, 43), 14, [32, 17, 25, 12, 30, 31, 8, 1] * 8, 5),
    ("vessel", (10, 16, 34), 5, [40, 23, 36, 40, 7, 33], 2),
])
def test_forward_is_bit_reproducible(dims_name, ind, C, n_nodes, L):
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
    dims = fx.CONSTRUCT_DIMS if dims_name == "construct" else fx.VESSEL_DIMS
    m = fx.ragged_mask(n_nodes)
    B, n = m.shape
    model = GraphTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0)).eval()
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 5))
    model = model.cuda()
    g = torch.Generator().manual_seed(1)
    X = torch.randn(B, n, ind[0], generator=g).cuda()
    E = torch.randn(B, n, n, ind[1], generator=g)
    E = (0.5 * (E + E.transpose(1, 2))).cuda()
    y = torch.randn(B, ind[2], generator=g).cuda()
    inp = model_input(X, E, y, m.cuda())
    ref = model(inp)
    refE, refX = ref.E.clone(), ref.X.clone()
    hid = [tuple(t.clone() for t in model.forward_hidden(inp, l)) for l in range(L)]
    bad = 0
    for r in range(150):
        out = model(inp)
        bad += int(not (torch.equal(out.E, refE) and torch.equal(out.X, refX)))
    for r in range(30):
        for l in range(L):
            h = model.forward_hidden(inp, l)
            bad += int(not all(torch.equal(a, b) for a, b in zip(h, hid[l])))
    assert bad == 0, f"{bad} of {150 + 30 * L} repeated forwards differ from the first"


@pytest.mark.parametrize("dims_name,ind,C,n_nodes,L", [
    ("tiny", (1, 6, 1), 6, [9, 4, 7, 1], 2),
    ("construct", (10, 25, 43), 14, [32, 17, 25, 12, 30, 31, 8, 1], 5),
    ("vessel", (10, 16, 34), 5, [40, 23, 36, 7], 2),
])
def test_outputs_do_not_depend_on_what_the_workspace_held(dims_name, ind, C, n_nodes, L):
    """The caller's workspace is uninitialised memory: whatever it holds (zeros in a fresh process; -1 labels, NaNs or old
    activations after other work on the device) must not reach the outputs.  Round 3: the fused edge kernel read the K / V
    rows of masked nodes, which the row-list Linears never write, and relied on their being finite."""
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
    dims = {"tiny": fx.TINY_DIMS, "construct": fx.CONSTRUCT_DIMS, "vessel": fx.VESSEL_DIMS}[dims_name]
    m = fx.ragged_mask(n_nodes)
    if dims_name == "construct":
        m[2, 3] = m[2, 11] = False                                             # a scattered mask as well
    B, n = m.shape
    model = GraphTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0)).eval()
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 5))
    model = model.cuda()
    g = torch.Generator().manual_seed(2)
    E = torch.randn(B, n, n, ind[1], generator=g)
    inp = model_input(torch.randn(B, n, ind[0], generator=g).cuda(), (0.5 * (E + E.transpose(1, 2))).cuda(),
                      torch.randn(B, ind[2], generator=g).cuda(), m.cuda())
    dev = torch.device("cuda", torch.cuda.current_device())
    model._ensure_engine(dev)
    model._workspace(B, n, dev)
    (ws,) = model._ws.values()
    outs = []
    for byte in (0, 0xFF, 0x7F, 0x3F):                                         # 0xFF..: NaN as float, -1 as a label
        ws.fill_(byte)
        out = model(inp)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(out.E).any() or torch.isnan(out.X).any())
        outs.append((out.E.clone(), out.X.clone()))
    for e, x in outs[1:]:
        assert torch.equal(e, outs[0][0]) and torch.equal(x, outs[0][1])


def _run_with_workspace_patterns(model, call, prepare):
    """call() under four byte patterns in the model's workspace -> all outputs equal, none NaN."""
    prepare()
    (ws,) = model._ws.values()
    outs = []
    for byte in (0, 0xFF, 0x7F, 0x3F):
        ws.fill_(byte)
        out = [t.clone() for t in call()]
        torch.cuda.synchronize()
        assert not any(bool(torch.isnan(t).any()) for t in out)
        outs.append(out)
    for out in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(out, outs[0]))


def test_3d_denoiser_and_sit_do_not_depend_on_what_the_workspace_held():
    """The same property for the position-aware denoiser (scattered mask, empty first chunk) and the SiT coordinate model."""
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.node_coordinate_diffusion import SiT_models
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(8)
    dims, ind, od, L = fx.VESSEL_DIMS, (1, 5, 1, 0), (1, 5, 0, 0), 3
    model = Graph3DTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(*od)).eval()
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=51))
    model = model.cuda()
    B, n = 5, 23
    mask = torch.rand(B, n, generator=g) < 0.6
    mask[0] = True
    mask[2, :16] = False
    mask[2, 16:] = True
    E = fx.one_hot_edges(fx.random_labels(mask, 5, 0.2, 9), mask, 5)
    inp = PlaceHolder(X=mask[..., None].float().to(dev), E=E.to(dev), y=torch.rand(B, 1, generator=g).to(dev),
                      pos=torch.randn(B, n, 3, generator=g).to(dev), node_mask=mask.to(dev),
                      radius=torch.zeros(B, n, n, 1, device=dev))

    def call3d():
        out = model(inp)
        return out.E, out.X, out.radius
    _run_with_workspace_patterns(model, call3d, call3d)

    sit = SiT_models["SiT-T"](in_channels=3)
    sit.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in sit.state_dict().items()}, seed=604))
    sit = sit.cuda()
    m = fx.ragged_mask([70, 33, 1, 12])
    x = torch.randn(*m.shape, 3, generator=g).to(dev)
    t = torch.tensor([999, 500, 1, 3]).to(dev)
    callsit = lambda: (sit(x, t, m.to(dev)),)
    _run_with_workspace_patterns(sit, callsit, callsit)
    m2 = fx.ragged_mask([30, 12, 1, 25])                                       # the matrix-pipe attention path (n <= 64)
    x2 = torch.randn(*m2.shape, 3, generator=g).to(dev)
    callsit2 = lambda: (sit(x2, t, m2.to(dev)),)
    _run_with_workspace_patterns(sit, callsit2, callsit2)


def test_reverse_step_inputs_and_logits_are_bit_reproducible():
    """The state route of one reverse step at the benchmark's dims (structural + eigen features on the side stream, then
    the forward) on a fixed state, 100 times."""
    import bench
    dev = torch.device("cuda:0")
    sampler, model = bench.build_sampler(500, dev)
    B = 64
    mask = fx.ragged_mask(bench.node_counts(B, 0), 32)
    lab = fx.random_labels(mask, 14, 0.08, 3)
    live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(32, dtype=torch.bool)[None]
    lab8 = torch.where(live, lab, torch.full_like(lab, -1)).to(torch.int8).to(dev)
    md = mask.to(dev)
    st = sampler._alloc_state(md)

    def once():
        out = model(sampler._build_inputs(st, lab8, md, 250))
        return out.E.clone(), st["X_in"].clone(), st["y_in"].clone(), st["E_in"].clone()
    ref = once()
    bad = 0
    for r in range(100):
        cur = once()
        bad += int(not all(torch.equal(a, b) for a, b in zip(cur, ref)))
    assert bad == 0, f"{bad} of 100 repeated steps differ from the first"
