"""GPU parity of the training step (row a11) against the reference fixture F10 (apply_noise + train-mode forward with dropout
0 + TrainLoss + backward, recorded from the real reference) and against the oracle's train-mode forward with dropout on."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import denoiser as D
from oracle import fixtures as fx
from oracle import training as TR
from test_host_logic import make_noise_model

pytestmark = pytest.mark.gpu


def close(a, b, rtol, atol):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def tiny_model(C, n_layers, seed):
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    model = GraphTransformer(n_layers, fx.dims_ns(1, C, 1), fx.TINY_DIMS["hidden_mlp_dims"], fx.TINY_DIMS["hidden_dims"], fx.dims_ns(1, C, 0))
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    return model.cuda(), sd


@pytest.mark.parametrize("name", ["f10_training_atm_tiny", "f10_training_cow_marg_tiny"])
def test_training_step_matches_reference(name):
    """z_t bit-exact, logits 1e-4, loss 1e-5, every parameter gradient 1e-3 (the fixture's own tolerance in the oracle test)."""
    from semantically_consistent_graph_generation_amd.extra_features import DummyExtraFeatures
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.training import training_step, TrainLoss
    g = load_golden(name)
    C = int(g["C"])
    m = torch.from_numpy(g["node_mask"])
    nm = make_noise_model(str(g["transition"]), C, T=500, top_k=4, e_marg=g["e_marginals"] if g["e_marginals"].size else None)
    model, sd = tiny_model(C, int(g["n_layers"]), int(g["sd_seed"]))
    model.train()
    lab0 = torch.from_numpy(g["labels0"]).long()
    dense = PlaceHolder(X=m[..., None].float().cuda(), charges=torch.zeros(*m.shape, 0, device="cuda"),
                        E=fx.one_hot_edges(lab0, m, C).cuda(), y=torch.zeros(m.shape[0], 0, device="cuda"), node_mask=m.cuda(),
                        pos=torch.from_numpy(g["pos"]).cuda())
    loss, pred, z_t = training_step(model, nm, DummyExtraFeatures(), dense, lab0.cuda(), loss_fn=TrainLoss((1.0, 2.0, 5.0, 0.0)),
                                    t_int=torch.from_numpy(g["t_int"]).cuda(), rng=Rng(u0=torch.from_numpy(g["u0"]).cuda()),
                                    dropout_seed=None)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(z_t.E.cpu().numpy(), g["zt_E"])
    close(pred.E, g["logits_E"], 1e-4, 1e-5)
    close(loss, g["loss"], 1e-5, 1e-6)
    checked = 0
    for k, p in model.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        if "grad/" + k in g:
            close(gr, g["grad/" + k], 1e-3, 2e-6)
            checked += 1
        elif "gradhead/" + k in g:
            close(gr.flatten()[:1024], g["gradhead/" + k], 1e-3, 2e-6)
            close(gr.double().norm(), g["gradstat/" + k][0], 1e-4, 1e-5)
            close(gr.double().sum(), g["gradstat/" + k][1], 1e-4, 2e-5 * float(g["gradstat/" + k][0]))   # a cancelling sum of 32768 terms
            checked += 1
    assert checked == len(list(model.named_parameters()))
    # eval mode on the same weights: the fused engine gives the same logits as the differentiable forward
    from semantically_consistent_graph_generation_amd.transformer_model import model_input
    model.eval()
    X = z_t.X.float()
    E = z_t.E.float()
    y = z_t.t.float()
    out = model(model_input(X, E, y, z_t.node_mask))
    close(out.E, pred.E.detach(), 1e-4, 2e-5)


def test_dropout_kernel_is_the_oracle_stream():
    """scgg_dropout == oracle/training.py:dropout_keep bit for bit (mask and scaling), forward and backward."""
    from semantically_consistent_graph_generation_amd.training import DropoutSites
    x = torch.randn(7, 33, 5, generator=torch.Generator().manual_seed(2))
    for p, seed, site in ((0.1, 12345, 0), (0.5, (7 << 32) | 9, 117), (0.9, 1, 52)):
        d = DropoutSites(p, seed)
        d.site = site
        xc = x.cuda().requires_grad_(True)
        out = d(xc)
        out.sum().backward()
        keep = torch.from_numpy(TR.dropout_keep(seed, site, x.numel(), p)).reshape(x.shape)
        want = x * keep.to(x.dtype) / (1.0 - p)
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want.numpy())
        np.testing.assert_array_equal(xc.grad.cpu().numpy(), (torch.ones_like(x) * keep / (1.0 - p)).numpy())
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.05


def test_train_forward_with_dropout_matches_oracle():
    """Train-mode forward with the reference's layer dropout (p = 0.1) and in / out dropout on, seeded: the device masks
    are the oracle's, so the logits agree to the fp32 tolerance; the gradients through the dropped network agree too."""
    from semantically_consistent_graph_generation_amd.training import train_forward, masked_cross_entropy
    C, n_layers, seed = 5, 2, 99
    model, sd = tiny_model(C, n_layers, 21)
    model.dropout, model.dropout_in_and_out = 0.2, True
    model.train()
    m = fx.ragged_mask([9, 4, 11])
    B, n = m.shape
    gen = torch.Generator().manual_seed(3)
    lab = fx.random_labels(m, C, 0.3, 4)
    E = fx.one_hot_edges(lab, m, C)
    X = m[..., None].float()
    y = torch.rand(B, 1, generator=gen)
    pred = train_forward(model, X.cuda(), E.cuda(), y.cuda(), m.cuda(), seed=seed)
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tgt = torch.where(live, lab, torch.full_like(lab, -1))
    loss = masked_cross_entropy(pred.E, tgt.cuda())
    loss.backward()
    dd = D.DenoiserDims(n_layers, fx.TINY_DIMS["hidden_dims"]["n_head"], 1, C, 0)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oX, _, oE, _ = TR.graph_transformer_train(leaf, dd, X, E, y, m, p_layer=0.1, p_io=0.2, seed=seed)
    oloss = torch.nn.functional.cross_entropy(oE[live], lab[live], reduction="sum") / int(live.sum())
    oloss.backward()
    close(pred.E, oE.detach(), 1e-4, 1e-5)
    close(loss, oloss.detach(), 1e-5, 1e-6)
    for k, p in model.named_parameters():
        og = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])
        close(p.grad if p.grad is not None else torch.zeros_like(p), og, 2e-3, 5e-6)
    # and the masks matter: another seed gives other logits
    pred2 = train_forward(model, X.cuda(), E.cuda(), y.cuda(), m.cuda(), seed=seed + 1)
    assert float((pred2.E - pred.E).abs().max()) > 1e-3


def test_masked_cross_entropy_kernel_against_torch():
    from semantically_consistent_graph_generation_amd.training import masked_cross_entropy
    gen = torch.Generator().manual_seed(5)
    for rows, C in ((1, 1), (300, 14), (5000, 5)):
        z = (3 * torch.randn(rows, C, generator=gen)).cuda().requires_grad_(True)
        t = torch.randint(-1, C, (rows,), generator=gen)
        loss = masked_cross_entropy(z, t.cuda())
        (2.5 * loss).backward()
        zc = z.detach().cpu().double().requires_grad_(True)
        keep = t >= 0
        ref = torch.nn.functional.cross_entropy(zc[keep], t[keep], reduction="sum") / max(int(keep.sum()), 1)
        (2.5 * ref).backward()
        close(loss, ref.detach(), 1e-6, 1e-6)
        close(z.grad, zc.grad, 1e-5, 1e-7)


def test_training_step_with_extra_features_runs_and_learns():
    """The full step at CoW widths with extra features on: a few Adam steps on one batch lower the loss (smoke for the
    pieces working together: apply_noise -> features -> train forward -> loss -> backward -> engine refresh in eval)."""
    from types import SimpleNamespace
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.training import training_step
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    C = 14
    cfg = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5, num_eigenvalues=9,
                                                eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=16))
    in_dims = ef.update_input_dims(fx.dims_ns(1, C, 1))
    torch.manual_seed(0)
    model = GraphTransformer(2, in_dims, fx.TINY_DIMS["hidden_mlp_dims"], fx.TINY_DIMS["hidden_dims"], fx.dims_ns(1, C, 0)).cuda().train()
    nm = make_noise_model("absorbing_edges", C, T=50, top_k=4)
    m = fx.ragged_mask([16, 9, 12, 5])
    lab0 = fx.random_labels(m, C, 0.15, 8)
    dense = PlaceHolder(X=m[..., None].float().cuda(), charges=torch.zeros(*m.shape, 0, device="cuda"), E=fx.one_hot_edges(lab0, m, C).cuda(),
                        y=torch.zeros(m.shape[0], 0, device="cuda"), node_mask=m.cuda(), pos=None)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    losses = []
    for it in range(12):
        opt.zero_grad()
        loss, _, _ = training_step(model, nm, ef, dense, lab0.cuda(), t_int=torch.full((4, 1), 25, device="cuda"),
                                   rng=Rng(seed=3, step=0), dropout_seed=100 + it)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses


def test_training_step_of_the_3d_class_matches_reference():
    """G10: Graph3DTransformer in training mode (dropout off) + TrainLoss with the radius term + backward against the
    fixture recorded from the reference: logits / radius 1e-4, loss 1e-5, every gradient 1e-3, and exactly the parameters the
    reference leaves without a gradient get none here; the train-mode forward also equals the fused eval engine; dropout on
    changes the result and is reproducible from the seed."""
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.pos_transformer_model import Graph3DTransformer
    from semantically_consistent_graph_generation_amd.training import TrainLoss
    g = load_golden("g10_training_3d_tiny")
    C, dims = int(g["C"]), fx.TINY_DIMS
    m = torch.from_numpy(g["node_mask"])
    B, n = m.shape
    model = Graph3DTransformer(int(g["n_layers"]), fx.dims_ns(1, C, 1), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0))
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, int(g["sd_seed"])))
    model = model.cuda().train()
    lab0, lab_t = torch.from_numpy(g["labels0"]).long(), torch.from_numpy(g["labels_t"]).long()
    inp = PlaceHolder(X=m[..., None].float().cuda(), E=fx.one_hot_edges(lab_t, m, C).cuda(), y=torch.from_numpy(g["t"]).cuda(),
                      pos=torch.from_numpy(g["pos"]).cuda(), radius=torch.zeros(B, n, n, 1, device="cuda"), node_mask=m.cuda())
    # through training_step: it recognises the position-aware classes, hands pos on and adds the radius term.
    # The fixture gives the noisy state itself (apply_noise is pinned by F10), so the noise model here only returns it.
    from types import SimpleNamespace
    from semantically_consistent_graph_generation_amd.extra_features import DummyExtraFeatures
    from semantically_consistent_graph_generation_amd.training import training_step
    z_in = inp.copy()
    z_in.charges, z_in.t, z_in.y = torch.zeros(B, n, 0, device="cuda"), inp.y, torch.zeros(B, 0, device="cuda")
    dense = PlaceHolder(X=m[..., None].float().cuda(), E=fx.one_hot_edges(lab0, m, C).cuda(), y=torch.zeros(B, 0, device="cuda"),
                        pos=inp.pos, radius=torch.from_numpy(g["radius0"]).cuda(), node_mask=m.cuda())
    loss, pred, _ = training_step(model, SimpleNamespace(apply_noise=lambda d, t_int=None, rng=None: z_in), DummyExtraFeatures(), dense,
                                  lab0.cuda(), loss_fn=TrainLoss((1.0, 2.0, 5.0, 0.0)))
    loss.backward()
    torch.cuda.synchronize()
    close(pred.E, g["logits_E"], 1e-4, 1e-5)
    close(pred.radius, g["radius"], 1e-4, 1e-5)
    close(loss, g["loss"], 1e-5, 1e-6)
    unused = set(str(k) for k in g["unused"])
    checked = 0
    for k, p in model.named_parameters():
        if k in unused:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if "grad/" + k in g:
            close(p.grad, g["grad/" + k], 1e-3, 2e-6)
            checked += 1
        else:
            close(p.grad.flatten()[:1024], g["gradhead/" + k], 1e-3, 2e-6)
            nrm, sm = g["gradstat/" + k]
            close(p.grad.double().norm(), nrm, 1e-4, 1e-8)
            close(p.grad.double().sum(), sm, 1e-3, 1e-4 * float(nrm))
            checked += 1
    assert checked > 60
    # the same weights through the fused eval engine
    model.eval()
    ev = model(inp)
    close(ev.E, pred.E, 1e-4, 1e-5)
    close(ev.radius, pred.radius, 1e-4, 1e-5)
    # dropout: reproducible from the seed, different from the no-dropout forward
    model.train()
    model.dropout_seed = 77
    with torch.no_grad():
        a, b = model(inp), model(inp)
        model.dropout_seed = 78
        c = model(inp)
    assert torch.equal(a.E, b.E) and not torch.equal(a.E, c.E)
    assert float((a.E - pred.E.detach()).abs().max()) > 1e-3


def test_training_step_at_config3_shape():
    """BASELINE configs[2] at its own size: ATM-shaped trees of 192..256 nodes, B = 4, vessel dims (dx 128, de 64, 4 heads),
    5 layers, C = 5, extra features on (the tiled adjacency pass + the eigen features beyond 32 nodes), layer dropout 0.1.
    (1) one step with dropout: finite loss, every parameter the path uses gets a finite, non-zero gradient and exactly the
    last layer's y modules get none; (2) dropout off: train-forward logits == the fused eval engine's on the same inputs at
    1e-4 of the scale; (3) logits and loss of a 2-graph slice against oracle/training.py (CPU, same inputs, dropout 0)."""
    from types import SimpleNamespace
    from semantically_consistent_graph_generation_amd import noise_model as NM
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.training import training_step, train_forward, TrainLoss
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
    C, n, B, L = 5, 256, 4, 5
    dims = fx.VESSEL_DIMS if hasattr(fx, "VESSEL_DIMS") else dict(
        hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
        hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
    nu = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)
    cfg = SimpleNamespace(model=SimpleNamespace(nu=nu, transition="absorbing_edges", diffusion_steps=500, top_k=4, num_degree=10,
                                                max_degree=10, cycle_features=True, num_eigenvectors=5, num_eigenvalues=9,
                                                eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=n))
    in_dims = ef.update_input_dims(fx.dims_ns(1, C, 1))
    model = GraphTransformer(L, in_dims, dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0))
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 311)
    model.load_state_dict(sd)
    model = model.cuda().train()
    nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
    g = torch.Generator().manual_seed(7)
    n_nodes = [256, 192, 230, 201]
    m = fx.ragged_mask(n_nodes)
    lab = torch.zeros(B, n, n, dtype=torch.long)
    for b, nb in enumerate(n_nodes):                       # random trees with short-range parents (airway-like)
        idx = torch.arange(1, nb)
        par = (torch.rand(nb - 1, generator=g) * idx).long()
        cls = torch.randint(1, C, (nb - 1,), generator=g)
        lab[b, idx, par] = cls
        lab[b, par, idx] = cls
    dense = PlaceHolder(X=m[..., None].float().cuda(), charges=torch.zeros(B, n, 0, device="cuda"), E=fx.one_hot_edges(lab, m, C).cuda(),
                        y=torch.zeros(B, 0, device="cuda"), node_mask=m.cuda(), pos=None)
    t_int = torch.tensor([[250], [40], [470], [120]], device="cuda")
    # (1) the step as it trains
    loss, pred, z_t = training_step(model, nm, ef, dense, lab.cuda(), loss_fn=TrainLoss((1.0, 2.0, 5.0, 0.0)), t_int=t_int,
                                    rng=Rng(seed=5), dropout_seed=1234)
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and float(loss) > 0
    last = f"tf_layers.{L - 1}."
    for k, p in model.named_parameters():
        # y never reaches the loss from the last layer on (output_dims.y = 0), and X only reaches the node logits, which a
        # single node class makes constant: the reference's DDP needs find_unused_parameters for the same set
        feeds_nothing = (k.startswith(last) and any(t in k for t in ("y_y", "x_y", "e_y", "y_out", "lin_y", "norm_y")))
        if p.grad is None:
            assert feeds_nothing, k
            continue
        assert torch.isfinite(p.grad).all(), k
        if not feeds_nothing and not k.startswith("mlp_out_X") and not any(
                k.startswith(last + t) for t in ("linX", "normX", "self_attn.x_out", "self_attn.y_x", "self_attn.v")):
            assert float(p.grad.abs().max()) > 0, k
    model.zero_grad(set_to_none=True)
    # (2) dropout off == the fused eval engine
    extra = ef(z_t)
    X = torch.cat((z_t.X, z_t.charges, extra.X), dim=2).float()
    E = torch.cat((z_t.E, extra.E), dim=3).float()
    y = torch.hstack((z_t.y, extra.y, z_t.t)).float()
    with torch.no_grad():
        tr = train_forward(model, X, E, y, z_t.node_mask, seed=None)
        model.eval()
        ev = model(model_input(X, E, y, z_t.node_mask))
        model.train()
    scale = float(tr.E.abs().max())
    assert float((ev.E - tr.E).abs().max()) / scale < 1e-4
    # (3) a 2-graph slice against the CPU oracle, dropout 0
    sl = slice(1, 3)
    dd = D.DenoiserDims(L, dims["hidden_dims"]["n_head"], 1, C, 0)
    oX, _, oE, _ = TR.graph_transformer_train({k: v for k, v in sd.items()}, dd, X[sl].cpu(), E[sl].cpu(), y[sl].cpu(), m[sl],
                                              p_layer=0.0, p_io=0.0, seed=None)
    with torch.no_grad():
        tr2 = train_forward(model, X[sl].contiguous(), E[sl].contiguous(), y[sl].contiguous(), z_t.node_mask[sl].contiguous(), seed=None)
        l2 = TrainLoss((1.0, 2.0, 5.0, 0.0))(tr2, torch.zeros(2, n, dtype=torch.long, device="cuda"), lab[sl].cuda(), m[sl].cuda())
    assert float((tr2.E.cpu() - oE).abs().max()) / float(oE.abs().max()) < 1e-4
    live = (m[sl][:, :, None] & m[sl][:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    want = 5.0 * torch.nn.functional.cross_entropy(oE[live], lab[sl][live], reduction="sum") / int(live.sum())
    close(l2, want, 1e-4, 1e-5)


@pytest.mark.parametrize("dims_name,n_nodes,seed", [("tiny", [9, 4, 11, 17], None), ("tiny", [9, 4, 11, 17], 99),
                                                    ("vessel", [40, 23, 33], 7), ("construct", [20, 32], 5)])
def test_fused_edge_layer_matches_the_unfused_forward_and_backward(dims_name, n_nodes, seed, monkeypatch):
    """Row a11: the train-mode forward through the fused edge kernels and their hand-written backward (csrc/train_edge.hip)
    against the unfused tensor-op path (SCGG_TRAIN_MODE=eager, itself pinned by the reference fixture F10): logits 1e-5 of
    the scale, loss 1e-6, every parameter gradient 2e-4 of its scale -- with and without dropout (same Philox masks)."""
    from semantically_consistent_graph_generation_amd.training import train_forward, masked_cross_entropy
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    C, L = 5, 3
    dims = {"tiny": fx.TINY_DIMS,
            "vessel": dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
                           hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128}),
            "construct": dict(hidden_mlp_dims={"X": 256, "E": 128, "y": 128},
                              hidden_dims={"dx": 256, "de": 64, "dy": 64, "n_head": 8, "dim_ffX": 256, "dim_ffE": 128, "dim_ffy": 128})}[dims_name]
    model = GraphTransformer(L, fx.dims_ns(3, C + 2, 4), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0))
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 77))
    model = model.cuda().train()
    m = fx.ragged_mask(n_nodes)
    B, n = m.shape
    gen = torch.Generator().manual_seed(13)
    lab = fx.random_labels(m, C, 0.3, 4)
    E = torch.cat([fx.one_hot_edges(lab, m, C), torch.randn(B, n, n, 2, generator=gen)], dim=-1)
    E = 0.5 * (E + E.transpose(1, 2))
    X = torch.randn(B, n, 3, generator=gen) * m[..., None]
    y = torch.rand(B, 4, generator=gen)
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tgt = torch.where(live, lab, torch.full_like(lab, -1)).cuda()
    out = {}
    for mode in ("eager", "fused"):
        monkeypatch.setenv("SCGG_TRAIN_MODE", mode)
        model.zero_grad(set_to_none=True)
        pred = train_forward(model, X.cuda(), E.cuda(), y.cuda(), m.cuda(), seed=seed)
        loss = masked_cross_entropy(pred.E, tgt) + 0.3 * (pred.X ** 2).mean()      # the X term reaches wV / newX in every layer
        loss.backward()
        torch.cuda.synchronize()
        out[mode] = (pred.E.detach().clone(), float(loss), {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()})
    sc = float(out["eager"][0].abs().max())
    assert float((out["fused"][0] - out["eager"][0]).abs().max()) / sc < 1e-5
    assert abs(out["fused"][1] - out["eager"][1]) < 1e-6 * max(1.0, abs(out["eager"][1]))
    worst = ("", 0.0)
    for k, ge in out["eager"][2].items():
        gf = out["fused"][2][k]
        assert (ge is None) == (gf is None), k
        if ge is None:
            continue
        err = float((gf - ge).abs().max()) / max(float(ge.abs().max()), 1e-12)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 2e-4, worst


def test_fused_training_step_does_not_depend_on_what_freed_memory_held():
    """The fused forward / backward allocate their factor tensors and partials with torch.empty: after other work on the
    device those blocks hold old data (NaN patterns, -1 labels).  Same step, fresh memory vs such leftovers: every gradient
    bit for bit the same."""
    from semantically_consistent_graph_generation_amd.training import train_forward, masked_cross_entropy
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    C, L = 5, 2
    dims = dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
                hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
    model = GraphTransformer(L, fx.dims_ns(3, C + 2, 4), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0))
    model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 77))
    model = model.cuda().train()
    m = fx.ragged_mask([40, 23, 33, 1])
    m[0, 5] = m[0, 17] = False                                                 # masked nodes inside a graph
    B, n = m.shape
    gen = torch.Generator().manual_seed(13)
    lab = fx.random_labels(m, C, 0.3, 4)
    E = torch.cat([fx.one_hot_edges(lab, m, C), torch.randn(B, n, n, 2, generator=gen)], dim=-1)
    E = (0.5 * (E + E.transpose(1, 2))).cuda()
    X = (torch.randn(B, n, 3, generator=gen) * m[..., None]).cuda()
    y = torch.rand(B, 4, generator=gen).cuda()
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tgt = torch.where(live, lab, torch.full_like(lab, -1)).cuda()

    def step():
        model.zero_grad(set_to_none=True)
        pred = train_forward(model, X, E, y, m.cuda(), seed=5)
        loss = masked_cross_entropy(pred.E, tgt) + 0.3 * (pred.X ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    def dirty(byte):
        xs = [torch.full((sz,), byte, dtype=torch.uint8, device="cuda") for _ in range(40)
              for sz in (300, 5000, 70000, 1 << 20, 1 << 22, 1 << 24, 1 << 26)]
        torch.cuda.synchronize()
        del xs

    ref_loss, ref = step()
    for byte in (0xFF, 0x7F, 0x01):
        dirty(byte)
        loss, grads = step()
        assert loss == ref_loss
        for k, g in grads.items():
            assert not bool(torch.isnan(g).any()), k
            assert torch.equal(g, ref[k]), k


@pytest.mark.parametrize("A,R,Cc", [(2, 256, 65536), (2, 65536, 64), (3, 100, 20), (1, 7, 4), (4, 1000, 260)])
def test_sum_rows_kernel_against_torch(A, R, Cc):
    from semantically_consistent_graph_generation_amd.training import _sum_rows
    t = torch.randn(A, R, Cc, generator=torch.Generator().manual_seed(R)).cuda()
    got = _sum_rows(t)
    torch.cuda.synchronize()
    want = t.double().sum(1)
    assert float((got.double() - want).abs().max()) / float(want.abs().max()) < 2e-6
    assert torch.equal(got, _sum_rows(t))


def test_sum_rows_into_a_slice_of_a_larger_buffer():
    from semantically_consistent_graph_generation_amd.training import _sum_rows
    t = torch.randn(2, 300, 64, generator=torch.Generator().manual_seed(5)).cuda()
    out = torch.full((3, 5, 64), float("nan"), device="cuda")
    _sum_rows(t, out=out[:, 1:3], splits=3)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[:, 0]).all()) and bool(torch.isnan(out[:, 3:]).all())
    want = t.double().sum(1)
    assert float((out[:, 1:3].double().sum(0) - want).abs().max()) / float(want.abs().max()) < 2e-6


def test_wgrad_products_share_one_partial_buffer():
    """Two products, two chunks of slots each, into one [rows, ld] buffer; one sum over the rows gives both gradients."""
    from semantically_consistent_graph_generation_amd import _lib
    from semantically_consistent_graph_generation_amd.training import _wgrad, _wgrad_into, _sum_rows
    g = torch.Generator().manual_seed(3)
    S, slab = 5000, 256
    Ga, Xa = torch.randn(2 * S, 128, generator=g).cuda(), torch.randn(2 * S, 64, generator=g).cuda()
    Gb, Xb = torch.randn(2 * S, 64, generator=g).cuda(), torch.randn(2 * S, 32, generator=g).cuda()
    wa, wb = 128 * 64 + 128, 64 * 32 + 64
    nsl = int(_lib.lib().scgg_train_wgrad_slabs(S, slab))
    assert nsl == (S + slab - 1) // slab
    P = torch.full((2 * nsl, wa + wb), float("nan"), device="cuda")
    for c in range(2):
        _wgrad_into(Ga[c * S:(c + 1) * S], Xa[c * S:(c + 1) * S], P, c * nsl, 0, slab)
        _wgrad_into(Gb[c * S:(c + 1) * S], Xb[c * S:(c + 1) * S], P, c * nsl, wa, slab)
    tot = _sum_rows(P.view(1, 2 * nsl, wa + wb))[0]
    torch.cuda.synchronize()
    for G_, X_, c0, M, N in ((Ga, Xa, 0, 128, 64), (Gb, Xb, wa, 64, 32)):
        dW, db = tot[c0:c0 + M * N].view(M, N), tot[c0 + M * N:c0 + M * N + M]
        want_W, want_b = G_.double().t() @ X_.double(), G_.double().sum(0)
        assert float((dW.double() - want_W).abs().max()) / float(want_W.abs().max()) < 1e-5
        assert float((db.double() - want_b).abs().max()) / float(want_b.abs().max()) < 1e-5
        dW1, db1 = _wgrad(G_, X_)                                      # the one-product helper agrees
        assert float((dW1 - dW).abs().max()) / float(want_W.abs().max()) < 1e-5
        assert float((db1 - db).abs().max()) / float(want_b.abs().max()) < 1e-5


def test_pack_layer_matches_single_packs():
    from semantically_consistent_graph_generation_amd.training import _pack, _pack_layer
    g = torch.Generator().manual_seed(2)
    Ws = [torch.randn(r, c, generator=g).cuda() for r, c in ((128, 64), (128, 64), (64, 128), (128, 64), (64, 128))]
    fwd, bwd = _pack_layer(Ws)
    for W, f, b in zip(Ws, fwd, bwd):
        assert torch.equal(f, _pack(W, False)) and torch.equal(b, _pack(W, True))


def test_edge_stats_finish_matches_the_tensor_op_merge():
    from semantically_consistent_graph_generation_amd import _lib
    B, n, de, slices = 3, 40, 64, 8
    g = torch.Generator().manual_seed(9)
    E = torch.randn(B, n, n, de, generator=g).cuda()
    mask = (torch.arange(n)[None] < torch.tensor([40, 33, 17])[:, None]).cuda()
    m8 = mask.view(torch.uint8).contiguous()
    L = _lib.lib()
    so = torch.empty(B, slices, 5, de, device="cuda")
    si = torch.empty(B, slices, 2, de, dtype=torch.int32, device="cuda")
    _lib.check(L.scgg_train_edge_stats(_lib.ptr(E), _lib.ptr(m8), B, n, de, slices, _lib.ptr(so), _lib.ptr(si), _lib.stream_ptr()), "stats")
    stats = torch.empty(B, 4 * de, device="cuda")
    mean, cnt = torch.empty(B, de, device="cuda"), torch.empty(B, de, device="cuda")
    imn, imx = torch.empty(B, de, dtype=torch.int32, device="cuda"), torch.empty(B, de, dtype=torch.int32, device="cuda")
    _lib.check(L.scgg_train_edge_stats_finish(_lib.ptr(so), _lib.ptr(si), B, slices, de, _lib.ptr(stats), _lib.ptr(mean), _lib.ptr(cnt),
                                              _lib.ptr(imn), _lib.ptr(imx), _lib.stream_ptr()), "finish")
    torch.cuda.synchronize()
    pm = (mask[:, :, None] & mask[:, None, :])
    for b in range(B):
        v = E[b][pm[b]].double()                                       # [pairs, de]
        want = torch.cat([v.mean(0), v.min(0)[0], v.max(0)[0], v.var(0, unbiased=False)])
        assert float((stats[b].double() - want).abs().max()) < 1e-5
        assert float(cnt[b, 0]) == v.shape[0]
        flat = E[b].reshape(n * n, de)
        assert torch.equal(flat.gather(0, imn[b].long()[None])[0], stats[b, de:2 * de])
        assert torch.equal(flat.gather(0, imx[b].long()[None])[0], stats[b, 2 * de:3 * de])


def test_dropout_key_from_device_memory_gives_the_same_step():
    """The dropout kernels can read their Philox key from device memory at run time (scgg_dropout_devkey,
    scgg_train_edge_args.key_dev, scgg_train_y_args.key_dev): the same masks, hence the same loss and gradients, as the key
    passed by value."""
    from types import SimpleNamespace
    from semantically_consistent_graph_generation_amd.training import train_forward
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    ns = lambda X, E, y: SimpleNamespace(X=X, E=E, y=y, charges=0)
    torch.manual_seed(5)
    dims = dict(hidden_mlp_dims={"X": 64, "E": 32, "y": 32},
                hidden_dims={"dx": 64, "de": 32, "dy": 32, "n_head": 4, "dim_ffX": 64, "dim_ffE": 64, "dim_ffy": 64})
    model = GraphTransformer(2, ns(9, 5, 12), dims["hidden_mlp_dims"], dims["hidden_dims"], ns(1, 5, 0)).cuda().train()
    B, n = 3, 21
    g = torch.Generator().manual_seed(1)
    mask = (torch.arange(n)[None] < torch.tensor([21, 9, 14])[:, None]).cuda()
    X = torch.randn(B, n, 9, generator=g).cuda()
    E = torch.randn(B, n, n, 5, generator=g).cuda()
    E = 0.5 * (E + E.transpose(1, 2))
    y = torch.randn(B, 12, generator=g).cuda()
    out = []
    seed = (7 << 32) | 12345
    for sd in (seed, torch.full((1,), seed, dtype=torch.int64, device="cuda")):
        model.zero_grad(set_to_none=True)
        pred = train_forward(model, X, E, y, mask, seed=sd)
        (pred.E.square().mean() + pred.X.square().mean()).backward()
        out.append((pred.E.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}))
    assert torch.equal(out[0][0], out[1][0])
    for k, gq in out[0][1].items():
        assert torch.equal(gq, out[1][1][k]), k


@pytest.mark.parametrize("Cc", [64, 5])
def test_sym_mask_kernel_equals_the_tensor_ops_it_replaces(Cc):
    from semantically_consistent_graph_generation_amd.training import _SymMask
    B, n = 3, 37
    g = torch.Generator().manual_seed(Cc)
    R = torch.randn(B, n, n, Cc, generator=g).cuda().requires_grad_(True)
    mask = (torch.arange(n)[None] < torch.tensor([37, 20, 1])[:, None]).cuda()
    m = mask.float()
    w = (m[:, :, None] * m[:, None, :] * (1.0 - torch.eye(n, device="cuda"))[None])[..., None]
    want = (0.5 * (R + R.transpose(1, 2))) * w
    got = _SymMask.apply(R, mask.view(torch.uint8).contiguous())
    assert torch.equal(got, want)
    go = torch.randn(B, n, n, Cc, generator=g).cuda()
    (gw,) = torch.autograd.grad(want, R, go, retain_graph=True)
    (gg,) = torch.autograd.grad(got, R, go)
    assert torch.equal(gg, gw)


def test_node_stats_kernel_matches_the_tensor_ops():
    from semantically_consistent_graph_generation_amd.training import _NodeStats, _node_stats
    B, n, dx = 3, 45, 128
    g = torch.Generator().manual_seed(8)
    mask = (torch.arange(n)[None] < torch.tensor([45, 17, 2])[:, None]).cuda()
    m = mask.float()
    X = (torch.randn(B, n, dx, generator=g).cuda() * m[:, :, None]).requires_grad_(True)
    want = _node_stats(X, m)
    got = _NodeStats.apply(X, mask.view(torch.uint8).contiguous())
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    go = torch.randn(B, 4 * dx, generator=g).cuda()
    (gw,) = torch.autograd.grad(want, X, go, retain_graph=True)
    (gg,) = torch.autograd.grad(got, X, go)
    assert float((gg - gw).abs().max()) <= 2e-6 * float(gw.abs().max())


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_y_round_kernels_match_the_tensor_ops(p, monkeypatch):
    """The y path of a layer as three kernels against the same layer through tensor ops under autograd (SCGG_TRAIN_Y=ops):
    logits and every gradient of a two-layer network at vessel widths, dim_ffy = 2048 as the reference hard-wires it."""
    from types import SimpleNamespace
    from semantically_consistent_graph_generation_amd.training import train_forward
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    ns = lambda X, E, y: SimpleNamespace(X=X, E=E, y=y, charges=0)
    torch.manual_seed(3)
    dims = dict(hidden_mlp_dims={"X": 64, "E": 32, "y": 32},
                hidden_dims={"dx": 64, "de": 32, "dy": 64, "n_head": 4, "dim_ffX": 64, "dim_ffE": 64, "dim_ffy": 2048})
    model = GraphTransformer(3, ns(9, 5, 12), dims["hidden_mlp_dims"], dims["hidden_dims"], ns(1, 5, 0)).cuda().train()
    B, n = 5, 21
    g = torch.Generator().manual_seed(1)
    mask = (torch.arange(n)[None] < torch.tensor([21, 20, 9, 2, 14])[:, None]).cuda()
    X = torch.randn(B, n, 9, generator=g).cuda()
    E = torch.randn(B, n, n, 5, generator=g).cuda()
    E = 0.5 * (E + E.transpose(1, 2))
    y = torch.randn(B, 12, generator=g).cuda()
    out = {}
    for mode in ("fused", "ops"):
        monkeypatch.setenv("SCGG_TRAIN_Y", mode)
        model.zero_grad(set_to_none=True)
        pred = train_forward(model, X, E, y, mask, seed=123 if p > 0 else None, p_layer=p)
        (pred.E.square().mean() + pred.X.square().mean()).backward()
        out[mode] = (pred.E.detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None})
    assert float((out["fused"][0] - out["ops"][0]).abs().max()) <= 1e-5 * float(out["ops"][0].abs().max())
    assert out["fused"][1].keys() == out["ops"][1].keys()
    for k, gq in out["ops"][1].items():
        scale = float(gq.abs().max())
        assert float((out["fused"][1][k] - gq).abs().max()) <= 2e-4 * max(scale, 1e-8), k
    assert any("lin_y1" in k for k in out["ops"][1])
