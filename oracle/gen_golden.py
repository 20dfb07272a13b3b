"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

TEST INFRASTRUCTURE; runs only in the build container (the reference never travels to the GPU box).
Usage:  python -m oracle.gen_golden            (from the repo root)

Third-party modules that the reference imports at module level but that are absent here
(pyvista, wandb, omegaconf, torch_geometric, torchmetrics) are registered as empty
stubs before the import; the node-coordinate DDPM that NoiseModel.__init__ hard-loads
(noise_model.py:62-79, checkpoint not in the repo) is replaced by a stub class.  None of that is on
the arithmetic path being pinned.

torch.multinomial has no uniform input, so inside this process ``Tensor.multinomial`` is replaced by
the documented inverse-CDF rule (oracle.noise.inverse_cdf_draws) fed from a recorded uniform stream;
all code around it (masking to uniform, slot-0 overwrite, triu, mirror, one-hot, .mask()) is the
reference's own.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

from oracle import fixtures as fx
from oracle.noise import inverse_cdf_draws

REF = "/root/reference"
OUT = os.environ.get("SCGG_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    stub("pyvista")
    stub("wandb", run=None)
    stub("omegaconf", OmegaConf=_Dummy, DictConfig=_Dummy)
    tg = stub("torch_geometric")
    tg.utils = stub("torch_geometric.utils", to_dense_adj=None, to_dense_batch=None, remove_self_loops=None)
    stub("torchmetrics", Metric=_Dummy, MeanSquaredError=_Dummy, MeanAbsoluteError=_Dummy,
         MetricCollection=_Dummy, KLDivergence=_Dummy)

    class _CoordModel:
        def __init__(self, device=None):
            self.model = SimpleNamespace(load_state_dict=lambda *_a, **_k: None, eval=lambda: None)

        def to_cuda_device(self, *_a):
            pass

        def sample(self, pos, node_mask):
            return pos * node_mask[..., None]

    for ds in ("cow", "atm"):
        stub(f"src.bootstrap.node_coordinate_diffusion.{ds}_diff_nodes_with_3d_checks",
             NodeCoordsDiffusionModel=_CoordModel)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import src.ConStruct.utils as U
    from src.ConStruct.models import transformer_model as TM
    from src.ConStruct.models import layers as LY
    from src.ConStruct.diffusion import diffusion_utils as DU, extra_features as EF, noise_model as NM
    return SimpleNamespace(U=U, TM=TM, LY=LY, DU=DU, EF=EF, NM=NM)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(sd, prefix="sd/"):
    return {prefix + k: v for k, v in sd.items()}


# --------------------------------------------------------------------------------------------
def gen_block_and_layer(R):
    """F1/F2: one NodeEdgeBlock and one XEyTransformerLayer (eval) at tiny dims, ragged batch."""
    torch.manual_seed(11)
    dx, de, dy, h = 32, 32, 16, 4
    mask = fx.ragged_mask([7, 4, 1], 7)
    B, n = mask.shape
    X = torch.randn(B, n, dx)
    E = torch.randn(B, n, n, de)
    E = 0.5 * (E + E.transpose(1, 2))
    y = torch.randn(B, dy)
    ph = R.U.PlaceHolder(X=X, E=E, y=y, node_mask=mask).mask()
    X, E = ph.X, ph.E
    for predicts_y in (True, False):
        blk = R.TM.NodeEdgeBlock(dx, de, dy, h, predicts_y=predicts_y).eval()
        sd = fx.make_state_dict({k: v.shape for k, v in blk.state_dict().items()}, seed=21)
        blk.load_state_dict(sd)
        with torch.no_grad():
            nX, nE, ny = blk(X, E, y, mask)
        save(f"f1_node_edge_block_y{int(predicts_y)}", X=X, E=E, y=y, node_mask=mask, n_head=h,
             newX=nX, newE=nE, new_y=ny if ny is not None else np.zeros(0), **sd_arrays(sd))
    lay = R.TM.XEyTransformerLayer(dx, de, dy, h, dim_ffX=40, dim_ffE=64, dim_ffy=24).eval()
    sd = fx.make_state_dict({k: v.shape for k, v in lay.state_dict().items()}, seed=22)
    lay.load_state_dict(sd)
    with torch.no_grad():
        out = lay(R.U.PlaceHolder(X=X, E=E, y=y, node_mask=mask))
    save("f2_xey_layer", X=X, E=E, y=y, node_mask=mask, n_head=h, outX=out.X, outE=out.E, outy=out.y,
         **sd_arrays(sd))


def _gt_case(R, name, dims, in_dims, out_dims, n_layers, n_nodes, seed, store_weights, symmetric_E=True,
             onehot_C=None):
    torch.manual_seed(seed)
    mask = fx.ragged_mask(n_nodes)
    B, n = mask.shape
    model = R.TM.GraphTransformer(n_layers=n_layers, input_dims=fx.dims_ns(*in_dims),
                                  hidden_mlp_dims=dims["hidden_mlp_dims"], hidden_dims=dims["hidden_dims"],
                                  output_dims=fx.dims_ns(*out_dims), dropout=0.1).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = fx.make_state_dict(shapes, seed=seed)
    model.load_state_dict(sd)
    X = torch.randn(B, n, in_dims[0])
    E = torch.randn(B, n, n, in_dims[1])
    if symmetric_E:
        E = 0.5 * (E + E.transpose(1, 2))
    if onehot_C is not None:  # realistic sampling input: masked one-hot labels in the first C channels
        lab = fx.random_labels(mask, onehot_C, 0.15, seed)
        E[..., :onehot_C] = fx.one_hot_edges(lab, mask, onehot_C)
    y = torch.randn(B, in_dims[2])
    with torch.no_grad():
        out = model(SimpleNamespace(X=X, E=E, y=y, node_mask=mask))
    arrays = dict(X=X, E=E, y=y, node_mask=mask, outX=out.X, outE=out.E,
                  outy=out.y if out.y is not None else np.zeros(0),
                  n_layers=n_layers, seed=seed, in_dims=np.array(in_dims), out_dims=np.array(out_dims),
                  shape_keys=np.array(sorted(shapes)),
                  shape_vals=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]))
    if store_weights:
        arrays.update(sd_arrays(sd))
    save(name, **arrays)


def gen_graph_transformer(R):
    """F3: full GraphTransformer forward."""
    # tiny, weights stored, non-symmetric E input, y output requested
    _gt_case(R, "f3_gt_tiny", fx.TINY_DIMS, (5, 9, 6), (3, 4, 2), 2, [9, 5, 2], 31, True, symmetric_E=False)
    # tiny, CoW-like class counts, no y output (last layer has no y modules)
    _gt_case(R, "f3_gt_tiny_noy", fx.TINY_DIMS, (4, 11, 7), (1, 6, 0), 3, [6, 10, 3, 1], 32, True, onehot_C=6)
    # ConStruct dims, CoW shapes with extra-feature widths (X=10,E=25,y=43) -> (1,14,0); weights by recipe
    _gt_case(R, "f3_gt_construct_cow", fx.CONSTRUCT_DIMS, (10, 25, 43), (1, 14, 0), 5, [12, 7], 33, False, onehot_C=14)
    # vessel dims, ATM shapes without extra features (X=1,E=5,y=1) -> (1,5,0)
    _gt_case(R, "f3_gt_vessel_atm", fx.VESSEL_DIMS, (1, 5, 1), (1, 5, 0), 5, [12, 9], 34, False, onehot_C=5)
    # n > 32 exercises the multi-chunk key loop of the fused kernel
    _gt_case(R, "f3_gt_vessel_n40", fx.VESSEL_DIMS, (1, 5, 1), (1, 5, 0), 2, [40, 33], 35, False, onehot_C=5)


def gen_pooling(R):
    """F4: Xtoy / Etoy pooled statistics and masked_softmax (incl. the all-masked case)."""
    torch.manual_seed(41)
    mask = fx.ragged_mask([6, 3, 1], 6)
    B, n = mask.shape
    X = torch.randn(B, n, 8) * mask[..., None]
    E = torch.randn(B, n, n, 4)
    E = R.U.PlaceHolder(X=X, E=0.5 * (E + E.transpose(1, 2)), y=None, node_mask=mask).mask().E
    xt, et = R.LY.Xtoy(8, 5), R.LY.Etoy(4, 5)
    with torch.no_grad():
        xt.lin.weight.copy_(torch.eye(5, 32)); xt.lin.bias.zero_()
        et.lin.weight.copy_(torch.eye(5, 16)); et.lin.bias.zero_()
    # capture the pooled z by hooking lin's input
    zs = {}
    xt.lin.register_forward_hook(lambda m, i, o: zs.__setitem__("x", i[0].detach().clone()))
    et.lin.register_forward_hook(lambda m, i, o: zs.__setitem__("e", i[0].detach().clone()))
    xm = mask[..., None]
    with torch.no_grad():
        xt(X, xm)
        et(E, xm.unsqueeze(2), xm.unsqueeze(1))
    Y = torch.randn(B, n, n, 2, 3)
    sm = R.LY.masked_softmax(Y, xm.unsqueeze(1).expand(-1, n, -1, 2), dim=2)
    sm0 = R.LY.masked_softmax(Y, torch.zeros(B, n, n, 2, dtype=torch.bool), dim=2)
    save("f4_pooling", X=X, E=E, node_mask=mask, zx=zs["x"], ze=zs["e"], Y=Y, softmax=sm, softmax_allmasked=sm0)


def _hand_graphs():
    """Small labelled graphs: path/tree, C3..C6 cycles, disconnected, empty, dense."""
    def edges_to(n, edges):
        a = np.zeros((n, n), dtype=np.int64)
        for k, (i, j) in enumerate(edges):
            a[i, j] = a[j, i] = 1 + (k % 3)
        return a
    n = 8
    gs = [
        (6, [(0, 1), (1, 2), (2, 3), (1, 4), (4, 5)]),                      # tree
        (3, [(0, 1), (1, 2), (2, 0)]),                                      # C3
        (4, [(0, 1), (1, 2), (2, 3), (3, 0)]),                              # C4
        (5, [(i, (i + 1) % 5) for i in range(5)]),                          # C5
        (6, [(i, (i + 1) % 6) for i in range(6)]),                          # C6
        (8, [(0, 1), (1, 2), (2, 0), (3, 4), (5, 6), (6, 7), (7, 5), (5, 3)]),   # two components
        (5, []),                                                            # empty
        (6, [(i, j) for i in range(6) for j in range(i + 1, 6)]),           # K6
        (7, [(0, 1), (0, 2), (0, 3), (3, 4), (4, 5), (5, 3), (5, 6), (1, 2)]),
    ]
    labs = np.zeros((len(gs), n, n), dtype=np.int64)
    nn_ = []
    for b, (k, ed) in enumerate(gs):
        labs[b, :k, :k] = edges_to(k, ed)
        nn_.append(k)
    return torch.from_numpy(labs), torch.tensor(nn_), n


def gen_extra_features(R):
    """F5/F6: AdjacencyFeatures + EigenFeatures + full ExtraFeatures on hand-built graphs."""
    labs, nn_, n = _hand_graphs()
    mask = fx.ragged_mask(nn_, n)
    C = 4
    E = fx.one_hot_edges(labs, mask, C)
    X = mask[..., None].float()
    cfg = SimpleNamespace(model=SimpleNamespace(eigenfeatures=True, num_degree=10, max_degree=10, cycle_features=True,
                                                num_eigenvectors=5, num_eigenvalues=9))
    ef = R.EF.ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=n))
    data = SimpleNamespace(X=X, E=E, y=torch.zeros(len(nn_), 0), node_mask=mask)
    torch.manual_seed(51)
    out = ef(data)
    ax, ay, ae = ef.adj_features(data)
    torch.manual_seed(51)
    evl, evc = ef.eigenfeatures.compute_features(data)
    dims = ef.update_input_dims(fx.dims_ns(1, 14, 1))
    save("f5_extra_features", labels=labs, n_nodes=nn_, node_mask=mask, C=C, X=X, E=E,
         adj_x=ax, adj_y=ay, adj_e=ae, eval_feat=evl, evec_feat=evc,
         out_X=out.X, out_E=out.E, out_y=out.y, kpowers=ef.adj_features.k,
         upd_dims=np.array([dims.X, dims.E, dims.y]))
    # a random sparse batch at CoW-like size
    mask = fx.ragged_mask([20, 13, 17, 9], 20)
    labs = fx.random_labels(mask, 14, 0.12, 52)
    E = fx.one_hot_edges(labs, mask, 14)
    X = mask[..., None].float()
    data = SimpleNamespace(X=X, E=E, y=torch.zeros(4, 0), node_mask=mask)
    ef2 = R.EF.ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=32))
    ax, ay, ae = ef2.adj_features(data)
    torch.manual_seed(52)
    evl, evc = ef2.eigenfeatures.compute_features(data)
    save("f5_extra_features_random", labels=labs, node_mask=mask, C=14, X=X, E=E, adj_x=ax, adj_y=ay, adj_e=ae,
         eval_feat=evl, evec_feat=evc)


def _noise_model(R, transition, dataset, C_e, top_k, T=500, e_marg=None):
    cfg = SimpleNamespace(
        model=SimpleNamespace(nu=dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1), transition=transition,
                              diffusion_steps=T, top_k=top_k),
        dataset=SimpleNamespace(name=dataset))
    xm = torch.ones(1)
    em = torch.ones(C_e) / C_e if e_marg is None else e_marg
    cm = torch.zeros(0)
    real_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        if transition == "absorbing_edges":
            nm = R.NM.AbsorbingEdgesTransition(cfg, xm, em, cm, 0)
        elif transition == "marginal":
            nm = R.NM.MarginalTransition(cfg, xm, em, cm, 0)
        elif transition == "uniform":
            nm = R.NM.DiscreteUniformTransition(cfg, fx.dims_ns(1, C_e, 0, 0))
        elif transition == "absorbing":
            nm = R.NM.AbsorbingTransition(cfg, fx.dims_ns(1, C_e, 0, 0))
    finally:
        torch.load = real_load
    return nm


def gen_schedules(R):
    """F7: beta / alpha_bar tables and Qt / Qt_bar samples."""
    arrays = {}
    ts = torch.tensor([[500], [499], [250], [2], [1], [0]])
    for tr in ("absorbing_edges", "marginal", "uniform", "absorbing"):
        em = torch.tensor([0.7, 0.1, 0.1, 0.05, 0.05]) if tr in ("marginal", "absorbing_edges") else None
        nm = _noise_model(R, tr, "cow", 5, 4, e_marg=em)
        arrays[f"{tr}/betas"] = nm._betas
        arrays[f"{tr}/alphas_bar"] = nm._alphas_bar
        if tr == "absorbing_edges":
            arrays[f"{tr}/betas_abs"] = nm._betas_abs
            arrays[f"{tr}/alphas_abs_bar"] = nm._alphas_abs_bar
        qt, qtb = nm.get_Qt(ts), nm.get_Qt_bar(ts)
        arrays[f"{tr}/Qt_E"], arrays[f"{tr}/Qtb_E"] = qt.E, qtb.E
        arrays[f"{tr}/Qt_X"], arrays[f"{tr}/Qtb_X"] = qt.X, qtb.X
    save("f7_schedules", t=ts, e_marginals=np.array([0.7, 0.1, 0.1, 0.05, 0.05], dtype=np.float32), **arrays)


class _UniformFeed:
    """Replaces Tensor.multinomial inside the oracle process (see module docstring)."""

    def __init__(self):
        self.queue = []

    def install(self):
        feed = self
        self._orig = torch.Tensor.multinomial

        def patched(t, num_samples, replacement=False, *, generator=None):
            u = feed.queue.pop(0)
            assert u.shape == (t.shape[0], num_samples), (u.shape, t.shape, num_samples)
            return torch.from_numpy(inverse_cdf_draws(t.detach().numpy(), u))
        torch.Tensor.multinomial = patched

    def remove(self):
        torch.Tensor.multinomial = self._orig


def gen_reverse_step(R):
    """F8/F9: prob_X/prob_E and sampled E_s / extra_info of sample_zs_from_zt_and_pred under injected uniforms."""
    feed = _UniformFeed()
    for name, tr, C, top_k in (("cow_abs", "absorbing_edges", 14, 4), ("atm_abs", "absorbing_edges", 5, 4),
                               ("marg", "marginal", 5, 4)):
        em = torch.tensor([0.7, 0.1, 0.1, 0.05, 0.05]) if tr == "marginal" else None
        nm = _noise_model(R, tr, "cow", C, top_k, e_marg=em)
        mask = fx.ragged_mask([9, 5, 1, 12], 12)
        B, n = mask.shape
        arrays = dict(node_mask=mask, C=C, top_k=-1 if top_k is None else top_k, transition=tr)
        if em is not None:
            arrays["e_marginals"] = em
        for t in (500, 499, 250, 2, 1):
            g = torch.Generator().manual_seed(1000 + t)
            lab = fx.random_labels(mask, C, 0.0 if t == 500 else 0.25 * (1 - t / 500) + 0.05, 2000 + t)
            ztE = fx.one_hot_edges(lab, mask, C)
            ztX = mask[..., None].float()
            logits = torch.randn(B, n, n, C, generator=g) * 2.0
            logits = 0.5 * (logits + logits.transpose(1, 2))
            predX = torch.randn(B, n, 1, generator=g)
            ux = torch.rand(B * n, 1, generator=g).numpy()
            u0 = torch.rand(B * n * n, 1, generator=g).numpy()
            k = top_k if top_k is not None else 0
            uk = torch.rand(B * n * n, max(k, 1), generator=g).numpy()
            z_t = R.U.PlaceHolder(X=ztX, charges=torch.zeros(B, n, 0), E=ztE, y=torch.zeros(B, 0),
                                  t_int=torch.full((B, 1), t), t=torch.full((B, 1), t / 500.0), node_mask=mask,
                                  pos=torch.zeros(B, n, 3))
            pred = R.U.PlaceHolder(X=predX, charges=torch.zeros(B, n, 0), E=logits, y=None, node_mask=mask)
            s_int = torch.full((B, 1), t - 1)
            # capture prob_X / prob_E as handed to sample_discrete_features (before its in-place edits)
            captured = {}
            orig_sdf = R.DU.sample_discrete_features

            def spy(probX, probE, prob_charges, node_mask, top_k=None):
                captured["pX"], captured["pE"] = probX.clone(), probE.clone()
                return orig_sdf(probX, probE, prob_charges, node_mask=node_mask, top_k=top_k)
            R.DU.sample_discrete_features = spy
            feed.queue = [ux, u0] + ([uk] if top_k is not None else [])
            feed.install()
            try:
                z_s = nm.sample_zs_from_zt_and_pred(z_t, pred, s_int)
            finally:
                feed.remove()
                R.DU.sample_discrete_features = orig_sdf
            assert not feed.queue
            p = f"t{t}/"
            arrays.update({p + "labels_t": lab, p + "logits_E": logits, p + "pred_X": predX,
                           p + "ux": ux.reshape(B, n), p + "u0": u0.reshape(B, n, n),
                           p + "uk": uk.reshape(B, n, n, -1), p + "prob_X": captured["pX"], p + "prob_E": captured["pE"],
                           p + "X_s": z_s.X, p + "E_s": z_s.E,
                           p + "extra_info": z_s.extra_info if z_s.extra_info is not None else np.zeros(0)})
        save(f"f8_reverse_step_{name}", **arrays)


def gen_sample_discrete(R):
    """F9b: sample_discrete_features called directly, without top_k (the apply_noise / training form;
    the reference's reverse step itself requires top_k, noise_model.py:433)."""
    feed = _UniformFeed()
    g = torch.Generator().manual_seed(77)
    mask = fx.ragged_mask([6, 11, 2], 11)
    B, n = mask.shape
    C = 5
    pE = torch.rand(B, n, n, C, generator=g)
    pE[..., 3] = 0                       # a zero-probability class
    pE = pE / pE.sum(-1, keepdim=True)
    pX = torch.rand(B, n, 3, generator=g)
    pX = pX / pX.sum(-1, keepdim=True)
    ux = torch.rand(B * n, 1, generator=g).numpy()
    u0 = torch.rand(B * n * n, 1, generator=g).numpy()
    feed.queue = [ux, u0]
    feed.install()
    try:
        out = R.DU.sample_discrete_features(pX.clone(), pE.clone(), torch.zeros(B, n, 0), mask)
    finally:
        feed.remove()
    save("f9_sample_discrete_notopk", node_mask=mask, prob_X=pX, prob_E=pE, ux=ux.reshape(B, n), u0=u0.reshape(B, n, n),
         X=out.X, E=out.E)


def gen_placeholder(R):
    """F11: PlaceHolder.mask / collapse."""
    torch.manual_seed(61)
    mask = fx.ragged_mask([5, 3], 5)
    X, E = torch.randn(2, 5, 3), torch.randn(2, 5, 5, 4)
    E = 0.5 * (E + E.transpose(1, 2))
    pos = torch.randn(2, 5, 3)
    ph = R.U.PlaceHolder(X=X.clone(), E=E.clone(), y=torch.zeros(2, 0), charges=torch.zeros(2, 5, 0), node_mask=mask,
                         pos=pos.clone()).mask()
    col = ph.collapse(None)
    save("f11_placeholder", X=X, E=E, pos=pos, node_mask=mask, mX=ph.X, mE=ph.E, mpos=ph.pos, cX=col.X, cE=col.E)


def gen_projector(R):
    """F12: the host-side validity projectors (projector_3d_utils.py) driven over a synthetic absorbing-style
    trajectory: at step s every still-empty pair takes the step's proposal, the REAL projector edits z_s in place,
    and the edited state is the next z_t.  Stored: proposals, candidates, the projected labels per step, the
    blocked-edge table and (CoW) the correction counter.  np.random is seeded once before the constructor."""
    import contextlib
    import io
    from src.ConStruct.projector import projector_3d_utils as PJ
    cases = (("cow", "CoWProjector", 14, [10, 7, 12, 3, 1, 9]), ("cow_vanilla", "CoWVanillaProjector", 14, [10, 7, 12, 3, 1, 9]),
             ("atm", "ATMTreeProjector", 5, [10, 7, 12, 3, 1, 9]), ("atm_vanilla", "ATMVanillaTreeProjector", 5, [10, 7, 12, 3, 1, 9]),
             ("tree", "TreeProjector", 5, [10, 7, 12, 3, 1, 9]))
    S, k = 12, 4
    for ci, (name, cls, C, n_nodes) in enumerate(cases):
        rs = np.random.RandomState(900 + ci)
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        valid = (mask[:, :, None] & mask[:, None, :]).numpy() & ~np.eye(n, dtype=bool)[None]
        props = np.zeros((S, B, n, n), dtype=np.int8)
        cands = np.zeros((S, B, n, n, k), dtype=np.int8)
        for s in range(S):
            lab = np.where(rs.rand(B, n, n) < 0.5, rs.randint(1, min(C, 4), (B, n, n)), rs.randint(1, C, (B, n, n)))
            on = rs.rand(B, n, n) < (0.10 if s % 4 else 0.03)
            lab = np.triu(lab * on, 1)
            props[s] = (lab + lab.transpose(0, 2, 1)) * valid
            cd = np.argsort(rs.rand(B, n, n, C), axis=-1)[..., :k]          # k distinct classes, 0 included
            iu = np.triu(np.ones((n, n), dtype=bool), 1)[None, :, :, None]
            cands[s] = np.where(iu, cd, cd.transpose(0, 2, 1, 3))
        seed = 4000 + ci
        np.random.seed(seed)
        z = torch.zeros(B, n, n, dtype=torch.long)
        def holder(lab, cd):
            return R.U.PlaceHolder(X=mask[..., None].float(), E=fx.one_hot_edges(lab, mask, C), y=torch.zeros(B, 0),
                                   node_mask=mask, pos=torch.zeros(B, n, 3), extra_info=cd)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            z_T = holder(z, None)
            proj = getattr(PJ, cls)(z_T) if cls == "TreeProjector" else getattr(PJ, cls)(z_T, torch.device("cpu"))
            outs = np.zeros((S, B, n, n), dtype=np.int8)
            for s in range(S):
                raw = torch.where(z == 0, torch.from_numpy(props[s]).long(), z)
                z_s = holder(raw, torch.from_numpy(cands[s]).long())
                proj.project(z_s)
                z = torch.argmax(z_s.E, dim=3) * (z_s.E.sum(-1) > 0)
                outs[s] = z.numpy()
        blocked = np.zeros((B, n, n), dtype=bool)
        for b, d in proj.blocked_edges.items():
            for (i, j), v in d.items():
                blocked[b, i, j] = v
        counter = getattr(getattr(proj, "top_cow_validity_check_and_fix_obj", None), "counter", 0)
        n_del = int(sum(((np.where(outs[s] == 0, 1, 0) * (props[s] != 0))).sum() for s in range(S)))
        print(f"{name}: kept {int((outs[-1] != 0).sum()) // 2} edges, blocked {int(blocked.sum())}, counter {counter}, "
              f"relabelled {int(((outs[-1] != 0) & (sum(props.astype(np.int64)) != 0) & (outs[-1] != np.where(outs[-1] != 0, outs[-1], 0))).sum())}")
        save(f"f12_projector_{name}", node_mask=mask, C=C, top_k=k, seed=seed, props=props, cands=cands, out_labels=outs,
             blocked=blocked, counter=counter)


def gen_sample_plan(R):
    """F13: the batching rule of sample_n_graphs (diffusion_model_discrete_3d.py:753-843), recorded by calling the REAL
    method on a stand-in ``self`` whose sample_batch only notes the node counts it is given.  The module's other
    imports (Lightning, visualisation, molecular metrics, the 3D models, torch_geometric loaders) are stubbed: none of
    them is touched by this method."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class _D:
        def __init__(self, *a, **k):
            pass
    before = dict(sys.modules)
    tg_path = sys.modules["torch_geometric"].__dict__.get("__path__", None)
    stub("pytorch_lightning", LightningModule=_D)
    stub("src.ConStruct.analysis.visualization", Visualizer=_D)
    stub("src.ConStruct.diffusion.extra_features_molecular", ExtraMolecularFeatures=_D)
    stub("src.ConStruct.metrics.train_metrics", TrainLoss=_D, TrainMolecularMetrics=_D)
    stub("src.ConStruct.metrics.abstract_metrics", NLL=_D, TrainAbstractMetrics=_D, XKl=_D, ChargesKl=_D, EKl=_D)
    stub("src.ConStruct.models.equivariant_pos_transformer_model", EquivariantGraph3DTransformer=_D)
    stub("src.ConStruct.models.pos_transformer_model", Graph3DTransformer=_D)
    sys.modules["torch_geometric"].__path__ = []
    stub("torch_geometric.data", Batch=_D, Dataset=_D, LightningDataset=_D)
    stub("torch_geometric.data.data", BaseData=_D)
    stub("torch_geometric.loader", DataLoader=_D)
    import src.ConStruct.diffusion_model_discrete_3d as DM
    arrays = {}
    cases = [  # name, to_generate, chains_to_save, samples_to_save, batch_size, adaptive, reference_bs
        ("fixed_small", 41, 2, 5, 4, False, 0), ("fixed_only_first", 9, 3, 20, 3, False, 0),
        ("fixed_none_first", 50, 0, 0, 5, False, 0), ("adaptive", 60, 1, 12, 0, True, 6), ("adaptive_big", 90, 0, 7, 0, True, 40),
        ("single", 1, 0, 0, 8, False, 0)]
    for ci, (name, n_gen, n_chain, n_save, bs, adaptive, ref_bs) in enumerate(cases):
        rs = np.random.RandomState(700 + ci)
        draws = []

        class Dist:
            def sample_n(self, k, device):
                v = torch.from_numpy(rs.randint(3, 41, size=k))
                draws.append(v.clone())
                return v
        calls = []
        me = SimpleNamespace(nodes_dist=Dist(), device="cpu", number_chain_steps=10,
                             cfg=SimpleNamespace(dataset=SimpleNamespace(adaptive_loader=adaptive),
                                                 train=SimpleNamespace(batch_size=bs, reference_batch_size=ref_bs)),
                             sample_batch=lambda n_nodes, **k: calls.append([int(v) for v in n_nodes]) or len(calls))
        DM.Discrete3dDenoisingDiffusion.sample_n_graphs(me, n_gen, n_chain, n_save, False)
        arrays[f"{name}/args"] = np.array([n_gen, n_chain, n_save, bs, int(adaptive), ref_bs])
        arrays[f"{name}/first"] = draws[0].numpy() if min(n_gen, max(n_save, n_chain)) > 0 else np.zeros(0, np.int64)
        rest = draws[-1].numpy() if len(draws) > (1 if min(n_gen, max(n_save, n_chain)) > 0 else 0) else np.zeros(0, np.int64)
        arrays[f"{name}/rest"] = rest
        arrays[f"{name}/batch_sizes"] = np.array([len(c) for c in calls])
        arrays[f"{name}/batches_flat"] = np.array([v for c in calls for v in c])
        print(f"{name}: {len(calls)} batches, sizes {[len(c) for c in calls]}")
    save("f13_sample_plan", names=np.array([c[0] for c in cases]), **arrays)
    # take the stand-ins out again: later generators import the REAL pos_transformer_model / train_metrics modules
    for k in [k for k in sys.modules if k not in before]:
        del sys.modules[k]
    sys.modules.update(before)
    if tg_path is None:
        sys.modules["torch_geometric"].__dict__.pop("__path__", None)
    for pkg, attr in (("src.ConStruct.models", "pos_transformer_model"), ("src.ConStruct.models", "equivariant_pos_transformer_model"),
                      ("src.ConStruct.metrics", "train_metrics"), ("src.ConStruct.metrics", "abstract_metrics"),
                      ("src.ConStruct.analysis", "visualization"), ("src.ConStruct.diffusion", "extra_features_molecular"),
                      ("src.ConStruct", "diffusion_model_discrete_3d")):
        m = sys.modules.get(pkg)
        if m is not None and attr in m.__dict__ and (pkg + "." + attr) not in sys.modules:
            delattr(m, attr)


def _fresh_metrics_modules():
    """(Re-)import the reference's train_metrics / abstract_metrics against the torchmetrics stand-in installed just before:
    drop the cached modules AND the attributes on their package (``from pkg import mod`` would hand back the old object)."""
    import importlib
    pkg = sys.modules.get("src.ConStruct.metrics")
    for mod in ("train_metrics", "abstract_metrics"):
        sys.modules.pop("src.ConStruct.metrics." + mod, None)
        if pkg is not None and mod in pkg.__dict__:
            delattr(pkg, mod)
    return importlib.import_module("src.ConStruct.metrics.train_metrics")


def _spanning_tree_labels(n_nodes, n, C, seed):
    """Airway-like clean graphs: a random spanning tree over the valid nodes, labels in 1..C-1."""
    rs = np.random.RandomState(seed)
    lab = np.zeros((len(n_nodes), n, n), dtype=np.int64)
    for b, k in enumerate(n_nodes):
        for v in range(1, int(k)):
            u = rs.randint(0, v)
            lab[b, u, v] = lab[b, v, u] = rs.randint(1, C)
    return torch.from_numpy(lab)


def gen_training(R):
    """F10: one training step of the reference (diffusion_model_discrete_3d.py:215-239): apply_noise (noise_model.py:187-233)
    under injected uniforms, GraphTransformer forward in train mode with dropout 0, TrainLoss (train_metrics.py:25-85,
    lambda_train = [1, 2, 5, 0]), loss.backward().  Stored: the clean graphs, t, the uniforms, z_t, logits, the loss and
    every parameter gradient (tiny dims).  torchmetrics is absent here, so ``Metric`` is a minimal stand-in with the
    published forward semantics (the value of the CURRENT batch: reset, update, compute, then merge into the running
    state); the arithmetic inside update / compute is the reference's own (abstract_metrics.py:77-93)."""
    import torch.nn as nn

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default
            setattr(self, name, default.clone())

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

        def forward(self, *a, **k):
            keep = {n: getattr(self, n) for n in self._defaults}
            self.reset()
            self.update(*a, **k)
            val = self.compute()
            for n in self._defaults:
                setattr(self, n, keep[n] + getattr(self, n).detach())
            return val
    sys.modules["torchmetrics"].Metric = Metric
    sys.modules["torchmetrics"].MeanSquaredError = Metric
    sys.modules["torchmetrics"].MetricCollection = Metric
    TL = _fresh_metrics_modules()
    feed = _UniformFeed()
    for name, tr, C, n_nodes, dims, n_layers in (("atm_tiny", "absorbing_edges", 5, [11, 6, 9, 2], fx.TINY_DIMS, 2),
                                                 ("cow_marg_tiny", "marginal", 6, [8, 12, 5], fx.TINY_DIMS, 2)):
        em = torch.tensor([0.7, 0.1, 0.1, 0.05, 0.03, 0.02]) if tr == "marginal" else None
        nm = _noise_model(R, tr, "cow", C, 4, e_marg=em)
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        lab = _spanning_tree_labels(n_nodes, n, C, 300 + C)
        E0 = fx.one_hot_edges(lab, mask, C)
        # the reference's dense_data (utils.to_dense): zero rows on padding AND on the diagonal (mask()), one-hot elsewhere
        X0 = mask[..., None].float()
        g = torch.Generator().manual_seed(910 + C)
        pos = torch.randn(B, n, 3, generator=g)
        dense = R.U.PlaceHolder(X=X0, charges=torch.zeros(B, n, 0), E=E0, y=torch.zeros(B, 0), node_mask=mask, pos=pos)
        t_int = torch.randint(1, 501, (B, 1), generator=g)
        ux = torch.rand(B * n, 1, generator=g).numpy()
        u0 = torch.rand(B * n * n, 1, generator=g).numpy()
        feed.queue = [ux, u0]
        feed.install()
        try:
            z_t = nm.apply_noise(dense, t_int=t_int)
        finally:
            feed.remove()
        assert not feed.queue
        model = R.TM.GraphTransformer(n_layers=n_layers, input_dims=fx.dims_ns(1, C, 1), hidden_mlp_dims=dims["hidden_mlp_dims"],
                                      hidden_dims=dims["hidden_dims"], output_dims=fx.dims_ns(1, C, 0), dropout=0.0).train()
        # quirk: GraphTransformer does not hand its ``dropout`` to the layers (transformer_model.py:309-317), which keep the
        # default 0.1 (:35) whatever the config says; "dropout 0" therefore needs the modules switched off one by one
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = fx.make_state_dict(shapes, seed=70 + C)
        model.load_state_dict(sd)
        inp = SimpleNamespace(X=z_t.X.float(), E=z_t.E.float(), y=z_t.t.float(), node_mask=mask)
        pred = model(inp)
        loss_fn = TL.TrainLoss(lambda_train=[1, 2, 5, 0])
        loss, _ = loss_fn(masked_pred=pred, masked_true=dense, log=False)
        loss.backward()
        grads = {}
        for k, p in model.named_parameters():          # big tensors (lin_y1/2: dy x 2048): head + norm + sum only
            assert p.grad is not None, k
            if p.grad.numel() <= 4096:
                grads["grad/" + k] = p.grad
            else:
                grads["gradhead/" + k] = p.grad.flatten()[:1024]
                grads["gradstat/" + k] = torch.stack([p.grad.double().norm(), p.grad.double().sum()])
        save(f"f10_training_{name}", node_mask=mask, C=C, transition=tr, n_layers=n_layers, labels0=lab, t_int=t_int, pos=pos,
             ux=ux.reshape(B, n), u0=u0.reshape(B, n, n), zt_E=z_t.E, zt_X=z_t.X, zt_pos=z_t.pos, zt_t=z_t.t,
             logits_E=pred.E.detach(), logits_X=pred.X.detach(), loss=loss.detach(), sd_seed=70 + C,
             e_marginals=em if em is not None else np.zeros(0), **grads)
        print(f"{name}: loss {float(loss.detach()):.6f}")


def gen_extra_features_large(R):
    """F5b: AdjacencyFeatures + EigenFeatures beyond n = 32 (SURVEY configs 3 / 5): n_max = 40 (CoW classes) and
    n_max = 256 (ATM classes, tree-like graphs with a few extra edges so that cycles exist).  Only what does not depend
    on the eigenvector basis is stored in full; E features as the labels they derive from are large, so the reachability /
    Adamic-Adar tensor is stored for a strided subset of rows."""
    cfg = SimpleNamespace(model=SimpleNamespace(eigenfeatures=True, num_degree=10, max_degree=10, cycle_features=True,
                                                num_eigenvectors=5, num_eigenvalues=9))
    for name, n_nodes, C, dens, seed in (("n40", [40, 33, 37, 21], 14, 0.06, 53), ("n256", [256, 201], 5, None, 54)):
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        if dens is not None:
            labs = fx.random_labels(mask, C, dens, seed)
        else:
            labs = _spanning_tree_labels(n_nodes, n, C, seed)
            rs = np.random.RandomState(seed)
            for b, k in enumerate(n_nodes):            # a few chords -> cycles of several lengths
                for _ in range(12):
                    i, j = rs.randint(0, k, 2)
                    if i != j:
                        labs[b, i, j] = labs[b, j, i] = rs.randint(1, C)
        E = fx.one_hot_edges(labs, mask, C)
        X = mask[..., None].float()
        data = SimpleNamespace(X=X, E=E, y=torch.zeros(B, 0), node_mask=mask)
        ef = R.EF.ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=n))
        ax, ay, ae = ef.adj_features(data)
        torch.manual_seed(seed)
        evl, evc = ef.eigenfeatures.compute_features(data)
        rows = np.arange(0, n, 1 if n <= 40 else 9)
        save(f"f5_extra_features_{name}", labels=labs.to(torch.int8), node_mask=mask, C=C, adj_x=ax, adj_y=ay, rows=rows,
             adj_e_rows=ae[:, rows].to(torch.float16) if n > 40 else ae[:, rows], eval_feat=evl, evec_feat=evc,
             kdiag=torch.stack([torch.diagonal(ef.adj_features.k[i], dim1=-2, dim2=-1) for i in range(10)]))


def gen_graph_transformer_large(R):
    """F3b: GraphTransformer forward at ATM size (n_max = 256, vessel dims, 5 layers; SURVEY config 3's forward).  The output
    E is [B,256,256,5]: a strided subset of rows is stored, plus the per-graph sums over all entries."""
    torch.manual_seed(36)
    dims, in_dims, out_dims, n_layers, n_nodes, seed = fx.VESSEL_DIMS, (1, 5, 1), (1, 5, 0), 5, [256, 203], 36
    mask = fx.ragged_mask(n_nodes)
    B, n = mask.shape
    model = R.TM.GraphTransformer(n_layers=n_layers, input_dims=fx.dims_ns(*in_dims), hidden_mlp_dims=dims["hidden_mlp_dims"],
                                  hidden_dims=dims["hidden_dims"], output_dims=fx.dims_ns(*out_dims), dropout=0.1).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(fx.make_state_dict(shapes, seed=seed))
    lab = _spanning_tree_labels(n_nodes, n, 5, seed)
    E = fx.one_hot_edges(lab, mask, 5)
    X = mask[..., None].float()
    y = torch.tensor([[0.37], [0.81]])
    with torch.no_grad():
        out = model(SimpleNamespace(X=X, E=E, y=y, node_mask=mask))
    rows = np.arange(0, n, 17)
    save("f3_gt_vessel_n256", labels=lab.to(torch.int8), node_mask=mask, y=y, n_layers=n_layers, seed=seed, rows=rows,
         outE_rows=out.E[:, rows], outX=out.X, outE_sum=out.E.double().sum(dim=(1, 2)),
         outE_abs_sum=out.E.double().abs().sum(dim=(1, 2)),
         shape_keys=np.array(sorted(shapes)), shape_vals=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]))


def gen_graph3d(R):
    """G1/G3: the position-aware denoiser the shipped driver instantiates: one 3D NodeEdgeBlock with and
    without the y branch, Graph3DTransformer and EquivariantGraph3DTransformer forwards (eval).  Padded nodes carry
    non-zero positions on purpose (the model has to mask them), and the batches are ragged (the re-centring quirk of
    PlaceHolder.mask only shows with padding)."""
    from src.ConStruct.models import pos_transformer_model as P3
    from src.ConStruct.models import equivariant_pos_transformer_model as EQ
    torch.manual_seed(41)
    dx, de, dy, h = 32, 32, 16, 4
    mask = fx.ragged_mask([7, 4, 1], 7)
    B, n = mask.shape
    X = torch.randn(B, n, dx)
    E = torch.randn(B, n, n, de)
    E = 0.5 * (E + E.transpose(1, 2))
    y = torch.randn(B, dy)
    pos = torch.randn(B, n, 3)
    ph = R.U.PlaceHolder(X=X, E=E, y=y, node_mask=mask).mask()
    X, E = ph.X, ph.E
    for predicts_y in (True, False):
        blk = P3.NodeEdgeBlock(dx, de, dy, h, predicts_y=predicts_y).eval()
        sd = fx.make_state_dict({k: v.shape for k, v in blk.state_dict().items()}, seed=42)
        blk.load_state_dict(sd)
        with torch.no_grad():
            nX, nE, ny = blk(X, E, y, pos, mask)
        save(f"g1_block3d_y{int(predicts_y)}", X=X, E=E, y=y, pos=pos, node_mask=mask, n_head=h, newX=nX, newE=nE,
             new_y=ny if ny is not None else np.zeros(0), **sd_arrays(sd))

    def case(name, cls, dims, in_dims, out_dims, n_layers, n_nodes, seed, store_weights, onehot_C=None, radius=True,
             rows=None):
        torch.manual_seed(seed)
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        model = cls(n_layers=n_layers, input_dims=fx.dims_ns(*in_dims), hidden_mlp_dims=dims["hidden_mlp_dims"],
                    hidden_dims=dims["hidden_dims"], output_dims=fx.dims_ns(*out_dims), dropout=0.1).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = fx.make_state_dict(shapes, seed=seed)
        model.load_state_dict(sd)
        X = torch.randn(B, n, in_dims[0] + in_dims[3])
        E = torch.randn(B, n, n, in_dims[1])
        E = 0.5 * (E + E.transpose(1, 2))
        if onehot_C is not None:
            E[..., :onehot_C] = fx.one_hot_edges(fx.random_labels(mask, onehot_C, 0.15, seed), mask, onehot_C)
        y = torch.rand(B, in_dims[2])
        pos = torch.randn(B, n, 3) * 1.7 + 0.4
        rad = torch.zeros(B, n, n, 1) if radius else None
        with torch.no_grad():
            out = model(R.U.PlaceHolder(X=X, E=E, y=y, pos=pos, radius=rad, node_mask=mask))
        arrays = dict(X=X, E=E, y=y, pos=pos, node_mask=mask, outX=out.X, out_charges=out.charges, out_pos=out.pos,
                      n_layers=n_layers, seed=seed, in_dims=np.array(in_dims), out_dims=np.array(out_dims),
                      shape_keys=np.array(sorted(shapes)),
                      shape_vals=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]))
        if rows is None:
            arrays.update(outE=out.E, out_radius=out.radius if radius else np.zeros(0))
        else:
            arrays.update(rows=rows, outE_rows=out.E[:, rows], out_radius_rows=out.radius[:, rows],
                          outE_sum=out.E.double().sum(dim=(1, 2)), outE_abs_sum=out.E.double().abs().sum(dim=(1, 2)))
            del arrays["E"]
            arrays["labels"] = E[..., :onehot_C].argmax(-1).to(torch.int8)
        assert out.y is None
        if store_weights:
            arrays.update(sd_arrays(sd))
        save(name, **arrays)

    # (X, E, y, charges) widths
    case("g3_g3d_tiny", P3.Graph3DTransformer, fx.TINY_DIMS, (3, 9, 4, 2), (2, 4, 0, 1), 2, [9, 5, 2], 43, True)
    case("g3_g3d_tiny_equivariant", EQ.EquivariantGraph3DTransformer, fx.TINY_DIMS, (3, 9, 4, 2), (2, 4, 0, 1), 3,
         [6, 10, 3, 1], 44, True, radius=False)
    # the shipped vessel configuration (discrete_vessels.yaml): no extra features, y = t; CoW (14 edge classes) and ATM (5)
    case("g3_g3d_vessel_cow", P3.Graph3DTransformer, fx.VESSEL_DIMS, (1, 14, 1, 0), (1, 14, 0, 0), 5, [12, 7, 20], 45, False,
         onehot_C=14)
    case("g3_g3d_vessel_n40", P3.Graph3DTransformer, fx.VESSEL_DIMS, (1, 5, 1, 0), (1, 5, 0, 0), 2, [40, 33], 46, False,
         onehot_C=5)
    case("g3_g3d_vessel_equivariant", EQ.EquivariantGraph3DTransformer, fx.VESSEL_DIMS, (1, 5, 1, 0), (1, 5, 0, 0), 5, [17, 32], 47,
         False, onehot_C=5)


def gen_training_3d(R):
    """G10: the loss and every parameter gradient of one training step through the class the driver builds
    (Graph3DTransformer, train mode with every Dropout switched off, TrainLoss with the radius term,
    train_metrics.py:25-85, lambda_train = [1, 2, 5, 0]) on a given noisy state (apply_noise itself is pinned by F10).
    Tiny dims, graphs of at most 25 nodes (torch.cdist takes differences there)."""
    import torch.nn as nn

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            self._defaults[name] = default
            setattr(self, name, default.clone())

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone())

        def forward(self, *a, **k):
            self.reset()
            self.update(*a, **k)
            return self.compute()

    class MSE(Metric):                         # torchmetrics.MeanSquaredError: sum of squared errors / number of elements
        def __init__(self, *a, **k):
            super().__init__()
            self.add_state("sse", torch.tensor(0.0))
            self.add_state("n", torch.tensor(0.0))

        def update(self, preds, target):
            self.sse = self.sse + ((preds - target) ** 2).sum()
            self.n = self.n + target.numel()

        def compute(self):
            return self.sse / self.n
    sys.modules["torchmetrics"].Metric = Metric
    sys.modules["torchmetrics"].MeanSquaredError = MSE
    sys.modules["torchmetrics"].MetricCollection = Metric
    TL = _fresh_metrics_modules()
    from src.ConStruct.models import pos_transformer_model as P3
    C, n_nodes, dims, n_layers = 5, [11, 6, 9, 2], fx.TINY_DIMS, 2
    mask = fx.ragged_mask(n_nodes)
    B, n = mask.shape
    lab0 = _spanning_tree_labels(n_nodes, n, C, 411)
    E0 = fx.one_hot_edges(lab0, mask, C)
    X0 = mask[..., None].float()
    g = torch.Generator().manual_seed(412)
    pos = torch.randn(B, n, 3, generator=g)
    rad0 = (torch.rand(B, n, n, generator=g) + 0.2)
    rad0 = (0.5 * (rad0 + rad0.transpose(1, 2)) * (lab0 > 0))[..., None]
    dense = R.U.PlaceHolder(X=X0, charges=torch.zeros(B, n, 0), E=E0, y=torch.zeros(B, 0), node_mask=mask, pos=pos, radius=rad0)
    lab_t = torch.where(torch.rand(B, n, n, generator=g) < 0.5, lab0, torch.zeros_like(lab0))      # a noisy state: half the edges absorbed
    lab_t = torch.triu(lab_t, 1)
    lab_t = lab_t + lab_t.transpose(1, 2)
    Et = fx.one_hot_edges(lab_t, mask, C)
    t = torch.rand(B, 1, generator=g)
    model = P3.Graph3DTransformer(n_layers=n_layers, input_dims=fx.dims_ns(1, C, 1), hidden_mlp_dims=dims["hidden_mlp_dims"],
                                  hidden_dims=dims["hidden_dims"], output_dims=fx.dims_ns(1, C, 0), dropout=0.0).train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = fx.make_state_dict(shapes, seed=413)
    model.load_state_dict(sd)
    pred = model(R.U.PlaceHolder(X=X0, E=Et, y=t, pos=pos, radius=torch.zeros(B, n, n, 1), node_mask=mask))
    loss, _ = TL.TrainLoss(lambda_train=[1, 2, 5, 0])(masked_pred=pred, masked_true=dense, log=False)
    loss.backward()
    grads, unused = {}, []
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            unused.append(k)
        elif p_.grad.numel() <= 4096:
            grads["grad/" + k] = p_.grad
        else:
            grads["gradhead/" + k] = p_.grad.flatten()[:1024]
            grads["gradstat/" + k] = torch.stack([p_.grad.double().norm(), p_.grad.double().sum()])
    save("g10_training_3d_tiny", node_mask=mask, C=C, n_layers=n_layers, labels0=lab0, labels_t=lab_t, t=t, pos=pos, radius0=rad0,
         logits_E=pred.E.detach(), logits_X=pred.X.detach(), radius=pred.radius.detach(), loss=loss.detach(), sd_seed=413,
         unused=np.array(unused), **grads)
    print(f"g10: loss {float(loss.detach()):.6f}, {len(unused)} parameters without gradient")


def _timm_stand_in():
    """``timm`` is absent here; sit_model.py takes two things from it: the CONSTRUCTOR of ``Attention`` (the forward is the
    reference's own ``AttentionWithPads.forward``) and ``Mlp``.  The stand-in states their published layout -- qkv / proj
    Linears, head scale, fc1 / activation / fc2 -- and nothing else; every line of arithmetic that runs, apart from Mlp's
    fc2(act(fc1(x))), is the reference's."""
    import torch.nn as nn

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
            super().__init__()
            self.num_heads = num_heads
            self.scale = (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(0.0)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(0.0)

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, act_layer=nn.GELU, drop=0.0, **kw):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))
    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.vision_transformer = types.ModuleType("timm.models.vision_transformer")
    timm.models.vision_transformer.Attention = Attention
    timm.models.vision_transformer.Mlp = Mlp
    timm.optim = types.ModuleType("timm.optim")
    sys.modules.update({"timm": timm, "timm.models": timm.models, "timm.models.vision_transformer": timm.models.vision_transformer,
                        "timm.optim": timm.optim})


def gen_sit(R):
    """H1: the reference's SiT-T denoiser of the node-coordinate DDPM (sit_model.py) on ragged batches, with the timm
    stand-in above.  Weights: make_state_dict (NOT the zero adaLN initialisation, which would make every block the identity)."""
    _timm_stand_in()
    from src.bootstrap.node_coordinate_diffusion import sit_model as SM
    for name, n_nodes, seed in (("a", [9, 5, 2, 12], 501), ("b", [40, 33], 502)):
        torch.manual_seed(seed)
        model = SM.SiT_models["SiT-T"](in_channels=3).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = fx.make_state_dict(shapes, seed=seed)
        model.load_state_dict(sd)
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        x = torch.randn(B, n, 3)
        t = torch.tensor([999, 500, 1, 37][:B] if B == 4 else [250, 3])
        with torch.no_grad():
            out = model(x, t, mask)
        save(f"h1_sit_{name}", x=x, t=t, node_mask=mask, out=out, seed=seed, shape_keys=np.array(sorted(shapes)),
             shape_vals=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]))


def gen_sit_sample(R):
    """H2: the reference's ancestral sampler itself -- ``NodeCoordsDiffusionModel.sample``
    (cow_diff_nodes_with_3d_checks.py:63-81 and the identical atm_ twin) run from the REAL module.  Its module-level imports
    that are absent here and untouched by ``__init__`` / ``sample`` (open3d, timm.optim, torch_geometric.loader / .utils,
    the dataset class, environment_setup) are stubbed -- ordinary ModuleNotFoundErrors; the SiT network it builds is the
    reference's with the timm stand-in of gen_sit.  ``torch.randn`` / ``torch.randn_like`` are wrapped for the duration of the
    call so that every normal draw of the loop is RECORDED (the values are torch's own); the fixture holds x_T, the per-step
    draws and the result, for a whole short chain (noise_steps = 10: down to i = 1, the step without noise) and for the
    whole shipped 1000-step schedule on a tiny batch."""
    _timm_stand_in()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    before = dict(sys.modules)
    stub("open3d")
    tgl = stub("torch_geometric.loader", DataLoader=object)
    sys.modules["torch_geometric"].loader = tgl
    sys.modules["torch_geometric.utils"].to_dense_batch = None
    stub("src.ConStruct.datasets.cow_dataset", CoWGraphDataset=object)
    stub("src.ConStruct.datasets.atm_dataset", ATMGraphDataset=object)
    stub("src.environment_setup", PROJECT_ROOT_DIR="/nonexistent")
    for ds in ("cow", "atm"):      # import_reference() registered a stand-in for NoiseModel's sake: take it out
        sys.modules.pop(f"src.bootstrap.node_coordinate_diffusion.{ds}_diff_nodes_with_3d_checks", None)
    import importlib
    NC = importlib.import_module("src.bootstrap.node_coordinate_diffusion.cow_diff_nodes_with_3d_checks")
    real_randn, real_like = torch.randn, torch.randn_like
    for name, n_nodes, steps, seed in (("short", [9, 5, 12, 1], 10, 701), ("full", [7, 3], 1000, 702)):
        torch.manual_seed(seed)
        ncd = NC.NodeCoordsDiffusionModel(noise_steps=steps, device="cpu")
        shapes = {k: tuple(v.shape) for k, v in ncd.model.state_dict().items()}
        sd = fx.make_state_dict(shapes, seed=seed)
        ncd.model.load_state_dict(sd)
        mask = fx.ragged_mask(n_nodes)
        B, n = mask.shape
        draws = []

        def rec_randn(*a, **k):
            v = real_randn(*a, **k)
            draws.append(v.clone())
            return v

        def rec_like(x, *a, **k):
            v = real_like(x, *a, **k)
            draws.append(v.clone())
            return v
        torch.randn, torch.randn_like = rec_randn, rec_like
        seen = []                      # the state entering every iteration = the network's input (a pre-forward hook, no edit)
        hook = ncd.model.register_forward_pre_hook(lambda _m, args: seen.append(args[0].clone()))
        try:
            out = ncd.sample(torch.zeros(B, n, 3), mask)
        finally:
            torch.randn, torch.randn_like = real_randn, real_like
            hook.remove()
        assert len(draws) == steps - 1 and len(seen) == steps - 1, (len(draws), steps)   # x_T + one per iteration with i > 1
        at = [k for k in (1, 4, 8, 32, 128, 512) if k < steps - 1]
        save(f"h2_sit_sample_{name}", node_mask=mask, noise_steps=steps, seed=seed, x_T=draws[0], noise=torch.stack(draws[1:]),
             out=out, after_steps=np.array(at), x_after=torch.stack([seen[k] for k in at]), shape_keys=np.array(sorted(shapes)),
             shape_vals=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]))
        print(f"h2 {name}: |out| max {float(out.abs().max()):.4f}")
    for k in [k for k in sys.modules if k not in before]:
        del sys.modules[k]
    sys.modules.update(before)


def main():
    R = import_reference()
    extra = {"sit": gen_sit, "sit_sample": gen_sit_sample, "training_3d": gen_training_3d, "graph3d": gen_graph3d, "training": gen_training, "extra_large": gen_extra_features_large, "gt_large": gen_graph_transformer_large}
    if len(sys.argv) == 2 and sys.argv[1] in extra:
        return extra[sys.argv[1]](R)
    if sys.argv[1:] == ["projector"]:
        return gen_projector(R)
    if sys.argv[1:] == ["sample_plan"]:
        return gen_sample_plan(R)
    gen_block_and_layer(R)
    gen_graph_transformer(R)
    gen_pooling(R)
    gen_extra_features(R)
    gen_schedules(R)
    gen_reverse_step(R)
    gen_sample_discrete(R)
    gen_placeholder(R)
    gen_projector(R)
    gen_sample_plan(R)
    gen_training(R)
    gen_extra_features_large(R)
    gen_graph_transformer_large(R)
    gen_graph3d(R)
    gen_training_3d(R)
    gen_sit(R)
    gen_sit_sample(R)


if __name__ == "__main__":
    main()
