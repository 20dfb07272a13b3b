"""GPU tests of the reverse-sampling loop: the device chain against a CPU oracle chain (bit-exact labels given the
same Philox stream), the PlaceHolder route against the state route, shard independence and chain invariants."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import denoiser as D
from oracle import fixtures as fx
from oracle import noise as N
from test_host_logic import make_noise_model

pytestmark = pytest.mark.gpu
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)


def tiny_model(C, in_E, in_X, in_y, seed=5, dims=fx.TINY_DIMS, n_layers=2):
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    model = GraphTransformer(n_layers, fx.dims_ns(in_X, in_E, in_y), dims["hidden_mlp_dims"], dims["hidden_dims"],
                             fx.dims_ns(1, C, 0)).eval()
    sd = fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    return model.cuda(), sd


def allowed_flips(dev_labels, r, u0, what, tol=1e-4):
    """Device labels against one oracle step from the same state.  The device's logits carry fp32 round-off (1e-4 of their
    scale at most, the denoiser tests' bound), so a draw may land on the other side of a class boundary only where its
    uniform sits that close to the boundary of the oracle's cumulative distribution; anywhere else a difference is an error.
    -> number of such boundary draws."""
    diff = dev_labels != r["labels_E"]
    if not bool(diff.any()):
        return 0
    cdf = r["prob_E"].double().cumsum(-1)
    margin = (cdf - torch.as_tensor(u0, dtype=torch.float64)[..., None]).abs().amin(-1)
    margin = torch.minimum(margin, margin.transpose(1, 2))                     # a pair is drawn once, for (i, j) with i < j
    assert bool((margin[diff] < tol).all()), f"{what}: labels differ away from a class boundary (margin {float(margin[diff].max()):.2e})"
    return int(diff.sum())


def test_chain_matches_cpu_oracle_bit_exactly():
    """No extra features (their eigenvectors are basis-dependent on empty graphs): T=12 absorbing-edge chain, the oracle
    stepping from the device's state: identical labels, except that a draw within fp32 round-off of a class boundary may
    fall either way (see allowed_flips)."""
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    C, T, top_k, seed, base = 6, 12, 4, 99, 40
    model, sd = tiny_model(C, C, 1, 1)
    nm = make_noise_model("absorbing_edges", C, T=T, top_k=top_k)
    sampler = DiscreteSampler(model, nm)
    n_nodes = [9, 4, 7, 1]
    labels, cands, mask, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, return_state=True)
    torch.cuda.synchronize()
    # ---- the same chain on the CPU oracle
    m = fx.ragged_mask(n_nodes)
    B, n = m.shape
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    tr = N.Transition(N.Schedule(T, "absorbing_edges", NU), [1.0], np.ones(C, np.float32) / C)
    dd = D.DenoiserDims(2, fx.TINY_DIMS["hidden_dims"]["n_head"], 1, C, 0)
    lab = torch.where(live, 0, -1)
    X = m[..., None].float()
    flips = 0
    for step, s in enumerate(reversed(range(T))):
        t = s + 1
        E = fx.one_hot_edges(lab.clamp(min=0), m, C)
        y = torch.full((B, 1), t / T)
        _, _, logits, _ = D.graph_transformer_forward(sd, dd, X, E, y, m)
        u0, uk = N.philox_edge_uniforms(seed, s, base + np.arange(B), n, top_k)
        r = N.reverse_step(tr, X, E, torch.zeros(B, n, 1), logits, m, torch.full((B, 1), t), torch.full((B, 1), s),
                           np.zeros((B, n), np.float32), u0, uk)
        dl, dc, _, _ = sampler.sample_batch(n_nodes, seed=seed, graph_id_base=base, steps=step + 1, return_state=True)
        dl = dl.cpu().long()
        flips += allowed_flips(dl, r, u0, f"step {step}")
        if torch.equal(dl, r["labels_E"]):
            assert int(((dc.cpu().long() * live[..., None]) != r["extra_info"]).sum()) == 0
        lab = dl                                                               # the oracle follows the device's chain
    assert flips <= 4
    assert int((lab > 0).sum()) > 0          # the chain did create edges


def make_cow_sampler(n_layers=2, T=20):
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    C = 14
    c = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5,
                                              num_eigenvalues=9, eigenfeatures=True))
    ef = ExtraFeatures(c, SimpleNamespace(max_n_nodes=32))
    model, _ = tiny_model(C, 25, 10, 43, n_layers=n_layers)
    nm = make_noise_model("absorbing_edges", C, T=T, top_k=4)
    return DiscreteSampler(model, nm, ef), nm


def test_placeholder_route_equals_state_route_with_extra_features():
    """One reverse step through forward()/sample_zs_from_zt (reference types) == the same step of the state loop."""
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    sampler, nm = make_cow_sampler()
    n_nodes = [12, 7, 10]
    seed = 3
    lab1, cand1, mask, _ = sampler.sample_batch(n_nodes, seed=seed, steps=1, return_state=True)
    lab3, cand3, _, _ = sampler.sample_batch(n_nodes, seed=seed, steps=3, return_state=True)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    z = nm.sample_limit_dist(mask, generator=gen)
    T = nm.T
    for s in (T - 1, T - 2, T - 3):
        s_int = torch.full((len(n_nodes), 1), s, dtype=torch.long, device="cuda")
        z = sampler.sample_zs_from_zt(z, s_int, Rng(seed=seed, step=s))
        if s == T - 1:
            assert torch.equal(z._labels, lab1)
    assert torch.equal(z._labels, lab3) and torch.equal(z._cands, cand3)
    # reference-typed view of the state
    E = z.E.cpu()
    assert E.shape[-1] == 14 and torch.equal(E, E.transpose(1, 2))
    assert torch.equal(E.argmax(-1)[E.sum(-1) > 0], z._labels.cpu().long()[E.sum(-1) > 0])
    assert z.extra_info.dtype == torch.int64 and torch.equal(z.extra_info[..., 0].cpu()[E.sum(-1) > 0], z._labels.cpu().long()[E.sum(-1) > 0])


def test_shard_independence_and_chain_invariants():
    sampler, nm = make_cow_sampler(T=10)
    n_nodes = [16, 9, 13, 5, 11, 8]
    labels, cands, mask, _ = sampler.sample_batch(n_nodes, seed=11, graph_id_base=0, n_max=16, return_state=True)
    a, _, _, _ = sampler.sample_batch(n_nodes[:3], seed=11, graph_id_base=0, n_max=16, return_state=True)
    b, _, _, _ = sampler.sample_batch(n_nodes[3:], seed=11, graph_id_base=3, n_max=16, return_state=True)
    assert torch.equal(torch.cat([a, b]), labels)        # same graphs whatever the sharding
    L = labels.cpu()
    m = mask.cpu()
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(16, dtype=torch.bool)[None]
    assert torch.equal(L, L.transpose(1, 2)) and bool((L[~live] == -1).all()) and bool((L[live] >= 0).all())
    out = sampler.sample_batch(n_nodes, seed=11, n_max=16)
    assert out.E.shape == (6, 16, 16) and int(out.E.min()) == -1 and bool((out.E[live.cuda()] == labels[live.cuda()].long()).all())
    assert bool((out.X.cpu()[~m] == -1).all())


def test_absorbing_chain_never_removes_an_edge():
    sampler, nm = make_cow_sampler(T=10)
    n_nodes = [14, 6]
    prev = None
    for steps in range(1, 11):
        lab, _, _, _ = sampler.sample_batch(n_nodes, seed=2, steps=steps, return_state=True)
        if prev is not None:
            had = prev > 0
            assert torch.equal(lab[had], prev[had])
        prev = lab
    assert int((prev > 0).sum()) > 0


def test_projector_state_route_equals_placeholder_route_in_the_loop():
    """Row a9 inside the loop: projector.py on the int8 device state (project_labels) gives the chain that the same
    projector gives through the reference-typed route (one-hot z_s.E + int64 extra_info, project()), same np.random
    seed; the ATM projector's result is a forest."""
    from semantically_consistent_graph_generation_amd import projector as PJ
    sampler, nm = make_cow_sampler(T=10)
    n_nodes = [14, 6, 11, 1]
    for cls in (PJ.CoWProjector, PJ.CoWVanillaProjector):
        made = []

        def via_state(z_T, dev):
            made.append(cls(z_T, dev))
            return made[-1]
        via_state.project_labels = True          # the sampler keys the route on this attribute

        def via_placeholder(z_T, dev):
            made.append(cls(z_T, dev))
            return SimpleNamespace(project=made[-1].project)
        np.random.seed(5)
        a, _, mask, _ = sampler.sample_batch(n_nodes, seed=4, projector=via_state, return_state=True)
        np.random.seed(5)
        b, _, _, _ = sampler.sample_batch(n_nodes, seed=4, projector=via_placeholder, return_state=True)
        free, _, _, _ = sampler.sample_batch(n_nodes, seed=4, return_state=True)
        assert torch.equal(a, b)
        assert made[0].blocked_edges == made[1].blocked_edges
        assert int((a > 0).sum()) > 0 and not torch.equal(a, free)       # the projector did intervene
        L = a.cpu()
        assert torch.equal(L, L.transpose(1, 2))
        for bi, g in enumerate(made[0].nx_graphs_list):                  # stored graph == adjacency of the state
            k = g.number_of_nodes()
            assert np.array_equal((L[bi, :k, :k] > 0).numpy(), g.to_numpy_array() != 0)
    # ATM setting: 5 classes, forest constraint
    from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
    model, _ = tiny_model(5, 5, 1, 1)
    smp = DiscreteSampler(model, make_noise_model("absorbing_edges", 5, T=10, top_k=4))
    np.random.seed(6)
    lab, _, _, _ = smp.sample_batch(n_nodes, seed=8, projector=PJ.ATMTreeProjector, return_state=True)
    for bi, k in enumerate(n_nodes):
        g = PJ.LabelGraph(k)
        for i, j in (lab[bi, :k, :k].cpu().triu(1) > 0).nonzero().tolist():
            g.add_edge(i, j, 1)
        assert PJ.has_no_cycles(g)


def test_pipelined_batches_equal_sequential_batches():
    """sample_batches interleaves the reverse steps of independent batches around the projector's host half; with a
    random stream per chain the result is what each batch gives alone."""
    from semantically_consistent_graph_generation_amd import projector as PJ
    sampler, nm = make_cow_sampler(T=10)

    def factory(cls, seed):
        rs = np.random.RandomState(seed)

        def make(z_T, dev):
            return cls(z_T, dev, rng=rs)
        make.project_labels = True
        return make

    def requests():
        return [dict(n_nodes=[14, 6, 11], seed=4, projector=factory(PJ.CoWProjector, 1), return_state=True),
                dict(n_nodes=[9, 12], seed=5, graph_id_base=3, projector=factory(PJ.CoWVanillaProjector, 2), return_state=True),
                dict(n_nodes=[8, 3], seed=6, graph_id_base=5, return_state=True),
                dict(n_nodes=[13], seed=7, graph_id_base=7, steps=4, projector=factory(PJ.CoWProjector, 3), return_state=True)]
    together = sampler.sample_batches(requests())
    alone = [sampler.sample_batch(**r) for r in requests()]
    for (la, ca, _, _), (lb, cb, _, _) in zip(together, alone):
        assert torch.equal(la, lb) and torch.equal(ca, cb)
    assert int((together[0][0] > 0).sum()) > 0


def test_sample_n_graphs_is_independent_of_the_pipeline_depth():
    from semantically_consistent_graph_generation_amd import projector as PJ
    sampler, nm = make_cow_sampler(T=6)
    first, rest = [9, 14, 5], [12, 7, 11, 8, 6, 10, 13]

    def run(depth):
        rs = [np.random.RandomState(40 + k) for k in range(8)]
        it = iter(rs)

        def make(z_T, dev):
            return PJ.CoWProjector(z_T, dev, rng=next(it))
        make.project_labels = True
        return sampler.sample_n_graphs(first, rest, batch_size=2, seed=9, projector=make, pipeline=depth, return_state=True,
                                       n_max=16)
    a, b = run(1), run(3)
    assert [r[0].shape[0] for r in a] == [3, 4, 3]            # int(1.8*2)=3 first, then 1.8*2=3.6 -> 4 per batch
    for (la, _, _, _), (lb, _, _, _) in zip(a, b):
        assert torch.equal(la, lb)


def test_keep_chain_frames_follow_the_reference_rule():
    """:622-631, :665-677, :689-699: frame N-1-(s*N)//T after every step s with (s*N) % T == 0, last frame = result."""
    sampler, nm = make_cow_sampler(T=10)
    n_nodes = [12, 7, 9]
    out = sampler.sample_batch(n_nodes, seed=21, keep_chain=2, number_chain_steps=5)
    ch = out.chains
    assert ch.E.shape == (5, 2, 12, 12) and ch.X.shape == (5, 2, 12) and ch.pos.shape == (5, 2, 12, 3)
    assert torch.equal(ch.E[-1], out.E[:2]) and torch.equal(ch.X[-1], out.X[:2])
    for frame, steps in ((0, 2), (1, 4), (2, 6), (3, 8)):          # s = 8, 6, 4, 2 are reached after 2, 4, 6, 8 steps
        part = sampler.sample_batch(n_nodes, seed=21, steps=steps)
        assert torch.equal(ch.E[frame], part.E[:2]), frame
    assert bool((ch.E[:, 1, 7:, :] == -1).all()) and bool((ch.X[:, 1, 7:] == -1).all())      # graph 1 has 7 nodes
    assert sampler.sample_batch(n_nodes, seed=21).chains is None
    state = sampler.sample_batch(n_nodes, seed=21, keep_chain=5, number_chain_steps=2, return_state=True)
    assert len(state) == 5 and state[4].E.shape == (2, 3, 12, 12)
