"""GPU parity of the reverse-step / sampling kernels (through the C ABI) against the reference's golden vectors
and the CPU oracle.  Sampled edge labels and candidates: bit-exact.  Probabilities: 1e-6 absolute."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fixtures as fx
from oracle import noise as N

pytestmark = pytest.mark.gpu
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)
STEPS = (500, 499, 250, 2, 1)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def expected_from_golden(g, p, m):
    """int labels (-1 on dead pairs) and masked candidates from the reference's one-hot E_s / extra_info."""
    E_s = torch.from_numpy(g[p + "E_s"])
    n = m.shape[1]
    pm = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    lab = torch.where(pm, E_s.argmax(-1), torch.full(E_s.shape[:-1], -1))
    return lab.numpy(), g[p + "extra_info"]


def sample_from_probs(prob, mask, top_k, u0=None, uk=None, seed=0, step=0, base=0):
    from semantically_consistent_graph_generation_amd import _lib
    L = _lib.lib()
    B, n, _, Cc = prob.shape
    labels = torch.full((B, n, n), 77, dtype=torch.int8, device="cuda")
    cands = torch.full((B, n, n, max(top_k, 1)), 77, dtype=torch.int8, device="cuda")
    rng = _lib.Rng(_lib.ptr(u0).value if u0 is not None else None, _lib.ptr(uk).value if uk is not None else None,
                   seed, step, base)
    _lib.check(L.scgg_sample_edges_from_probs(_lib.ptr(prob), _lib.ptr(mask), B, n, Cc, top_k, C.byref(rng),
                                              _lib.ptr(labels), _lib.ptr(cands) if top_k else None, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return labels.cpu().numpy(), cands.cpu().numpy()


def reverse_step(logits, labels_t, mask, t, s, sched, marg, top_k, u0=None, uk=None, seed=0, step=0, base=0,
                 want_prob=True):
    from semantically_consistent_graph_generation_amd import _lib
    L = _lib.lib()
    B, n, _, Cc = logits.shape
    labels = torch.full((B, n, n), 77, dtype=torch.int8, device="cuda")
    cands = torch.full((B, n, n, max(top_k, 1)), 77, dtype=torch.int8, device="cuda")
    prob = torch.full((B, n, n, Cc), -1.0, device="cuda") if want_prob else None
    beta, abar = sched
    rng = _lib.Rng(_lib.ptr(u0).value if u0 is not None else None, _lib.ptr(uk).value if uk is not None else None,
                   seed, step, base)
    t_int = torch.full((B,), t, dtype=torch.int32, device="cuda")
    s_int = torch.full((B,), s, dtype=torch.int32, device="cuda")
    _lib.check(L.scgg_reverse_step_edges(_lib.ptr(logits), _lib.ptr(labels_t), _lib.ptr(mask), _lib.ptr(t_int),
                                         _lib.ptr(s_int), _lib.ptr(beta), _lib.ptr(abar), _lib.ptr(marg), B, n, Cc, top_k,
                                         C.byref(rng), _lib.ptr(labels), _lib.ptr(cands), _lib.ptr(prob),
                                         _lib.stream_ptr()))
    torch.cuda.synchronize()
    return labels.cpu().numpy(), cands.cpu().numpy(), (prob.cpu().numpy() if want_prob else None)


def edge_tables(transition, C_e, e_marg=None):
    s = N.Schedule(500, transition, NU)
    if transition == "absorbing_edges":
        beta, abar = s.betas_abs[:, 2].float(), s.alphas_abs_bar[:, 2].float()
        marg = torch.zeros(C_e); marg[0] = 1
    else:
        beta, abar = s.betas[:, 2].float(), s.alphas_bar[:, 2].float()
        marg = torch.as_tensor(e_marg, dtype=torch.float32)
    return (beta.cuda().contiguous(), abar.cuda().contiguous()), marg.cuda().contiguous()


@pytest.mark.parametrize("name", ["cow_abs", "atm_abs", "marg"])
def test_sampler_bit_exact_on_reference_probabilities(name):
    g = load_golden(f"f8_reverse_step_{name}")
    m = torch.from_numpy(g["node_mask"])
    top_k = int(g["top_k"])
    for t in STEPS:
        p = f"t{t}/"
        lab, cand = sample_from_probs(dev(g[p + "prob_E"]), dev(g["node_mask"], torch.uint8), top_k,
                                      dev(g[p + "u0"]), dev(g[p + "uk"]))
        exp_lab, exp_extra = expected_from_golden(g, p, m)
        np.testing.assert_array_equal(lab, exp_lab)
        np.testing.assert_array_equal(cand * (exp_lab >= 0)[..., None], exp_extra)


@pytest.mark.parametrize("name", ["cow_abs", "atm_abs", "marg"])
def test_fused_reverse_step_matches_reference(name):
    g = load_golden(f"f8_reverse_step_{name}")
    m = torch.from_numpy(g["node_mask"])
    Cc, top_k = int(g["C"]), int(g["top_k"])
    sched, marg = edge_tables(str(g["transition"]), Cc, g["e_marginals"] if "e_marginals" in g else None)
    n = m.shape[1]
    live = ((m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]).numpy()
    mismatches = 0
    for t in STEPS:
        p = f"t{t}/"
        lab_t = torch.from_numpy(g[p + "labels_t"]).to(torch.int8)
        lab_t = torch.where(torch.from_numpy(live), lab_t, torch.full_like(lab_t, -1))
        lab, cand, prob = reverse_step(dev(g[p + "logits_E"]), lab_t.cuda().contiguous(), dev(g["node_mask"], torch.uint8),
                                       t, t - 1, sched, marg, top_k, dev(g[p + "u0"]), dev(g[p + "uk"]))
        np.testing.assert_allclose(prob[live], g[p + "prob_E"][live], rtol=2e-5, atol=1e-6)
        exp_lab, exp_extra = expected_from_golden(g, p, m)
        mismatches += int((lab != exp_lab).sum()) + int(((cand * (exp_lab >= 0)[..., None]) != exp_extra).sum())
        assert (lab == lab.transpose(0, 2, 1)).all()
    assert mismatches == 0


def test_sampler_without_topk_matches_reference():
    g = load_golden("f9_sample_discrete_notopk")
    m = torch.from_numpy(g["node_mask"])
    lab, _ = sample_from_probs(dev(g["prob_E"]), dev(g["node_mask"], torch.uint8), 0, dev(g["u0"]), None)
    n = m.shape[1]
    live = ((m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]).numpy()
    np.testing.assert_array_equal(lab[live], g["E"][live])
    assert (lab[~live] == -1).all()


def test_philox_stream_matches_numpy_twin():
    """Device Philox == oracle/noise.py Philox: sample with the kernel's own RNG and with the oracle sampler fed by
    the numpy stream; also independence from the batch split (graph_id_base)."""
    g = torch.Generator().manual_seed(5)
    mask = fx.ragged_mask([13, 32, 20, 7], 32)
    B, n = mask.shape
    Cc, top_k, seed, step = 14, 4, 0x1234567ABCDEF, 321
    prob = torch.rand(B, n, n, Cc, generator=g)
    prob[..., 5] = 0
    prob = prob / prob.sum(-1, keepdim=True)
    lab, cand = sample_from_probs(prob.cuda(), mask.cuda().view(torch.uint8), top_k, seed=seed, step=step, base=100)
    u0, uk = N.philox_edge_uniforms(seed, step, 100 + np.arange(B), n, top_k)
    es, extra = N.sample_edges(prob, mask, u0, uk)
    live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    np.testing.assert_array_equal(lab[live.numpy()], es.numpy()[live.numpy()])
    np.testing.assert_array_equal(cand * live.numpy()[..., None], (extra * live[..., None]).numpy())
    # shard independence: graphs 2..3 sampled alone with base 102 give the same labels
    lab2, cand2 = sample_from_probs(prob[2:].contiguous().cuda(), mask[2:].contiguous().cuda().view(torch.uint8), top_k,
                                    seed=seed, step=step, base=102)
    np.testing.assert_array_equal(lab2, lab[2:])
    np.testing.assert_array_equal(cand2, cand[2:])


def test_absorbing_properties_at_scale():
    """Size-independent properties at the BASELINE config-2 shape (B=512, n=32, C=14): present edges never change
    under the absorbing reverse step, output symmetric, dead pairs -1, new edges only where z_t had none."""
    B, n, Cc, top_k = 512, 32, 14, 4
    g = torch.Generator().manual_seed(9)
    n_nodes = torch.randint(12, 33, (B,), generator=g)
    n_nodes[0] = 32
    mask = fx.ragged_mask(n_nodes, n)
    lab_t = fx.random_labels(mask, Cc, 0.1, 3)
    live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
    lab_t8 = torch.where(live, lab_t, torch.full_like(lab_t, -1)).to(torch.int8)
    logits = torch.randn(B, n, n, Cc, generator=g)
    sched, marg = edge_tables("absorbing_edges", Cc)
    lab, cand, _ = reverse_step(logits.cuda(), lab_t8.cuda(), mask.cuda().view(torch.uint8), 250, 249, sched, marg, top_k,
                                seed=7, step=249, want_prob=False)
    assert (lab == lab.transpose(0, 2, 1)).all()
    assert (cand == cand.transpose(0, 2, 1, 3)).all()
    assert (lab[~live.numpy()] == -1).all()
    had = (lab_t8.numpy() > 0)
    assert (lab[had] == lab_t8.numpy()[had]).all()
    assert (cand[..., 0][live.numpy()] == lab[live.numpy()]).all()
    assert (lab[live.numpy()] >= 0).all() and (lab.max() < Cc)
