"""GPU parity of the extra-feature pass against the reference's golden vectors (adjacency features: exact / 1e-6)
and the CPU oracle (Laplacian, eigenvalue features; eigenvectors only where the spectrum is simple)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import extra_features as XF
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def make_ef(max_n, cycle=True):
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    c = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=cycle, num_eigenvectors=5,
                                              num_eigenvalues=9, eigenfeatures=True))
    return ExtraFeatures(c, SimpleNamespace(max_n_nodes=max_n))


def noisy(g):
    m = torch.from_numpy(g["node_mask"]).cuda()
    return SimpleNamespace(X=torch.from_numpy(g["X"]).cuda(), E=torch.from_numpy(g["E"]).cuda(),
                           y=torch.zeros(m.shape[0], 0, device="cuda"), node_mask=m)


@pytest.mark.parametrize("name,maxn", [("f5_extra_features", 8), ("f5_extra_features_random", 32)])
def test_adjacency_features_match_reference(name, maxn):
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    g = load_golden(name)
    ef = make_ef(maxn)
    d = noisy(g)
    C = int(g["C"])
    labels = NoiseModel._labels_of(d, C)
    x, y, e, lap = ef.adjacency_features(labels, d.node_mask, C)
    torch.cuda.synchronize()
    np.testing.assert_allclose(x.cpu().numpy(), g["adj_x"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), g["adj_y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(e.cpu().numpy(), g["adj_e"], rtol=1e-5, atol=1e-6)
    L = XF.laplacian_for_eigh(torch.from_numpy(g["E"]), torch.from_numpy(g["node_mask"]))
    np.testing.assert_array_equal(lap.cpu().numpy(), L.numpy())
    # the stand-alone Laplacian entry point (scgg_laplacian) gives the same matrix
    np.testing.assert_array_equal(ef.laplacian(labels, d.node_mask).cpu().numpy(), L.numpy())


def test_onehot_and_features_written_into_strided_input():
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    g = load_golden("f5_extra_features_random")
    ef = make_ef(32)
    d = noisy(g)
    C = 14
    labels = NoiseModel._labels_of(d, C)
    B, n = d.node_mask.shape
    dst = torch.full((B, n, n, C + 11), 9.0, device="cuda")
    ef.adjacency_features(labels, d.node_mask, C, e_dst=dst, e_off=C, want_onehot=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dst[..., :C].cpu().numpy(), g["E"])
    np.testing.assert_allclose(dst[..., C:].cpu().numpy(), g["adj_e"], rtol=1e-5, atol=1e-6)


def test_cycle_features_off_zeroes_the_cycle_channels():
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    g = load_golden("f5_extra_features")
    d = noisy(g)
    C = int(g["C"])
    x, y, e, _ = make_ef(8, cycle=False).adjacency_features(NoiseModel._labels_of(d, C), d.node_mask, C)
    ox, oy, oe = XF.adjacency_features(torch.from_numpy(g["X"]), torch.from_numpy(g["E"]), torch.from_numpy(g["node_mask"]),
                                       cycle_features=False)
    np.testing.assert_allclose(x.cpu().numpy(), ox.numpy(), atol=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), oy.numpy(), atol=1e-6)
    np.testing.assert_allclose(e.cpu().numpy(), oe.numpy(), atol=1e-6)


def test_full_extra_features_call():
    g = load_golden("f5_extra_features")
    out = make_ef(8)(noisy(g))
    torch.cuda.synchronize()
    X, E, y = out.X.cpu().numpy(), out.E.cpu().numpy(), out.y.cpu().numpy()
    assert X.shape == g["out_X"].shape and y.shape == g["out_y"].shape
    np.testing.assert_allclose(E, g["out_E"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(X[..., :3], g["out_X"][..., :3], atol=1e-6)
    np.testing.assert_allclose(y, g["out_y"], rtol=1e-4, atol=3e-5)           # includes the eigenvalue features
    # LCC indicator on graphs with a unique largest component; first eigenvectors of the tree up to sign
    np.testing.assert_array_equal(X[[0, 1, 2, 3, 4, 7], :, 3], g["out_X"][[0, 1, 2, 3, 4, 7], :, 3])
    for k in range(2):
        a, b = X[0, :, 4 + k], g["out_X"][0, :, 4 + k]
        np.testing.assert_allclose(a * np.sign((a * b).sum()), b, atol=3e-4)


def test_jacobi_eigensolver_properties():
    """A V = V diag(w), V orthonormal, w ascending and equal to LAPACK's eigenvalues, at config-2 scale."""
    gen = torch.Generator().manual_seed(8)
    nn_ = torch.randint(1, 33, (512,), generator=gen)
    nn_[:4] = torch.tensor([32, 1, 2, 31])
    mask = fx.ragged_mask(nn_, 32)
    dens = torch.rand(512, generator=gen)[:, None, None]
    lab = fx.random_labels(mask, 5, 1.0, 6) * (torch.rand(512, 32, 32, generator=gen) < dens)
    lab = torch.triu(lab, 1); lab = lab + lab.transpose(1, 2)
    E = fx.one_hot_edges(lab, mask, 5)
    L = XF.laplacian_for_eigh(E, mask)
    ef = make_ef(32)
    evl, evc, w, V = ef.eigen_features(L.cuda(), mask.cuda(), want_raw=True)
    torch.cuda.synchronize()
    w, V = w.cpu().double(), V.cpu().double()
    Ld = L.double()
    scale = Ld.abs().amax(dim=(1, 2)).clamp(min=1.0)
    res = (Ld @ V - V * w[:, None, :]).abs().amax(dim=(1, 2)) / scale
    assert float(res.max()) < 2e-5
    orth = (V.transpose(1, 2) @ V - torch.eye(32, dtype=torch.float64)[None]).abs().amax(dim=(1, 2))
    assert float(orth.max()) < 2e-5
    assert bool((w[:, 1:] >= w[:, :-1]).all())
    wl = torch.linalg.eigvalsh(Ld)
    assert float(((w - wl).abs().amax(dim=1) / scale).max()) < 2e-5
    # eigenvalue features against the oracle (LAPACK)
    oevl, _ = XF.eigen_features(E, mask)
    np.testing.assert_allclose(evl.cpu().numpy(), oevl.numpy(), rtol=1e-4, atol=3e-5)


def test_features_at_config2_scale_are_consistent():
    """B=512, n=32 (BASELINE config 2): properties that do not need the oracle at full size + oracle on a slice."""
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    gen = torch.Generator().manual_seed(3)
    nn_ = torch.randint(12, 33, (512,), generator=gen)
    mask = fx.ragged_mask(nn_, 32)
    lab = fx.random_labels(mask, 14, 0.08, 4)
    live = (mask[:, :, None] & mask[:, None, :]) & ~torch.eye(32, dtype=torch.bool)[None]
    lab8 = torch.where(live, lab, torch.full_like(lab, -1)).to(torch.int8).cuda()
    ef = make_ef(32)
    x, y, e, lap = ef.adjacency_features(lab8, mask.cuda(), 14)
    torch.cuda.synchronize()
    e, y, x = e.cpu(), y.cpu(), x.cpu()
    assert torch.equal(e, e.transpose(1, 2))                      # every edge feature is symmetric
    assert float(e.min()) >= 0 and float(e.max()) <= 1
    np.testing.assert_allclose(y[:, 4:16].sum(1).numpy(), 1.0, atol=1e-5)     # degree histogram
    np.testing.assert_allclose(y[:, 17:].sum(1).numpy(), 1.0, atol=1e-5)      # edge-class histogram
    E = fx.one_hot_edges(lab[:16], mask[:16], 14)
    ox, oy, oe = XF.adjacency_features(mask[:16, :, None].float(), E, mask[:16])
    np.testing.assert_allclose(x[:16].numpy(), ox.numpy(), atol=1e-6)
    np.testing.assert_allclose(y[:16].numpy(), oy.numpy(), atol=1e-6)
    np.testing.assert_allclose(e[:16].numpy(), oe.numpy(), rtol=1e-5, atol=1e-6)
