"""Oracle: structural extra features (cycle counts from adjacency powers, reachability,
Adamic-Adar-like index, histograms, Laplacian eigen-features).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows /root/reference/src/ConStruct/diffusion/extra_features.py
  ExtraFeatures.__call__ :60-77, update_input_dims :43-58,
  AdjacencyFeatures :198-385 (calculate_kpowers :256-268, k3..k6_cycle :270-327,
  path_features :329-333, local_neighbors :335-356, histograms :358-385),
  EigenFeatures :80-195.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def adjacency(E):
    """A = sum over the non-"no-edge" classes (:207)."""
    return E[..., 1:].sum(dim=-1).float()


def adjacency_powers(A, K):
    """[K,B,n,n] with entry k = A^(k+1) (:256-264)."""
    out = [A]
    for _ in range(1, K):
        out.append(out[-1] @ A)
    return torch.stack(out, dim=0)


def _diag(M):
    return torch.diagonal(M, dim1=-2, dim2=-1)


def cycle_counts(k, d):
    """k: powers [K>=6,B,n,n], d: degree [B,n].  Returns x[B,n,3] (3/4/5-cycles per node)
    and y[B,4] (3..6-cycles per graph), un-clamped (:270-327)."""
    A, A2, A3, A4, A5, A6 = (k[i] for i in range(6))
    tri = _diag(A3)
    c4 = _diag(A4) - d * (d - 1) - (A @ d[..., None]).sum(dim=-1)
    c5 = _diag(A5) - 2 * tri * d - (A @ tri[..., None]).sum(dim=-1) + tri
    d2 = _diag(A2)
    c6 = (_diag(A6).sum(-1) - 3 * _diag(A3 ** 2).sum(-1) + 9 * (A * A2 ** 2).sum(dim=(-2, -1))
          - 6 * (d2 * _diag(A4)).sum(-1) + 6 * _diag(A4).sum(-1) - 4 * tri.sum(-1)
          + 4 * (d2 ** 3).sum(-1) + 3 * A3.sum(dim=(-2, -1)) - 12 * (d2 ** 2).sum(-1) + 4 * d2.sum(-1))
    x = torch.stack([tri / 2, c4 / 2, c5 / 2], dim=-1)
    y = torch.stack([tri.sum(-1) / 6, c4.sum(-1) / 8, c5.sum(-1) / 10, c6 / 12], dim=-1)
    return x, y


def local_neighbors(A, num_nodes):
    """Adamic-Adar-like index (:335-356).  Column j of A is divided by deg(j); 0/0 and the
    deg==1 case (1/sqrt|log 1| = inf) both end as 0 through nan_to_num(posinf=0)."""
    w = A / A.sum(dim=-1)[:, None, :]
    w = torch.nan_to_num(1.0 / torch.sqrt(torch.log(w).abs()), posinf=0)
    s = w @ w.transpose(-2, -1)
    n = A.shape[-1]
    s = s * (1.0 - torch.eye(n))[None]
    nn_ = num_nodes.float()
    return (s * (nn_.log() / nn_)[:, None, None])[..., None]


def adjacency_features(X, E, node_mask, num_degree=10, max_degree=10, cycle_features=True):
    """AdjacencyFeatures.__call__ (:206-254) -> x[B,n,3], y[B,4+num_degree+2+dx+de], e[B,n,n,max_degree+1]."""
    A = adjacency(E)
    k = adjacency_powers(A, max_degree)
    d = A.sum(dim=-1)
    kx, ky = cycle_counts(k, d)
    dist = (k != 0).float().permute(1, 2, 3, 0)
    ngb = local_neighbors(A, node_mask.sum(dim=1))
    ky = torch.clamp(ky, 0, 5) / 5

    deg = d.long().clamp(max=num_degree + 1)
    oh = F.one_hot(deg, num_degree + 2).float() * node_mask[..., None]
    dd = oh.sum(dim=1)
    sden = dd.sum(dim=-1, keepdim=True)
    dd = dd / torch.where(sden == 0, torch.ones_like(sden), sden)

    nd = X.sum(dim=1)
    sden = nd.sum(-1, keepdim=True)
    nd = nd / torch.where(sden == 0, torch.ones_like(sden), sden)
    ed = E.sum(dim=(1, 2))
    sden = ed.sum(-1, keepdim=True)
    ed = ed / torch.where(sden == 0, torch.ones_like(sden), sden)

    if not cycle_features:
        kx, ky, dist = torch.zeros_like(kx), torch.zeros_like(ky), torch.zeros_like(dist)
    x = torch.clamp(kx, 0, 5) / 5 * node_mask[..., None]
    y = torch.cat([ky, dd, nd, ed], dim=-1)
    e = torch.clamp(torch.cat([dist, ngb], dim=-1), 0, 5) / 5
    return x, y, e


def laplacian_for_eigh(E, node_mask):
    """Combinatorial Laplacian with padded nodes pushed to eigenvalue 2n (:88-94, :114-127)."""
    m = node_mask.float()
    A = adjacency(E) * m[:, None, :] * m[:, :, None]
    L = torch.diag_embed(A.sum(dim=-1)) - A
    L = (L + L.transpose(1, 2)) / 2
    n = A.shape[-1]
    pad = 2 * n * torch.eye(n)[None] * (1 - m)[:, None, :] * (1 - m)[:, :, None]
    return L * m[:, None, :] * m[:, :, None] + pad


def eigen_features(E, node_mask, num_eigenvectors=5, num_eigenvalues=9, eigh=torch.linalg.eigh, pad_noise=None):
    """EigenFeatures.compute_features (:87-112) -> evalue_feat[B,1+num_eigenvalues], evector_feat[B,n,1+num_eigenvectors].

    ``pad_noise`` [B,n]: the ``torch.randn(bs, n)`` the reference adds on the PADDED slots before taking the mode of the
    rounded first eigenvector (:171-173, "to prevent 0 from becoming the mode"); None draws it from torch's global
    generator exactly as the reference does (so ``torch.manual_seed`` reproduces a reference run).  The padded values
    are distinct with probability 1, so they only matter when no valid value occurs twice: then ``torch.mode`` returns the
    smallest of all n values, valid or not.  The HIP kernel takes the mode over the valid nodes only, i.e. the limit in
    which a padded value is never the smallest."""
    B, n = node_mask.shape
    m = node_mask.float()
    L = laplacian_for_eigh(E, node_mask)
    ev, vec = eigh(L)
    ev = ev / m.sum(dim=1, keepdim=True)
    vec = vec * m[:, :, None] * m[:, None, :]
    ncc = (ev < 1e-5).sum(dim=-1)

    ext = int(ncc.max()) + num_eigenvalues - n
    evp = torch.cat([ev, 2 * torch.ones(B, ext)], dim=1) if ext > 0 else ev
    idx = torch.arange(num_eigenvalues)[None] + ncc[:, None]
    first_ev = torch.gather(evp, 1, idx)
    evalue_feat = torch.cat([ncc[:, None].to(ev.dtype), first_ev], dim=1)

    if pad_noise is None:
        pad_noise = torch.randn(B, n)                                          # :171
    v0 = torch.round(vec[:, :, 0], decimals=3) * m + pad_noise * (1 - m)       # :169-172
    mode = torch.mode(v0, dim=1).values                                        # :173
    not_lcc = ((v0 != mode[:, None]) & node_mask).float()                      # :174-175
    ext = int(ncc.max()) + num_eigenvectors - n
    vp = torch.cat([vec, torch.zeros(B, n, ext)], dim=2) if ext > 0 else vec
    idx = (torch.arange(num_eigenvectors)[None, None] + ncc[:, None, None]).expand(-1, n, -1)
    first_vec = torch.gather(vp, 2, idx) * m[:, :, None]
    return evalue_feat, torch.cat([not_lcc[..., None], first_vec], dim=-1)


def extra_features(X, E, node_mask, max_n_nodes, num_degree=10, max_degree=10, cycle_features=True,
                   eigenfeatures=True, num_eigenvectors=5, num_eigenvalues=9, pad_noise=None, eigh=torch.linalg.eigh):
    """ExtraFeatures.__call__ (:60-77) -> (X_extra, E_extra, y_extra)."""
    nfrac = node_mask.sum(dim=1, keepdim=True) / max_n_nodes
    x, y, e = adjacency_features(X, E, node_mask, num_degree, max_degree, cycle_features)
    if eigenfeatures:
        evl, evc = eigen_features(E, node_mask, num_eigenvectors, num_eigenvalues, eigh=eigh, pad_noise=pad_noise)
        return torch.cat([x, evc], dim=-1), e, torch.cat([nfrac, y, evl], dim=1)
    return x, e, torch.cat([nfrac, y], dim=1)
