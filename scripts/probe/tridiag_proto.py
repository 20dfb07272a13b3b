"""fp32 prototype of the n <= 256 eigen-feature kernel's algorithm (development aid, CPU only): Householder
tridiagonalisation, Sturm bisection, inverse iteration with pivoted LU on the tridiagonal, Gram-Schmidt inside clusters, back
transformation.  Checks residuals / eigenvalues against numpy's eigh in float64 on the committed Laplacian fixtures and on random
trees with many symmetric leaves (degenerate spectra)."""
import sys

import numpy as np

F = np.float32


def tridiag(A):
    """-> d, e, H: H[k] = (tau_k, v_k) with H_k = I - tau_k v_k v_k^T acting on rows k+1.."""
    A = A.astype(F).copy()
    n = A.shape[0]
    d = np.zeros(n, F)
    e = np.zeros(max(n - 1, 0), F)
    H = []
    amax = F(np.abs(A).max()) if n else F(0)
    tiny2 = F(1e-20) * amax * amax                      # a column below 1e-10 of the matrix is taken as zero
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        d[k] = A[k, k]
        sig = F(np.dot(x[1:], x[1:]))
        if sig <= tiny2:
            e[k] = x[0]
            H.append((F(0), None))
            continue
        nrm = F(np.sqrt(F(x[0] * x[0] + sig)))
        alpha = -nrm if x[0] >= 0 else nrm
        v = x.copy()
        v[0] = F(x[0] - alpha)
        t = F(1.0) / F(nrm * nrm - alpha * x[0])          # 2 / v.v
        A22 = A[k + 1:, k + 1:]
        p = (t * (A22 @ v)).astype(F)
        K = F(0.5) * t * F(np.dot(p, v))
        w = (p - K * v).astype(F)
        A22 -= (np.outer(v, w) + np.outer(w, v)).astype(F)
        e[k] = alpha
        H.append((t, v))
    if n >= 2:
        d[n - 2] = A[n - 2, n - 2]
        e[n - 2] = A[n - 1, n - 2]
    d[n - 1] = A[n - 1, n - 1]
    return d, e, H


def sturm_count(d, e2, x):
    """number of eigenvalues < x, vectorised over x"""
    n = len(d)
    cnt = np.zeros(x.shape, np.int32)
    q = (d[0] - x).astype(F)
    cnt += q < 0
    for i in range(1, n):
        q = np.where(np.abs(q) < PIVMIN, -PIVMIN, q).astype(F)
        q = (d[i] - x - e2[i - 1] / q).astype(F)
        cnt += q < 0
    return cnt


def bisect_all(d, e):
    n = len(d)
    e2 = (e * e).astype(F)
    r = np.zeros(n, F)
    r[:-1] += np.abs(e)
    r[1:] += np.abs(e)
    gl, gu = F((d - r).min()), F((d + r).max())
    span = F(gu - gl)
    gl, gu = F(gl - span * F(1e-6) - PIVMIN), F(gu + span * F(1e-6) + PIVMIN)
    lo = np.full(n, gl, F)
    hi = np.full(n, gu, F)
    j = np.arange(n)
    for _ in range(40):
        mid = ((lo + hi) * F(0.5)).astype(F)
        c = sturm_count(d, e2, mid)
        left = c > j                      # eigenvalue j is below mid
        hi = np.where(left, mid, hi)
        lo = np.where(left, lo, mid)
    return ((lo + hi) * F(0.5)).astype(F)


def lu_factor(d, e, lam, eps_t):
    n = len(d)
    dl = np.zeros(n, F); dd = (d - lam).astype(F); du = np.zeros(n, F); du2 = np.zeros(n, F); piv = np.zeros(n, bool)
    du[:-1] = e
    for i in range(n - 1):
        sub = e[i]
        if abs(dd[i]) >= abs(sub):
            if abs(dd[i]) < eps_t:
                dd[i] = eps_t
            f = F(sub / dd[i])
            dl[i] = f
            dd[i + 1] = F(dd[i + 1] - f * du[i])
            du2[i] = 0
        else:
            f = F(dd[i] / sub)
            dd[i] = sub
            dl[i] = f
            t = du[i]
            du[i] = dd[i + 1]
            dd[i + 1] = F(t - f * dd[i + 1])
            du2[i] = du[i + 1]
            du[i + 1] = F(-f * du[i + 1])
            piv[i] = True
    if abs(dd[n - 1]) < eps_t:
        dd[n - 1] = eps_t
    return dl, dd, du, du2, piv


def lu_solve(fac, b):
    dl, dd, du, du2, piv = fac
    n = len(dd)
    b = b.astype(F).copy()
    for i in range(n - 1):
        if piv[i]:
            t = b[i]; b[i] = b[i + 1]; b[i + 1] = F(t - dl[i] * b[i])
        else:
            b[i + 1] = F(b[i + 1] - dl[i] * b[i])
    b[n - 1] = F(b[n - 1] / dd[n - 1])
    if n > 1:
        b[n - 2] = F((b[n - 2] - du[n - 2] * b[n - 1]) / dd[n - 2])
    for i in range(n - 3, -1, -1):
        b[i] = F((b[i] - du[i] * b[i + 1] - du2[i] * b[i + 2]) / dd[i])
    return b


def eig_selected(L, sel_fn, iters=3):
    """-> (w all ascending, dict column -> eigenvector of L) for the columns sel_fn(w) asks for"""
    global PIVMIN
    n = L.shape[0]
    nrmT = F(np.abs(L).sum(1).max())
    PIVMIN = F(max(1e-30, 1e-12 * nrmT))
    d, e, H = tridiag(L)
    w = bisect_all(d, e)
    cols = sel_fn(w)
    eps_t = F(nrmT * 6e-8)
    rng = np.random.default_rng(1)
    Z, lam_used = {}, {}
    prev = []
    for c in cols:
        lam = w[c]
        # perturb eigenvalues that coincide with an earlier selected one
        for pc in prev:
            if abs(lam - lam_used[pc]) < 10 * eps_t:
                lam = F(lam_used[pc] + 10 * eps_t)
        lam_used[c] = lam
        fac = lu_factor(d, e, lam, eps_t)
        z = (rng.random(n) * 2 - 1).astype(F)
        for it in range(iters):
            z = lu_solve(fac, z)
            z = (z / F(np.sqrt(np.dot(z, z)))).astype(F)
        Z[c] = z
        prev.append(c)
    # Gram-Schmidt inside clusters, after the (independent) iterations: what the kernel does
    for a, c in enumerate(cols):
        z = Z[c]
        for rep in range(2):
            for pc in cols[:a]:
                if abs(w[pc] - w[c]) < F(1e-3) * nrmT:
                    z = (z - F(np.dot(z, Z[pc])) * Z[pc]).astype(F)
        Z[c] = (z / F(np.sqrt(np.dot(z, z)))).astype(F)
    # back transformation: x = H_0 H_1 ... H_{n-3} z
    out = {}
    for c, z in Z.items():
        x = z.copy()
        for k in range(n - 3, -1, -1):
            t, v = H[k]
            if t == 0:
                continue
            s = F(t * np.dot(v, x[k + 1:]))
            x[k + 1:] = (x[k + 1:] - s * v).astype(F)
        out[c] = x
    return w, out


def check(L, name):
    n = L.shape[0]
    wd, Vd = np.linalg.eigh(L.astype(np.float64))

    def sel(w):
        ncc = int((w / n < 1e-5).sum())
        return [0] + [c for c in range(ncc, min(n, ncc + 5)) if c != 0]
    w, X = eig_selected(L, sel)
    err_w = np.abs(w - wd).max()
    res, orth = 0.0, 0.0
    keys = list(X)
    for c in keys:
        x = X[c].astype(np.float64)
        res = max(res, np.abs(L @ x - w[c] * x).max())
    G = np.array([[abs(np.dot(X[a].astype(np.float64), X[b].astype(np.float64))) - (a == b) for b in keys] for a in keys])
    print(f"{name}: n={n} |dw|max={err_w:.2e} residual={res:.2e} orth={np.abs(G).max():.2e}  ncc={int((w / n < 1e-5).sum())} ref={(wd / n < 1e-5).sum()}")
    return err_w, res, np.abs(G).max()


def random_tree(n, seed, leaves=0.5, extra=0):
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for i in range(1, n):
        p = rng.integers(max(0, i - 6), i) if rng.random() > leaves else rng.integers(0, max(1, i // 8))
        A[i, p] = A[p, i] = 1
    for _ in range(extra):
        i, j = rng.integers(0, n, 2)
        if i != j:
            A[i, j] = A[j, i] = 1
    return np.diag(A.sum(1)) - A


if __name__ == "__main__":
    worst = [0, 0, 0]
    for n, seed in [(40, 1), (64, 2), (100, 3), (200, 4), (256, 5), (256, 6), (33, 7)]:
        r = check(random_tree(n, seed, extra=n // 20), f"tree{n}/{seed}")
        worst = [max(a, b) for a, b in zip(worst, r)]
    # disconnected: two components + isolated nodes
    L = np.zeros((120, 120))
    L[:50, :50] = random_tree(50, 11)
    L[50:110, 50:110] = random_tree(60, 12)
    perm = np.random.default_rng(3).permutation(120)
    L = L[perm][:, perm]
    r = check(L, "disconnected")
    # star (n-2 fold degenerate eigenvalue 1) and path
    n = 128
    A = np.zeros((n, n)); A[0, 1:] = 1; A[1:, 0] = 1
    check(np.diag(A.sum(1)) - A, "star")
    A = np.zeros((n, n))
    for i in range(n - 1):
        A[i, i + 1] = A[i + 1, i] = 1
    check(np.diag(A.sum(1)) - A, "path")
    for name in sys.argv[1:]:
        g = np.load(name)
        print(name, list(g.keys())[:12])


def sparse_check():
    rng = np.random.default_rng(4)
    for trial in range(30):
        s = int(rng.integers(10, 41))
        A = np.zeros((s, s))
        for _ in range(int(rng.integers(0, 2 * s))):
            i, j = rng.integers(0, s, 2)
            if i != j:
                A[i, j] = A[j, i] = 1
        L = np.diag(A.sum(1)) - A
        check(L, f"sparse{trial}")
