import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_eigen_tridiag import *
from semantically_consistent_graph_generation_amd import _lib
gen = torch.Generator().manual_seed(9)
n, B = 32, 16
sizes = torch.randint(2, n + 1, (B,), generator=gen); sizes[0] = n
mask = fx.ragged_mask(sizes, n)
adjs = [tree_adj(int(s), n, gen, extra=2) for s in sizes]
L = laplacians(adjs, mask).cuda()
ef = make_ef(n)
evl_j, evc_j, wj, Vj = ef.eigen_features(L, mask.cuda(), want_raw=True)
evl_t = torch.empty_like(evl_j); evc_t = torch.empty_like(evc_j)
w = torch.empty(B, n, device="cuda"); vs = torch.empty(B, n, 6, device="cuda")
_lib.check(_lib.lib().scgg_eigen_features_tridiag(_lib.ptr(L), _lib.ptr(mask.cuda().view(torch.uint8)), B, n, KE, KV, _lib.ptr(evl_t), evl_t.shape[-1], 0,
    _lib.ptr(evc_t), evc_t.shape[-1], 0, _lib.ptr(w), _lib.ptr(vs), _lib.stream_ptr()), "x")
torch.cuda.synchronize()
for b in range(B):
    d = (evc_t[b, :, 0] != evc_j[b, :, 0]).sum().item()
    if d:
        print("graph", b, "nv", int(sizes[b]), "mismatch", d)
        print(" tridiag flags", evc_t[b, :, 0].cpu().numpy())
        print(" jacobi flags ", evc_j[b, :, 0].cpu().numpy())
        print(" tri v0", vs[b, :, 0].cpu().numpy())
        print(" jac v0", Vj[b, :, 0].cpu().numpy())
        print(" w tri", w[b, :4].cpu().numpy(), "w jac", wj[b, :4].cpu().numpy())

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3

gen = torch.Generator().manual_seed(3)
for (B, n, lo_) in [(512, 32, 12), (64, 64, 30), (16, 128, 100), (4, 256, 256), (8, 256, 200), (32, 256, 200)]:
    sizes = torch.randint(lo_, n + 1, (B,), generator=gen)
    mask = fx.ragged_mask(sizes, n)
    L = laplacians([tree_adj(int(s), n, gen, extra=int(s) // 8) for s in sizes], mask).cuda()
    mc = mask.cuda().view(torch.uint8)
    evl = torch.empty(B, 10, device="cuda"); evc = torch.empty(B, n, 6, device="cuda")
    def tri():
        _lib.check(_lib.lib().scgg_eigen_features_tridiag(_lib.ptr(L), _lib.ptr(mc), B, n, KE, KV, _lib.ptr(evl), 10, 0, _lib.ptr(evc), 6, 0, None, None, _lib.stream_ptr()), "x")
    def jac():
        _lib.check(_lib.lib().scgg_eigen_features(_lib.ptr(L), _lib.ptr(mc), B, n, KE, KV, _lib.ptr(evl), 10, 0, _lib.ptr(evc), 6, 0, None, None, _lib.stream_ptr()), "x")
    line = f"B={B} n={n}: tridiag {timeit(tri):.0f} us"
    if n <= 32: line += f", jacobi {timeit(jac):.0f} us"
    else: line += f", rocsolver eigh {timeit(lambda: torch.linalg.eigh(L), 5):.0f} us"
    print(line)
