"""Phase times of eigen_features_tridiag_kernel (library built with EXTRA=-DSCGG_TRI_STAMP): 100 MHz wall-clock ticks."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_eigen_tridiag import *
from semantically_consistent_graph_generation_amd import _lib
gen = torch.Generator().manual_seed(3)
for (B, n) in [(4, 256), (4, 128), (4, 64)]:
    mask = torch.ones(B, n, dtype=torch.bool)
    L = laplacians([tree_adj(n, n, gen, extra=n // 8) for _ in range(B)], mask).cuda()
    mc = mask.cuda().view(torch.uint8)
    evl = torch.empty(B, 10, device="cuda"); evc = torch.empty(B, n, 6, device="cuda"); w = torch.empty(B, n, device="cuda")
    for _ in range(3):
        _lib.check(_lib.lib().scgg_eigen_features_tridiag(_lib.ptr(L), _lib.ptr(mc), B, n, KE, KV, _lib.ptr(evl), 10, 0, _lib.ptr(evc), 6, 0, _lib.ptr(w), None, _lib.stream_ptr()), "x")
    torch.cuda.synchronize()
    names = ["compact", "load", "tridiag", "bisect", "select", "invit", "gs", "backtr"]
    names = ["setup", "tridiag", "bisect", "select", "invit", "gs", "backtr", "outputs"]
    print(f"n={n}: " + ", ".join(f"{k} {v / 100:.1f} us" for k, v in zip(names, w[0, :8].cpu().tolist())))
    print("   step k=8, 10-ns ticks since its start: top done, symv done, after barrier B, combine done, after barrier C, update done:",
          [int(w[0, 8 + k].item()) for k in (1, 3, 4, 5, 6, 7)])
