"""Time scgg_train_wgrad on the four products of one backward chunk (B = 2, n = 256, vessel dims) for a few slab sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantically_consistent_graph_generation_amd import _lib
from semantically_consistent_graph_generation_amd.training import _wgrad_into
S = 2 * 256 * 256
L = _lib.lib()
g = torch.Generator().manual_seed(0)
prods = [(256, 64), (64, 128), (128, 64), (64, 128)]
Gs = [(torch.randn(S, M, generator=g).cuda(), torch.randn(S, N, generator=g).cuda()) for M, N in prods]
for slab in (256, 512, 1024, 2048):
    nsl = int(L.scgg_train_wgrad_slabs(S, slab))
    ld = sum(M * N + M for M, N in prods)
    P = torch.empty(nsl, ld, device="cuda")
    def run():
        c0 = 0
        for (G, X), (M, N) in zip(Gs, prods):
            _wgrad_into(G, X, P, 0, c0, slab)
            c0 += M * N + M
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    e0.record()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    host = (time.perf_counter() - t0) / 80
    e1.record()
    torch.cuda.synchronize()
    print(f"   host time per launch {host * 1e6:.1f} us")
    ms = e0.elapsed_time(e1) / 20
    byt = sum(S * 4 * (M * ((N + 63) // 64) + N * ((M + 63) // 64)) for M, N in prods)
    print(f"slab {slab}: {ms * 1e3:.1f} us for the four products ({byt / ms / 1e6:.0f} GB/s of operand reads), partial rows {nsl}")
