import sys, torch
sys.path.insert(0, ".")
from semantically_consistent_graph_generation_amd.training import _sum_rows
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3
for A, R, C in [(2, 256, 65536), (2, 65536, 64)]:
    t = torch.randn(A, R, C, device="cuda")
    print((A, R, C), "MB", t.numel() * 4 / 1e6, "ours us", round(timeit(lambda: _sum_rows(t))), "torch us", round(timeit(lambda: t.sum(1))))
