"""Probe: how long does the host need to ENQUEUE one reverse step, against the GPU's time to run it?"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch, bench
dev = torch.device("cuda:0")
sampler, model = bench.build_sampler(500, dev)
nn_ = bench.node_counts(512, 0)
sampler.sample_batch(nn_, seed=1, n_max=32, steps=5, return_state=True)
torch.cuda.synchronize()
gen = sampler._chain(nn_, seed=1, n_max=32, steps=200, return_state=True)
t0 = time.perf_counter()
try:
    while True:
        next(gen)
except StopIteration as e:
    pass
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step; device finished %.3f ms/step after start (%.1f ms of queued work left when the host was done)" % (
    (t1 - t0) * 5, (t2 - t0) * 5, (t2 - t1) * 1e3))
