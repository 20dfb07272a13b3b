"""Cost of the host-side projector in the loop at the bench workload (B=512 CoW-shaped graphs, T steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from semantically_consistent_graph_generation_amd import projector as PJ

T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
dev = torch.device("cuda:0")
sampler, _ = bench.build_sampler(T, dev)
n_nodes = bench.node_counts(512, 0)
sampler.sample_batch(n_nodes, seed=1, n_max=bench.N_MAX, steps=2, return_state=True)
def py(cls):
    f = lambda z, d: cls(z, d, engine="python")
    f.project_labels = True
    return f


cases = [("none", None), ("cow_proj", PJ.CoWProjector), ("cow_proj_vanilla", PJ.CoWVanillaProjector)]
if "--python" in sys.argv:
    cases += [("cow_proj (python engine)", py(PJ.CoWProjector))]
for name, proj in cases:
    np.random.seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lab, _, _, _ = sampler.sample_batch(n_nodes, seed=7, n_max=bench.N_MAX, projector=proj, return_state=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3 / T:.2f} ms/step, {512 / dt * T / 500:.0f} graphs/s at T=500 rate, edges {int((lab > 0).sum()) // 2}", flush=True)


def chain_rng(cls, seed):
    rs = np.random.RandomState(seed)
    f = lambda z, d: cls(z, d, rng=rs)
    f.project_labels = True
    return f


for nb in (2, 3):
    reqs = [dict(n_nodes=n_nodes, seed=7 + k, n_max=bench.N_MAX, graph_id_base=512 * k, projector=chain_rng(PJ.CoWProjector, k),
                 return_state=True) for k in range(nb)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.sample_batches(reqs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"cow_proj, {nb} batches pipelined: {dt * 1e3 / T / nb:.2f} ms/step/batch, {nb * 512 / dt * T / 500:.0f} graphs/s at T=500 rate", flush=True)
