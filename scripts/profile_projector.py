"""Host-side timing of the reverse loop with the CoW projector at the bench workload: wall time inside each C-ABI
call and inside the two synchronising reads."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from semantically_consistent_graph_generation_amd import projector as PJ, _lib

T = 100
sampler, _ = bench.build_sampler(T, torch.device("cuda:0"))
n_nodes = bench.node_counts(512, 0)
sampler.sample_batch(n_nodes, seed=1, n_max=bench.N_MAX, steps=2, return_state=True)
acc = collections.defaultdict(float)
L = _lib.lib()


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t0
        return r
    return w


for nm in ("scgg_new_edges", "scgg_apply_edge_edits", "scgg_projector_begin", "scgg_projector_finish", "scgg_denoiser_forward",
           "scgg_reverse_step_edges", "scgg_adjacency_features", "scgg_eigen_features", "scgg_laplacian"):
    setattr(L, nm, timed(nm, getattr(L, nm)))
torch.Tensor.tolist = timed("tolist", torch.Tensor.tolist)
torch.Tensor.cpu = timed("cpu", torch.Tensor.cpu)
for nm in ("__setitem__", "__getitem__", "copy_", "to", "record_stream", "numpy"):
    setattr(torch.Tensor, nm, timed("Tensor." + nm, getattr(torch.Tensor, nm)))
_lib.stream_ptr = timed("stream_ptr", _lib.stream_ptr)
PJ._Projector._run = timed("_run (incl. begin/finish)", PJ._Projector._run)
PJ._Projector._project_labels_device = timed("project_labels", PJ._Projector._project_labels_device)
np.random.seed(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
sampler.sample_batch(n_nodes, seed=7, n_max=bench.N_MAX, projector=PJ.CoWProjector, return_state=True)
torch.cuda.synchronize()
print(f"total {(time.perf_counter() - t0) * 1e3 / T:.3f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:34s} {v * 1e3 / T:.3f} ms/step")
