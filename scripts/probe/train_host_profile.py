"""cProfile of the host side of the training step (where do the ~15 us per launch go?)."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import bench_train as bt
from types import SimpleNamespace
a = SimpleNamespace(n=256, batch=4)
dev = torch.device("cuda", 0)
from semantically_consistent_graph_generation_amd import noise_model as NM
from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
from semantically_consistent_graph_generation_amd.noise_model import Rng
from semantically_consistent_graph_generation_amd.training import training_step
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
C = 5
torch.manual_seed(0)
cfg = SimpleNamespace(model=SimpleNamespace(nu=bt.NU, transition="absorbing_edges", diffusion_steps=500, top_k=4, num_degree=10, max_degree=10,
                                            cycle_features=True, num_eigenvectors=5, num_eigenvalues=9, eigenfeatures=True))
ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=a.n))
in_dims = ef.update_input_dims(bt.ns(1, C, 1))
model = GraphTransformer(5, in_dims, bt.DIMS["hidden_mlp_dims"], bt.DIMS["hidden_dims"], bt.ns(1, C, 0)).to(dev).train()
nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
dense, labd = bt.synthetic_batch(a, 0, C, dev)
opt = torch.optim.AdamW(model.parameters(), lr=2e-4, amsgrad=True, weight_decay=1e-12)
phase = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
import time
def step(it):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = training_step(model, nm, ef, dense, labd, rng=Rng(seed=11, step=it), dropout_seed=1000 + it)[0]
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    phase["fwd"] += t1 - t0; phase["bwd"] += t2 - t1; phase["opt"] += t3 - t2
for it in range(3):
    step(it)
torch.cuda.synchronize()
for k in phase: phase[k] = 0.0
N = 10
for it in range(N):          # host time per phase without the profiler (the device may lag: issue time only)
    step(3 + it)
torch.cuda.synchronize()
print({k: round(v / N * 1e3, 2) for k, v in phase.items()}, "ms per step (host issue, by phase)")
pr = cProfile.Profile()
pr.enable()
for it in range(N):
    step(20 + it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
