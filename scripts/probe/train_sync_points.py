"""Where does the training step make the host wait for the device?  Runs bench_train's step under torch's sync debug mode."""
import os, sys, warnings, traceback, collections
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
def step(it):
    opt.zero_grad(set_to_none=True)
    loss = training_step(model, nm, ef, dense, labd, rng=Rng(seed=11, step=it), dropout_seed=1000 + it)[0]
    loss.backward()
    opt.step()
for it in range(3):
    step(it)
torch.cuda.synchronize()
seen = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "semantically_consistent" in f.filename or "bench_train" in f.filename or "optim" in f.filename]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])
    seen[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(3)
torch.cuda.set_sync_debug_mode("default")
for k, v in seen.most_common():
    print(v, k)
