"""Reverse loop at the reference's own vessel configuration (configs/model/discrete_vessels.yaml: dx128 de64 dy64 h4,
ffX128 ffE128 ffy128, 5 layers, extra features off, absorbing_edges, top_k 4), CoW classes (C=14), B=512, n<=32."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import numpy as np
import torch
import bench
from semantically_consistent_graph_generation_amd import noise_model as NM, projector as PJ
from semantically_consistent_graph_generation_amd.sampling import DiscreteSampler
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = 14
dev = torch.device("cuda:0")
dims = dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
            hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
cfg = NS(model=NS(nu=bench.NU, transition="absorbing_edges", diffusion_steps=T, top_k=4))
torch.manual_seed(0)
model = GraphTransformer(5, NS(X=1, E=C, y=1, charges=0), dims["hidden_mlp_dims"], dims["hidden_dims"],
                         NS(X=1, E=C, y=0, charges=0)).eval().to(dev)
nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
sampler = DiscreteSampler(model, nm)
n_nodes = bench.node_counts(512, 0)
sampler.sample_batch(n_nodes, seed=1, n_max=32, steps=2, return_state=True)
for name, proj, nb in (("none", None, 1), ("cow_proj x2", PJ.CoWProjector, 2)):
    def fac(k):
        if proj is None:
            return None
        rs = np.random.RandomState(k)
        f = lambda z, d: proj(z, d, rng=rs)
        f.project_labels = True
        return f
    reqs = [dict(n_nodes=n_nodes, seed=7 + k, n_max=32, graph_id_base=512 * k, projector=fac(k), return_state=True) for k in range(nb)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sampler.sample_batches(reqs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3 / T / nb:.2f} ms/step/batch, {nb * 512 / dt * T / 500:.0f} graphs/s at T=500 rate, "
          f"edges {int((out[0][0] > 0).sum()) // 2}", flush=True)
