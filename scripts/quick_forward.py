"""Quick timing of one denoiser forward at the BASELINE config-2 shape (development aid, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from types import SimpleNamespace as _NS
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 32
torch.manual_seed(0)
m = GraphTransformer(5, _NS(X=10, E=25, y=43, charges=0), bench.DIMS["hidden_mlp_dims"], bench.DIMS["hidden_dims"],
                     _NS(X=1, E=14, y=0, charges=0)).eval().cuda()
g = torch.Generator().manual_seed(1234)
nn_ = torch.randint(12, 33, (B,), generator=g); nn_[0] = 32
mask = (torch.arange(n)[None, :] < torch.as_tensor(nn_)[:, None]).cuda()
X = torch.randn(B, n, 10, device="cuda"); E = torch.randn(B, n, n, 25, device="cuda"); E = 0.5 * (E + E.transpose(1, 2)).contiguous()
y = torch.randn(B, 43, device="cuda")
inp = model_input(X, E, y, mask)
for _ in range(3):
    out = m(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    out = m(inp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
pairs = int((nn_ ** 2).sum())
print(f"forward {dt*1e3:.3f} ms  valid pairs {pairs}  edge-layer flops {pairs*131072*5/1e9:.1f} GF -> {pairs*131072*5/dt/1e12:.1f} TF/s if all time were layers")
