"""Quick timing of one denoiser forward at the BASELINE config-2 shape (development aid, not the bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 32
torch.manual_seed(0)
m = GraphTransformer(5, fx.dims_ns(10, 25, 43), fx.CONSTRUCT_DIMS["hidden_mlp_dims"], fx.CONSTRUCT_DIMS["hidden_dims"],
                     fx.dims_ns(1, 14, 0)).eval().cuda()
g = torch.Generator().manual_seed(1234)
nn_ = torch.randint(12, 33, (B,), generator=g); nn_[0] = 32
mask = fx.ragged_mask(nn_, n).cuda()
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
