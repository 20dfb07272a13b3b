"""Probe: poison the engine workspace with NaN before every forward.  A row that is read before it is written (or whose
write is lost) then shows up as NaN in the output instead of as a small numerical difference."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer, model_input
dims, ind, C, n_nodes, L = fx.CONSTRUCT_DIMS, (10, 25, 43), 14, [32, 17, 25, 12, 30, 31, 8, 1] * 8, 5
m = fx.ragged_mask(n_nodes); B, n = m.shape
model = GraphTransformer(L, fx.dims_ns(*ind), dims["hidden_mlp_dims"], dims["hidden_dims"], fx.dims_ns(1, C, 0)).eval()
model.load_state_dict(fx.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 5))
model = model.cuda()
g = torch.Generator().manual_seed(1)
X = torch.randn(B, n, ind[0], generator=g).cuda(); E = torch.randn(B, n, n, ind[1], generator=g); E = (0.5 * (E + E.transpose(1, 2))).cuda()
y = torch.randn(B, ind[2], generator=g).cuda(); md = m.cuda()
inp = model_input(X, E, y, md)
ref = model(inp); refE, refX = ref.E.clone(), ref.X.clone()
torch.cuda.synchronize()
ws = list(model._ws.values())[0]
poison = os.environ.get("PROBE_POISON", "nan")
bad = nan = 0
for r in range(int(sys.argv[1])):
    if poison == "nan":
        ws.view(torch.float32)[: ws.numel() // 4].fill_(float("nan"))
    elif poison == "zero":
        ws.zero_()
    elif poison == "prev":      # the previous forward computes NaN everywhere: its values are what a stale read would return
        model(model_input(X * float("nan"), E * float("nan"), y * float("nan"), md))
    out = model(inp)
    dE = (out.E - refE).abs().max(); 
    isn = bool(torch.isnan(out.E).any() or torch.isnan(out.X).any())
    d = float(dE)
    if isn:
        nan += 1
        if nan <= 5:
            idx = torch.isnan(out.E).nonzero()
            print("rep", r, "NaN entries in E:", idx.shape[0], "first", idx[:3].cpu().tolist(), flush=True)
    elif d:
        bad += 1
        if bad <= 5: print("rep", r, "dE", d, flush=True)
print("poison", poison, ": NaN outputs", nan, " finite mismatches", bad, "of", sys.argv[1])
