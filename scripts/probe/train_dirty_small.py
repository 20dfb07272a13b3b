"""Does the training step at the captured-step test's shape (B=3, n=40, tiny widths) depend on what freed memory holds?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from semantically_consistent_graph_generation_amd import noise_model as NM
from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
from semantically_consistent_graph_generation_amd.noise_model import Rng
from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
from semantically_consistent_graph_generation_amd.training import training_step
from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
dev, C, B, n = torch.device("cuda"), 5, 3, 40
ns = lambda X, E, y: SimpleNamespace(X=X, E=E, y=y, charges=0)
cfg = SimpleNamespace(model=SimpleNamespace(nu=dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1), transition="absorbing_edges",
                                            diffusion_steps=500, top_k=4, num_degree=10, max_degree=10, cycle_features=True,
                                            num_eigenvectors=5, num_eigenvalues=9, eigenfeatures=True))
ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=n))
in_dims = ef.update_input_dims(ns(1, C, 1))
torch.manual_seed(0)
dims = dict(hidden_mlp_dims={"X": 64, "E": 32, "y": 32},
            hidden_dims={"dx": 64, "de": 32, "dy": 32, "n_head": 4, "dim_ffX": 64, "dim_ffE": 64, "dim_ffy": 64})
model = GraphTransformer(2, in_dims, dims["hidden_mlp_dims"], dims["hidden_dims"], ns(1, C, 0)).to(dev).train()
nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
g = torch.Generator().manual_seed(4)
nn_ = torch.tensor([40, 31, 18])
m = torch.arange(n)[None] < nn_[:, None]
lab = torch.zeros(B, n, n, dtype=torch.long)
for b in range(B):
    nb = int(nn_[b]); idx = torch.arange(1, nb)
    par = (torch.rand(nb - 1, generator=g) * idx).long(); cls = torch.randint(1, C, (nb - 1,), generator=g)
    lab[b, idx, par] = cls; lab[b, par, idx] = cls
live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(n, dtype=torch.bool)[None]
E0 = torch.nn.functional.one_hot(lab, C).float() * live[..., None]
dense = PlaceHolder(X=m[..., None].float().to(dev), charges=torch.zeros(B, n, 0, device=dev), E=E0.to(dev),
                    y=torch.zeros(B, 0, device=dev), node_mask=m.to(dev), pos=None)
labd = lab.to(dev)
t_int = torch.tensor([[100], [250], [400]], device=dev)
u0 = torch.rand(B, n, n, generator=torch.Generator().manual_seed(9)).to(dev)

def dirty(byte):
    xs = [torch.full((sz,), byte, dtype=torch.uint8, device="cuda") for _ in range(40) for sz in (300, 5000, 70000, 1 << 20, 1 << 22, 1 << 24)]
    torch.cuda.synchronize(); del xs

def step(mode_env):
    for k, v in mode_env.items(): os.environ[k] = v
    model.zero_grad(set_to_none=True)
    loss, pred, z = training_step(model, nm, ef, dense, labd, t_int=t_int.clone(), rng=Rng(u0=u0.clone()), dropout_seed=78)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, pred.E.detach().clone()

for env in ({"SCGG_TRAIN_Y": "fused"}, {"SCGG_TRAIN_Y": "ops"}, {"SCGG_TRAIN_MODE": "eager"}):
    ref = step(env)
    print(env, "loss", ref[0])
    for byte in (0xFF, 0x7F, 0x01, 0x3F):
        dirty(byte)
        cur = step(env)
        bad = [k for k in ref[1] if not torch.equal(ref[1][k], cur[1][k])]
        print(f"  after 0x{byte:02X}: loss {cur[0]} same={cur[0] == ref[0]} logits_same={torch.equal(ref[2], cur[2])} differing grads: {bad[:6]}")
    for k in env: os.environ.pop(k)
