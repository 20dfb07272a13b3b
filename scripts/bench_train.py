#!/usr/bin/env python3
"""Side measurement: one training step (apply_noise + extra features + train-mode forward with dropout + TrainLoss + backward
+ AdamW) at BASELINE.json configs[2]'s shape (ATM: B=4 graphs of up to 256 nodes, vessel dims, 5 layers, C=5) on one GPU.
Not the bench.py contract line.  Prints one JSON line."""
import argparse, json, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

DIMS = dict(hidden_mlp_dims={"X": 128, "E": 128, "y": 128},
            hidden_dims={"dx": 128, "de": 64, "dy": 64, "n_head": 4, "dim_ffX": 128, "dim_ffE": 128, "dim_ffy": 128})
NU = dict(x=1, c=1, e=1, y=1, e_small=1, e_mid=1, e_large=1)


def ns(X, E, y, charges=0):
    return SimpleNamespace(X=X, E=E, y=y, charges=charges)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from semantically_consistent_graph_generation_amd import noise_model as NM
    from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures
    from semantically_consistent_graph_generation_amd.noise_model import Rng
    from semantically_consistent_graph_generation_amd.placeholder import PlaceHolder
    from semantically_consistent_graph_generation_amd.training import training_step
    from semantically_consistent_graph_generation_amd.transformer_model import GraphTransformer
    C, dev = 5, torch.device("cuda:0")
    cfg = SimpleNamespace(model=SimpleNamespace(nu=NU, transition="absorbing_edges", diffusion_steps=500, top_k=4, num_degree=10, max_degree=10,
                                                cycle_features=True, num_eigenvectors=5, num_eigenvalues=9, eigenfeatures=True))
    ef = ExtraFeatures(cfg, SimpleNamespace(max_n_nodes=a.n))
    in_dims = ef.update_input_dims(ns(1, C, 1))
    torch.manual_seed(0)
    model = GraphTransformer(5, in_dims, DIMS["hidden_mlp_dims"], DIMS["hidden_dims"], ns(1, C, 0)).to(dev).train()
    nm = NM.AbsorbingEdgesTransition(cfg, torch.ones(1), torch.ones(C) / C, torch.zeros(0), 0)
    g = torch.Generator().manual_seed(7)
    nn_ = torch.randint(3 * a.n // 4, a.n + 1, (a.batch,), generator=g)
    nn_[0] = a.n
    m = torch.arange(a.n)[None] < nn_[:, None]
    lab = torch.zeros(a.batch, a.n, a.n, dtype=torch.long)
    for b in range(a.batch):                                   # random trees (vessel-like)
        nb = int(nn_[b])
        idx = torch.arange(1, nb)
        par = (torch.rand(nb - 1, generator=g) * idx).long()
        cls = torch.randint(1, C, (nb - 1,), generator=g)
        lab[b, idx, par] = cls
        lab[b, par, idx] = cls
    live = (m[:, :, None] & m[:, None, :]) & ~torch.eye(a.n, dtype=torch.bool)[None]
    E0 = torch.nn.functional.one_hot(lab, C).float() * live[..., None]
    dense = PlaceHolder(X=m[..., None].float().to(dev), charges=torch.zeros(a.batch, a.n, 0, device=dev), E=E0.to(dev),
                        y=torch.zeros(a.batch, 0, device=dev), node_mask=m.to(dev), pos=None)
    labd = lab.to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, amsgrad=True, weight_decay=1e-12)
    losses = []

    def step(it):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = training_step(model, nm, ef, dense, labd, rng=Rng(seed=11, step=it), dropout_seed=1000 + it)
        loss.backward()
        opt.step()
        return loss
    for it in range(3):
        step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        losses.append(step(3 + it))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(json.dumps({"workload": f"training step, vessel dims, B={a.batch}, n<={a.n}, 5 layers, C=5, extra features on, dropout 0.1 (layers)",
                      "ms_per_step": ms, "graphs_per_s": a.batch / ms * 1e3, "loss_first": float(losses[0]), "loss_last": float(losses[-1]),
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
