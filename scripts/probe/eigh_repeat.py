"""Probe: is torch.linalg.eigh on the device repeatable call to call, and where do two ExtraFeatures calls differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from types import SimpleNamespace
from oracle import fixtures as fx
from semantically_consistent_graph_generation_amd.extra_features import ExtraFeatures

c = SimpleNamespace(model=SimpleNamespace(num_degree=10, max_degree=10, cycle_features=True, num_eigenvectors=5, num_eigenvalues=9, eigenfeatures=True))
ef = ExtraFeatures(c, SimpleNamespace(max_n_nodes=40))
m = fx.ragged_mask([40, 23, 36])
for dens in (0.0, 0.01, 0.03, 0.1):
    lab = fx.random_labels(m, 5, dens, 3)
    E = fx.one_hot_edges(lab, m, 5).cuda()
    z = SimpleNamespace(X=m[..., None].float().cuda(), E=E, y=torch.zeros(3, 0, device="cuda"), node_mask=m.cuda())
    a = ef(z); b = ef(z)
    torch.cuda.synchronize()
    print("dens", dens, "X diff", float((a.X - b.X).abs().max()), "y diff", float((a.y - b.y).abs().max()), "E diff", float((a.E - b.E).abs().max()))
    from semantically_consistent_graph_generation_amd.noise_model import NoiseModel
    labels = NoiseModel._labels_of(z, 5)
    L = ef.laplacian(labels, z.node_mask)
    w1, V1 = torch.linalg.eigh(L); w2, V2 = torch.linalg.eigh(L.clone())
    print("   eigh repeat: w", float((w1 - w2).abs().max()), "V", float((V1 - V2).abs().max()))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        w3, V3 = torch.linalg.eigh(L)
    torch.cuda.synchronize()
    print("   eigh other stream: w", float((w1 - w3).abs().max()), "V", float((V1 - V3).abs().max()))
