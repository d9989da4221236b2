"""Latency of the vicinity kernels on the heaviest pairs alone (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e, k)
g = engine.DeviceGraph(rowptr, col, w)
rs = np.random.RandomState(7)
pairs = e[rs.permutation(len(e))[:37676]].astype(np.int32)
pd = torch.as_tensor(pairs).cuda()
g.pd_pi_batch(pd, 2)
nn, m2 = g.sizes(len(pairs))
order = np.argsort(-m2)
g.set_timing(True)
for cnt in (1, 8, 64, 512, 4096):
    sel = torch.as_tensor(pairs[order[:cnt]]).cuda()
    for _ in range(3):
        g.pd_pi_batch(sel, 2)
    t = g.timings()
    print(cnt, "pairs; n,m2 of heaviest:", nn[order[0]], m2[order[0]], {k: round(v, 3) for k, v in t.items() if v >= 0})
light = torch.as_tensor(pairs[order[4096:]]).cuda()
for _ in range(3):
    g.pd_pi_batch(light, 2)
print("without the 4096 heaviest:", {k: round(v, 3) for k, v in g.timings().items() if v >= 0})
