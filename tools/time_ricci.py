"""Timing of the curvature step on the PubMed-shaped graph (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth

n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e)
engine.ollivier_ricci_sinkhorn(rowptr, col, e)
for name, sub in (("all edges", e), ("edges with a small support", None), ("hub edges", None)):
    deg = np.diff(rowptr)
    prod = (deg[e[:, 0]] + 1) * (deg[e[:, 1]] + 1)
    if name.startswith("edges with"):
        sub = e[(prod <= 8192) & (deg[e[:, 0]] + deg[e[:, 1]] + 2 <= 256)]
    elif name.startswith("hub"):
        sub = e[(prod > 8192) | (deg[e[:, 0]] + deg[e[:, 1]] + 2 > 256)]
    if len(sub) == 0:
        continue
    torch.cuda.synchronize()
    t0 = time.time()
    kap, it = engine.ollivier_ricci_sinkhorn(rowptr, col, sub, want_iters=True)
    dt = time.time() - t0
    print("%-28s %7d edges  %8.1f ms  %9.0f edges/s  iterations mean %.1f max %d" % (name, len(sub), dt * 1e3, len(sub) / dt, it.mean(), it.max()))
