"""Timing of the pre-filtered sweep's two parts on the PubMed-shaped graph (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine, pi_cache
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
ci = engine.ComplementIndex(wl["rowptr"], wl["col"])
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    pairs, ranks = engine.near_pairs(ci, 2)
    torch.cuda.synchronize(); t1 = time.time()
    out, st = g.pd_pi_batch(pairs, 2)
    torch.cuda.synchronize(); t2 = time.time()
    print("near_pairs %.2f ms (%d pairs); pd_pi_batch %.2f ms (%.1f M PI/s)" % ((t1 - t0) * 1e3, len(pairs), (t2 - t1) * 1e3, len(pairs) / (t2 - t1) / 1e6), g.stats())
g.set_timing(True)
out, st = g.pd_pi_batch(pairs, 2)
torch.cuda.synchronize()
print({k: round(v, 3) for k, v in g.timings().items()})
