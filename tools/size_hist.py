"""Node / edge counts of the bench batch's vicinities: how the SMALL tier's population (17..64 nodes) is distributed."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
g.pd_pi_batch(pairs, 2)
n, m2 = g.sizes(E)
m = m2 // 2
print("tiers", g.tier_counts())
tiny = (n <= 16) & (m <= 24)
print("TINY (n<=16, m<=24): %d" % tiny.sum())
rest = ~tiny & (n <= 64) & (m <= 128)
print("SMALL rest: %d; n mean %.1f m mean %.1f" % (rest.sum(), n[rest].mean(), m[rest].mean()))
for nc, mc in ((20, 32), (24, 40), (24, 48), (32, 48), (32, 64), (40, 64), (48, 96)):
    s = rest & (n <= nc) & (m <= mc)
    print("  n<=%d m<=%d: %5d (%.1f %% of the SMALL tier's)  mean n %.1f m %.1f" % (nc, mc, s.sum(), 100.0 * s.sum() / rest.sum(), n[s].mean() if s.sum() else 0, m[s].mean() if s.sum() else 0))
print("m - n + 1 (independent cycles) among SMALL rest: mean %.1f p50 %d p90 %d" % ((m - n + 1)[rest].mean(), np.median((m - n + 1)[rest]), np.percentile((m - n + 1)[rest], 90)))
big = (n > 512) | (m > 1024)
print("LARGE / HUGE-sized (n > 512 or m > 1024): %d" % big.sum())
for nc, mc in ((640, 1280), (768, 1536), (1024, 2048), (1536, 3072), (2048, 4096)):
    s = big & (n <= nc) & (m <= mc)
    print("  n<=%d m<=%d: %3d   Pos edges (m - n + 1) median %d max %d" % (nc, mc, s.sum(), np.median((m - n + 1)[s]) if s.sum() else 0, (m - n + 1)[s].max() if s.sum() else 0))
print("  sizes:", sorted(zip(n[big].tolist(), m[big].tolist()))[-12:])
