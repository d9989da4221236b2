"""Vicinity sizes of the bench batch per tier, with the LDS a right-sized layout would need -- development aid."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = wl["pi_pairs"]
g.pd_pi_batch(torch.as_tensor(pairs).cuda(), wl["hop"])
nn, m2 = g.sizes(len(pairs))
tiers = engine.tier_of(nn, m2)
m = m2 // 2
pos = m - nn + 1
for name in sorted(set(tiers.tolist())):
    sel = (tiers == name) & (nn > 0)
    if not sel.any():
        continue
    print("%-22s %6d  n pct 10/50/90/99/max %s  m %s  pos %s" % (name, sel.sum(), np.percentile(nn[sel], [10, 50, 90, 99, 100]).astype(int),
          np.percentile(m[sel], [10, 50, 90, 99, 100]).astype(int), np.percentile(pos[sel], [10, 50, 90, 99, 100]).astype(int)))
for name, cuts in (("pd_tier_large", ((768, 1024), (1024, 1536), (1024, 2048), (1536, 3072))), ("pd_tier_medium", ((192, 384), (256, 512), (384, 768))),
                   ("pd_tier_medium_rest", ((192, 384), (256, 512), (384, 768))), ("pd_tier_small", ((32, 48), (32, 64), (48, 96))),
                   ("pd_tier_mid", ((96, 160), (96, 192), (128, 192)))):
    sel = (tiers == name) & (nn > 0)
    if not sel.any():
        continue
    for nc, mc in cuts:
        f = (nn[sel] <= nc) & (m[sel] <= mc)
        print("   %-20s n<=%d & m<=%d: %.1f%%" % (name, nc, mc, 100 * f.mean()))
