"""development: tier counts and vicinity-size histogram of the strong-scaling list (504 514 non-edges within hop distance)."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
ci = engine.ComplementIndex(wl["rowptr"], wl["col"], device=0)
near, ranks = engine.near_pairs(ci, wl["hop"])
near = near[torch.argsort(ranks)].contiguous()
out, st = g.pd_pi_batch(near, wl["hop"])
print(g.stats(), g.dc_stats())
n, m = g.sizes(len(near))
if n is not None:
    m = m // 2
    for lo, hi in ((0, 16), (16, 64), (64, 128), (128, 256), (256, 512), (512, 2048), (2048, 1 << 30)):
        k = (n > lo) & (n <= hi)
        kk = (m - n + 1)[k]
        print("n in (%d, %d]: %7d pairs, mean n %.0f, mean m %.0f  max m %d | Pos edges m-n+1: mean %.0f  median %.0f  p90 %.0f  max %d" % (lo, hi, k.sum(), n[k].mean() if k.any() else 0, m[k].mean() if k.any() else 0, m[k].max() if k.any() else 0, kk.mean() if k.any() else 0, np.median(kk) if k.any() else 0, np.percentile(kk, 90) if k.any() else 0, kk.max() if k.any() else 0))
def joint(tag, n, m):
    big = (n > 128) & (n <= 512) & (m <= 1024)
    print("%s: %d pairs with 128 < n <= 512, m <= 1024" % (tag, big.sum()))
    for nc, mc in ((256, 512), (320, 512), (384, 512), (448, 512), (512, 512), (384, 640), (384, 1024)):
        print("   n <= %d and m <= %d: %6.1f %%" % (nc, mc, 100.0 * ((n <= nc) & (m <= mc) & big).sum() / max(big.sum(), 1)))
joint("strong list", n, m)
pairs = torch.as_tensor(wl["pi_pairs"]).cuda()
g.pd_pi_batch(pairs, wl["hop"])
print(g.stats())
n2, m2 = g.sizes(len(pairs)); m2 = m2 // 2
joint("headline batch", n2, m2)
