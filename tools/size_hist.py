"""Vicinity size distribution of the bench batch per tier -- development aid."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from tlc_gnn_amd import engine
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = wl["pi_pairs"]
g.pd_pi_batch(torch.as_tensor(pairs).cuda(), wl["hop"])
nn, m2 = g.sizes(len(pairs))
tiers = engine.tier_of(nn, m2)
for name in ("pd_tier_small", "pd_tier_medium", "pd_tier_large"):
    sel = tiers == name
    n, m = nn[sel], m2[sel] // 2
    print(name, sel.sum(), "n pct 10/50/90/99/max:", np.percentile(n, [10, 50, 90, 99, 100]).astype(int), " m pct:", np.percentile(m, [10, 50, 90, 99, 100]).astype(int))
    if name == "pd_tier_medium":
        for nc, mc in ((128, 256), (192, 384), (256, 512), (384, 768)):
            print("   n<=%d & m<=%d: %.1f%%  (sum m share %.1f%%)" % (nc, mc, 100 * ((n <= nc) & (m <= mc)).mean(), 100 * m[(n <= nc) & (m <= mc)].sum() / m.sum()))
m_all = m2 // 2
ok = nn > 0
print("trees (m == n-1): %.1f%% of all pairs; by tier:" % (100 * (m_all[ok] == nn[ok] - 1).mean()),
      {t: round(100 * float(((m_all == nn - 1) & (tiers == t)).sum()) / max(1, int((tiers == t).sum())), 1) for t in ("pd_tier_small", "pd_tier_medium", "pd_tier_large")})
print("m - n + 1 (number of Pos edges) percentiles over small tier:", np.percentile((m_all - nn + 1)[tiers == "pd_tier_small"], [10, 25, 50, 75, 90]))
