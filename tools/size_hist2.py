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
m = m2 // 2
tiers = engine.tier_of(nn, m2)
print("n==0", (nn == 0).sum())
for nc, mc in ((2,1),(3,3),(4,6),(8,16),(12,16),(16,16),(16,24),(16,32),(24,32),(32,32),(32,48),(32,64),(48,96),(64,128)):
    sel = (nn <= nc) & (m <= mc) & (nn > 0)
    print("n<=%d m<=%d : %.1f%% of pairs, sum(m) share %.1f%%, trees %.1f%%" % (nc, mc, 100*sel.mean(), 100*m[sel].sum()/m.sum(), 100*((m==nn-1)&sel).sum()/max(sel.sum(),1)))
print("Pos count pct (all):", np.percentile((m - nn + 1)[nn>0], [25,50,75,90,99,100]))
for name in ("pd_tier_small","pd_tier_mid","pd_tier_medium","pd_tier_large"):
    sel = tiers == name
    print(name, sel.sum(), "sum m", m[sel].sum(), "sum Pos", (m-nn+1)[sel].sum(), "max Pos", (m-nn+1)[sel].max())
deg = np.diff(wl["rowptr"])
du, dv = deg[pairs[:,0]], deg[pairs[:,1]]
print("deg sum pct:", np.percentile(du+dv, [10,50,90,99,100]))
# ball sizes: sum over neighbours deg
ub = 1 + np.bincount(np.repeat(np.arange(len(deg)), deg), weights=(1+deg[wl["col"][:wl["rowptr"][-1]]]).astype(float), minlength=len(deg))
print("min ub pct:", np.percentile(np.minimum(ub[pairs[:,0]], ub[pairs[:,1]]), [10,50,90,99,100]))
print("sum ub (both balls) pct:", np.percentile(ub[pairs[:,0]] + ub[pairs[:,1]], [10,50,90,99,100]), "mean", (ub[pairs[:,0]] + ub[pairs[:,1]]).mean())
