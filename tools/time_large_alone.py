"""Duration of the LARGE tier kernel when its pairs are the whole batch (no co-running tiers) -- development aid."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = wl["pi_pairs"]
pd = torch.as_tensor(pairs).cuda()
g.pd_pi_batch(pd, wl["hop"])
nn, m2 = g.sizes(len(pairs))
tiers = engine.tier_of(nn, m2)
g.set_timing(True)
for name in ("pd_tier_large", "pd_tier_medium", "pd_tier_small"):
    sel = torch.as_tensor(pairs[tiers == name]).cuda()
    for _ in range(3):
        g.pd_pi_batch(sel, wl["hop"])
    print(name, "alone:", len(sel), "pairs", {k: round(v, 3) for k, v in g.timings().items() if v >= 0})
    order = np.argsort(-m2[tiers == name])
    one = sel[torch.as_tensor(order[:1].copy()).cuda()]
    for _ in range(3):
        g.pd_pi_batch(one, wl["hop"])
    print(name, "heaviest pair alone:", {k: round(v, 3) for k, v in g.timings().items() if v >= 0})
for _ in range(3):
    g.pd_pi_batch(pd, wl["hop"])
print("full batch:", {k: round(v, 3) for k, v in g.timings().items() if v >= 0})
