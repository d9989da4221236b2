"""development: what would homogeneous wavefronts buy the lane-per-subgraph kernel?  Its time is the slowest lane's; lanes = the TINY
list in pair order.  Model cost of a subgraph c(n, m) in {n + m, (n + m)^2, m^2}: sum over wavefronts of the maximum, list order vs
sorted by cost."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = torch.as_tensor(wl["pi_pairs"]).cuda()
g.pd_pi_batch(pairs, 2)
n, m2 = g.sizes(len(pairs))
m = m2 // 2
tiny = (n > 0) & (n <= 16) & (m <= 24)
n, m = n[tiny].astype(np.float64), m[tiny].astype(np.float64)
print("TINY pairs", tiny.sum(), "mean n %.1f m %.1f" % (n.mean(), m.mean()))
for name, c in (("n+m", n + m), ("(n+m)^2", (n + m) ** 2), ("m^2", m ** 2), ("n*m", n * m)):
    def waves(x):
        pad = (-len(x)) % 64
        return np.concatenate([x, np.zeros(pad)]).reshape(-1, 64).max(1).sum()
    a, b = waves(c), waves(np.sort(c))
    print("cost %-8s  list order %.3g   sorted %.3g   ratio %.2f   (mean lane / max lane in list order: %.2f)" % (name, a, b, a / b, c.mean() * len(c) / 64 / a * 1.0))
