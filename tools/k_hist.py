import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
g.pd_pi_batch(pairs, 2); torch.cuda.synchronize()
n, m2 = g.sizes(len(pairs)); m = m2 // 2
K = m - n + 1
med = ((n > 128) | (m > 256)) & (n <= 512) & (m <= 1024)
print("MEDIUM-sized:", med.sum(), " K quantiles:", np.percentile(K[med], [50, 90, 99, 100]).tolist())
for thr in (96, 128, 160, 200, 240, 280, 320):
    print("K >= %d: %d" % (thr, (med & (K >= thr)).sum()))
print("ten largest K:", np.sort(K[med])[-10:].tolist(), " their m:", m[med][np.argsort(K[med])[-10:]].tolist())
print(g.stats())
