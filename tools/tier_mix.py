import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
rot = [torch.from_numpy(b).cuda() for b in bench.rotated_batches(W, 8, seed=4321)]
for name, b in [("fixed", pairs)] + [("rot%d" % i, r) for i, r in enumerate(rot[:4])]:
    g.pd_pi_batch(b, 2); torch.cuda.synchronize()
    s = g.stats(); n, m2 = g.sizes(len(b))
    print(name, {k: v for k, v in s.items() if k.startswith("tier") or k == "induced_entries"}, "sum n", int(n.sum()), "max m", int(m2.max()) // 2)
