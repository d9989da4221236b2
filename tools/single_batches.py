"""A few stream-ordered single batches of the bench workload with a synchronisation after each (for rocprofv3 --kernel-trace +
tools/timeline.py: what ends one batch alone).  python tools/single_batches.py [option value ...]"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
for k in range(1, len(sys.argv) - 1, 2):
    g.set_option(sys.argv[k], int(sys.argv[k + 1]))
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
out = torch.empty((E, 25), dtype=torch.float64, device="cuda"); st = torch.empty(E, dtype=torch.uint8, device="cuda")
lat = []
for rep in range(8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.pd_pi_batch(pairs, 2, out=out, status=st); e1.record(); torch.cuda.synchronize()
    lat.append(e0.elapsed_time(e1))
print("latency ms:", " ".join("%.3f" % x for x in lat))
