"""One image batch alone with every kernel's event pair on (the library's own events), for the two values of a handle option --
where does a single batch spend its time?  python tools/time_batch_kernels.py <option> <a> <b>"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
out = torch.empty((E, 25), dtype=torch.float64, device="cuda"); st = torch.empty(E, dtype=torch.uint8, device="cuda")
g.set_timing(True)
for v in (va, vb, va, vb):
    g.set_option(opt, v)
    acc = {}
    lat = []
    for rep in range(9):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.pd_pi_batch(pairs, 2, out=out, status=st); e1.record(); torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
        if rep >= 2:
            for k, x in g.timings().items():
                acc.setdefault(k, []).append(x)
    print("%s=%d latency %.3f ms | " % (opt, v, float(np.median(lat))) + "  ".join("%s %.3f" % (k, float(np.median(x))) for k, x in acc.items()))
