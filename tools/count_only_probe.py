"""The extraction with and without its arena writes: tlc_vicinity_sizes (count only, dir == null) against the batch's own extraction on the
same 37 676 pairs, with the tier kernels masked out: development aid."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
def timed(fn, reps=15):
    fn(); torch.cuda.synchronize()
    v = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1))
    return float(np.median(v)), float(np.min(v))
print("sizes only (count pass, no arena writes): median %.3f ms  min %.3f ms" % timed(lambda: g.vicinity_sizes(pairs, 2)))
g.set_option("tier_mask", 0)
print("batch with the tier kernels masked out (classify + early + extraction + scan): median %.3f ms  min %.3f ms" % timed(lambda: g.pd_pi_batch(pairs, 2)))
g.set_timing(True); g.pd_pi_batch(pairs, 2); torch.cuda.synchronize()
print({k: round(v, 3) for k, v in g.timings().items() if v >= 0})
