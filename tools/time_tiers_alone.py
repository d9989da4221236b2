"""Each tier kernel of the PD/PI batch alone (the others not launched): standalone durations vs the overlapped batch."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
out = torch.empty((E, 25), dtype=torch.float64, device="cuda")
st = torch.empty(E, dtype=torch.uint8, device="cuda")
def run(mask, K=10):
    g.set_option("tier_mask", mask)
    for _ in range(3):
        g.pd_pi_batch(pairs, 2, out=out, status=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        g.pd_pi_batch(pairs, 2, out=out, status=st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
names = {0: "SMALL", 1: "MEDIUM", 2: "LARGE", 4: "MID", 5: "TINY", 6: "MEDHI"}
base = run(0)
print("no tier kernels (lead-in, extraction, scan, joins): %.3f ms" % base)
tot = 0.0
for t, nm in names.items():
    v = run(1 << t)
    tot += v - base
    print("only %-6s: %.3f ms  (+%.3f)" % (nm, v, v - base))
print("all tiers: %.3f ms; sum of the stand-alone increments %.3f ms" % (run(0xff), tot))
print("no LARGE: %.3f ms;  no TINY: %.3f;  no MEDIUM/MEDHI: %.3f;  no SMALL: %.3f;  TINY+SMALL only: %.3f;  MEDIUM+MEDHI+MID only: %.3f" % (
    run(0xff & ~4), run(0xff & ~32), run(0xff & ~(2 | 64 | 128)), run(0xff & ~1), run(33), run(2 | 64 | 128 | 16)))
