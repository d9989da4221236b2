"""development: the strong-scaling list (non-edges within hop distance, 504 514 pairs on the PubMed-shaped graph) in one call,
a few times (the library cuts a list above 2^20 pairs into pipelined chunks; TLC_CHUNK_PAIRS_TEST in the environment cuts smaller)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
ci = engine.ComplementIndex(wl["rowptr"], wl["col"], device=0)
near, ranks = engine.near_pairs(ci, wl["hop"])
near = near[torch.argsort(ranks)].contiguous()
E = len(near)
out = torch.empty((E, 25), dtype=torch.float64, device="cuda"); st = torch.empty(E, dtype=torch.uint8, device="cuda")
ref = None
# python tools/time_strong_list.py [option valueA valueB]: the list timed under the two values of a handle option in turn
opt = sys.argv[1] if len(sys.argv) > 3 else None
seq = [int(sys.argv[2]), int(sys.argv[3])] * 2 if opt else [0, 0]
for cp in seq:
    if opt: g.set_option(opt, cp)
    ts = []
    for _ in range(5 if cp >= 0 else 12):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.pd_pi_batch(near, wl["hop"], out=out, status=st); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if ref is None: ref = (out.clone(), st.clone())
    same = bool(torch.equal(out, ref[0]) and torch.equal(st, ref[1]))
    print((opt or "run") + "=%8d  %d pairs: median %.2f ms (%.2f M images/s)  runs %s  rows equal: %s" % (cp, E, float(np.median(ts[1:])), E / float(np.median(ts[1:])) / 1e3, " ".join("%.1f" % t for t in ts), same))
