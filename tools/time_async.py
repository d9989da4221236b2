"""Batches in flight on ONE handle: tlc_pd_pi_batch back to back vs tlc_pd_pi_batch_async + join (development aid).
TLC_CU_RESERVE=n in the environment reserves n CUs for the heavy chain (CU-masked streams)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
for _ in range(3):
    g.pd_pi_batch(pairs, 2, out=outs[0], status=sts[0])
torch.cuda.synchronize()
ref, rst = outs[0].clone(), sts[0].clone()
def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
def sync_loop():
    for k in range(K):
        g.pd_pi_batch(pairs, 2, out=outs[k % 3], status=sts[k % 3])
def async_loop():
    for k in range(K):
        g.pd_pi_batch(pairs, 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
res = {}
for name, fn in (("sync", sync_loop), ("async", async_loop), ("sync2", sync_loop), ("async2", async_loop)):
    res[name] = timed(fn)
ok = all(bool((o == ref).all()) for o in outs) and all(bool((s == rst).all()) for s in sts)
print("CU_RESERVE=%s  sync %.3f / %.3f ms  async %.3f / %.3f ms  (%.2f -> %.2f M images/s)  outputs equal: %s" % (
    os.environ.get("TLC_CU_RESERVE", "0"), res["sync"], res["sync2"], res["async"], res["async2"], E / res["sync2"] / 1e3, E / min(res["async"], res["async2"]) / 1e3, ok))
