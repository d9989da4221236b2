"""A/B of one handle option inside ONE process (boxes differ by several per cent): pipelined image batches, fixed and rotating,
alternating the option's two values.  python tools/ab_option.py <option> <value A> <value B> [K]"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 40
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
rot = [torch.from_numpy(b).cuda() for b in bench.rotated_batches(W, 8, seed=4321)]
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
def region(batches):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(batches[k % len(batches)], 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
def latency():
    v = []
    for _ in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.pd_pi_batch(pairs, 2, out=outs[0], status=sts[0]); e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1))
    return float(np.median(v))
for v in (va, vb):
    g.set_option(opt, v); region([pairs]); region(rot)
res = {va: [], vb: []}
for rep in range(4):
    for v in (va, vb):
        g.set_option(opt, v)
        res[v].append((region([pairs]), region(rot), latency()))
for v in (va, vb):
    a = np.array(res[v])
    print("%s=%d  fixed %.4f ms (%.2f M/s)  rotating %.4f ms (%.2f M/s)  latency %.4f ms   [runs fixed: %s]" % (
        opt, v, np.median(a[:, 0]), E / np.median(a[:, 0]) / 1e3, np.median(a[:, 1]), E / np.median(a[:, 1]) / 1e3, np.median(a[:, 2]),
        " ".join("%.3f" % x for x in a[:, 0])))
