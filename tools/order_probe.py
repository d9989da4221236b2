"""Does the ORDER of the MEDIUM list matter?  The tier lists come out in pair order (block-aggregated appends of the scan), so a permuted
batch gives a permuted list: the bench batch as it is / with its MEDIUM-sized pairs first, most Pos edges first / fewest first / with all
pairs sorted by size descending.  Pipelined regions, alternating.  python tools/order_probe.py [K]"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
P = np.ascontiguousarray(W["pi_pairs"])
E = len(P)
g.pd_pi_batch(torch.as_tensor(P).cuda(), 2)
n, m2 = g.sizes(E)
m = m2 // 2
pos = m - n + 1
med = ((n > 128) | (m > 256)) & (n <= 512) & (m <= 1024)
idx = np.arange(E)
def first(sel_order):
    rest = np.setdiff1d(idx, sel_order, assume_unique=True)
    return np.concatenate([sel_order, rest])
mi = idx[med]
orders = {"as it is": idx,
          "MEDIUM first, most Pos first": first(mi[np.argsort(-pos[mi], kind="stable")]),
          "MEDIUM first, fewest Pos first": first(mi[np.argsort(pos[mi], kind="stable")]),
          "MEDIUM last, most Pos first": np.concatenate([np.setdiff1d(idx, mi, assume_unique=True), mi[np.argsort(-pos[mi], kind="stable")]]),
          "all by edges descending": np.argsort(-m, kind="stable")}
# in place: the pairs of one tier permuted among their own positions (the extraction sees the same mix of sizes along the batch, the
# tier's LIST comes out in the new order)
small = (n > 16) & (n <= 64) & (m <= 128) & ~((n <= 16) & (m <= 24))
mid = ~small & ((n > 64) | (m > 128)) & (n <= 128) & (m <= 256)
def in_place(sel, key):
    o = idx.copy(); p_ = idx[sel]; o[p_] = p_[np.argsort(-key[p_], kind="stable")]; return o
orders["SMALL in place, most edges first"] = in_place(small, m)
orders["MEDIUM in place, most Pos first"] = in_place(med, pos)
orders["MEDIUM in place, most edges first"] = in_place(med, m)
o = in_place(small, m); o2 = idx.copy()
for sel, key in ((small, m), (mid, m), (med, pos)):
    p_ = idx[sel]; o2[p_] = p_[np.argsort(-key[p_], kind="stable")]
orders["SMALL / MID / MEDIUM in place"] = o2
wide = med & ((n > 384) | (m > 512) | (pos >= 320))
orders = {"as it is": idx,
          "SMALL in place, most edges first": in_place(small, m),
          "SMALL in place, two classes (m >= 40 first)": in_place(small, (m >= 40).astype(np.int64)),
          "MID in place, most Pos first": in_place(mid, pos),
          "MID in place, most edges first": in_place(mid, m),
          "MEDWIDE in place, most Pos first": in_place(wide, pos),
          "compact MEDIUM in place, most Pos first": in_place(med & ~wide, pos),
          "SMALL / MID / MEDIUM in place": o2}
batches = {k: torch.as_tensor(np.ascontiguousarray(P[o])).cuda() for k, o in orders.items()}
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
def region(b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(b, 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
for b in batches.values(): region(b)
res = {k: [] for k in batches}
for rep in range(5):
    for k, b in batches.items(): res[k].append(region(b))
for k, v in res.items():
    print("%-34s %.4f ms  [%s]" % (k, float(np.median(v)), " ".join("%.3f" % x for x in v)))
