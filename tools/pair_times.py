"""Per-pair wall-clock stamps of the extraction kernel (PHASE_DEBUG build): where the COUNT pass's time goes, by vicinity size."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine, _lib

W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
L = _lib.lib()
g.pd_pi_batch(pairs, 2)
L.tlc_debug_phase_profile(g._h, 1, None, 0, None)
g.pd_pi_batch(pairs, 2)
torch.cuda.synchronize()
E = len(W["pi_pairs"])
buf = np.zeros((E, 16), dtype=np.uint64)
L.tlc_debug_pair_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
L.tlc_debug_pair_times(g._h, buf.ctypes.data_as(C.c_void_p), E)
n, m2 = g.sizes(E)
pb = (C.c_uint64 * 256)()
L.tlc_debug_phase_profile(g._h, 0, C.cast(pb, C.c_void_p), 256, None)
a = np.array(list(pb), dtype=np.float64).reshape(8, 32)
c = a[3]
t = buf.astype(np.float64) / 100.0                                  # us
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
print("pairs stamped %d of %d; kernel span %.1f us" % (ok.sum(), E, t[ok, 7].max() - t0))
dur = t[:, 7] - t[:, 0]
for name, sel in (("n<=16", ok & (n <= 16)), ("16<n<=64", ok & (n > 16) & (n <= 64)), ("64<n<=128", ok & (n > 64) & (n <= 128)), ("128<n<=512", ok & (n > 128) & (n <= 512)), ("n>512", ok & (n > 512))):
    if sel.sum():
        d = dur[sel]
        seg = [(t[sel, k + 1] - t[sel, k]).mean() for k in range(7)]
        print("%-12s %6d pairs  mean %.1f us  p50 %.1f  p99 %.1f  max %.1f | bounds %.1f  lists+mark %.1f  filter %.1f  unmark %.1f  member bits %.1f  sweep %.1f  end %.1f   sum %.0f us" % (
            (name, sel.sum(), d.mean(), np.median(d), np.percentile(d, 99), d.max()) + tuple(seg) + (d.sum(),)))
# by the size of the smaller ball (<= 64: the pairs x_sweep_ball takes from the ball's subgraph list, round 5)
import scipy.sparse as sp
_n = W["n"]
_A = sp.csr_matrix((np.ones(len(W["col"]), dtype=np.int8), W["col"], W["rowptr"]), shape=(_n, _n))
_A = ((_A + sp.identity(_n, dtype=np.int8, format="csr")) > 0).astype(np.int32)
_bsz = np.diff(((_A @ _A) > 0).tocsr().indptr)
_P = W["pi_pairs"]
mnb = np.minimum(_bsz[_P[:, 0]], _bsz[_P[:, 1]]); mxb = np.maximum(_bsz[_P[:, 0]], _bsz[_P[:, 1]])
for name, sel in (("minball<=16", ok & (mnb <= 16)), ("16<mb<=64", ok & (mnb > 16) & (mnb <= 64)), ("mb<=64,big>256", ok & (mnb <= 64) & (mxb > 256)), ("64<mb<=128", ok & (mnb > 64) & (mnb <= 128)), ("128<mb<=256", ok & (mnb > 128) & (mnb <= 256)), ("mb>256", ok & (mnb > 256))):
    if sel.sum():
        d = dur[sel]
        seg = [(t[sel, k + 1] - t[sel, k]).mean() for k in range(7)]
        print("%-15s %6d pairs  mean %.1f us  p50 %.1f  p99 %.1f  max %.1f | bounds %.1f  lists+mark %.1f  filter %.1f  unmark %.1f  member bits %.1f  sweep %.1f  end %.1f   sum %.0f us" % (
            (name, sel.sum(), d.mean(), np.median(d), np.percentile(d, 99), d.max()) + tuple(seg) + (d.sum(),)))
# inside the sweep (single-wavefront pairs that wrote their entries): record wait | round 0 | rest of short rows | long rows | heavy pairs
sw = ok & (t[:, 13] > 0)
for name, sel in (("n<=16", sw & (n <= 16)), ("16<n<=64", sw & (n > 16) & (n <= 64))):
    if sel.sum():
        print("   sweep of %-9s (%d pairs): sweep entry %.2f | record arrives %.2f | round 0 %.2f | more short rows %.2f | long rows %.2f | heavy pairs %.2f | return %.2f us" % (
            (name, sel.sum(), (t[sel, 8] - t[sel, 5]).mean()) + tuple((t[sel, k + 1] - t[sel, k]).mean() for k in range(8, 13)) + ((t[sel, 6] - t[sel, 13]).mean(),)))
# how many pairs are in flight over time, when the last pair of each size class ends
end = t[ok, 7] - t0
for q in (50, 90, 99, 100):
    print("  %3d %% of the pairs ended by %.1f us" % (q, np.percentile(end, q)))
# concurrency: pairs in flight, sampled every 10 us
st, en = t[ok, 0] - t0, t[ok, 7] - t0
for x in range(0, int(en.max()) + 1, 20):
    print("  t=%4d us: %5d pairs in flight" % (x, int(((st <= x) & (en > x)).sum())), end="")
print()
late = np.argsort(-t[:, 7] * ok)[:10]
print("last to end:", [(int(n[i]), int(m2[i]), round(float(t[i, 0] - t0), 1), round(float(dur[i]), 1)) for i in late])

# per wavefront (stamp 14 = the workgroup): time inside pairs, gaps between consecutive pairs, first start, last end
wg = buf[:, 14].astype(np.int64)
order = np.lexsort((t[:, 0], wg))
order = order[ok[order]]
wgs, st_s, en_s = wg[order], t[order, 0] - t0, t[order, 7] - t0
same = wgs[1:] == wgs[:-1]
gaps = (st_s[1:] - en_s[:-1])[same]
first = np.ones(len(order), bool); first[1:] = ~same
last = np.ones(len(order), bool); last[:-1] = ~same
print("wavefronts with pairs: %d; pairs per wavefront mean %.1f; gap between consecutive pairs of a wavefront: mean %.2f us p50 %.2f p90 %.2f p99 %.2f (sum %.0f us); first pair starts at mean %.1f us (p99 %.1f), last pair ends at mean %.1f us (p1 %.1f)" % (
    first.sum(), len(order) / first.sum(), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), np.percentile(gaps, 99), gaps.sum(),
    st_s[first].mean(), np.percentile(st_s[first], 99), en_s[last].mean(), np.percentile(en_s[last], 1)))
main = st_s[first] > 50
print("main-pass wavefronts: %d; first start mean %.1f, last end mean %.1f, in-pair time per wavefront mean %.1f us" % (main.sum(), st_s[first][main].mean(), en_s[last][main].mean(), 0.0))
