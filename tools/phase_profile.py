"""Per-phase cycle shares of the PD tier kernels (diagnostic: thread 0's clock64 deltas, summed over workgroups)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth, _lib

NAMES = ["load", "bellman-ford", "tight+chain", "tie fallback", "normalise+edges", "sort asc", "uf asc", "sort desc",
         "uf desc", "tree bfs", "ext1 serial", "image", "TOTAL", "max per wg", "n wg"]
n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e, k)
g = engine.DeviceGraph(rowptr, col, w)
rs = np.random.RandomState(7)
pairs = torch.as_tensor(e[rs.permutation(len(e))[:37676]].astype(np.int32)).cuda()
if len(sys.argv) > 1 and sys.argv[1] == "bench":         # the bench batch itself (bench.build_workload: the training graph, its 37 676 positives)
    import bench
    Wb = bench.build_workload(0)
    g = engine.DeviceGraph(Wb["rowptr"], Wb["col"], Wb["w"])
    pairs = torch.as_tensor(Wb["pi_pairs"]).cuda()
if len(sys.argv) > 1 and sys.argv[1] == "near":          # the strong-scaling list: non-edges within hop distance
    ci = engine.ComplementIndex(rowptr, col, device=0)
    near, ranks = engine.near_pairs(ci, hop)
    pairs = near[torch.argsort(ranks)].contiguous()
L = _lib.lib()
L.tlc_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
g.pd_pi_batch(pairs, 2)
L.tlc_debug_phase_profile(g._h, 1, None, 0, None)
g.pd_pi_batch(pairs, 2)
buf = (C.c_uint64 * 320)()        # (TLC_N_TIERS + 2) rows of 32; the library writes at most what is passed as capacity
L.tlc_debug_phase_profile(g._h, 0, C.cast(buf, C.c_void_p), 320, None)
a = np.array(list(buf), dtype=np.float64).reshape(10, 32)
for t, tn in enumerate(["small", "medium", "large", "huge", "mid", "tiny (see tiny_profile.py)", "medhi", "medwide"]):
    if t == 5:
        continue
    if a[t, 14] == 0:
        continue
    print("tier %s: %d workgroups, mean %.0f cycles, max %.0f cycles" % (tn, a[t, 14], a[t, 12] / a[t, 14], a[t, 13]))
    for i in range(12):
        print("   %-16s %6.2f %%   mean %9.0f   slowest wg %9.0f" % (NAMES[i], 100 * a[t, i] / max(a[t, 12], 1), a[t, i] / a[t, 14], a[t, 16 + i]))
    print("   slowest wg: n=%d m=%d" % (a[t, 28], a[t, 29]))
    print("   ext1 split (sum over wgs): pre-walk %.0f  walk %.0f  post %.0f  (ext1 total %.0f)" % (a[t, 15], a[t, 30], a[t, 31], a[t, 10]))

c = a[3]
if c[4] > 0:
    print("COUNT pass: %d pairs; mean cycles per pair: balls %.0f | S sweep + id list %.0f | induced count %.0f | small-tier write %.0f" % (c[4], c[0] / c[4], c[1] / c[4], c[2] / c[4], c[3] / c[4]))
c = a[8]                     # row TLC_N_TIERS: the early pass
if c[4] > 0:
    print("early pass: %d pairs; mean cycles per pair: balls %.0f | S sweep + id list %.0f | induced count %.0f | write %.0f" % (c[4], c[0] / c[4], c[1] / c[4], c[2] / c[4], c[3] / c[4]))
    print("   early count pass, wave 0: short-row cycles %.0f | long-row cycles %.0f | batches %d | long rows %d" % (c[8], c[9], c[10], c[11]))
