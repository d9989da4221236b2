"""Development aid: the heavy pairs of the bench batch through tlc_pd_pi_batch several times; against the CPU checker, pair by pair."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine, _lib
from oracle import oracle
wl = bench.build_workload(0)
rowptr, col, w, hop = wl["rowptr"], wl["col"], wl["w"], wl["hop"]
pairs = wl["pi_pairs"]
deg = np.diff(rowptr)
score = np.minimum(deg[pairs[:, 0]], deg[pairs[:, 1]])
heavy = pairs[np.argsort(-score)[:int(sys.argv[1]) if len(sys.argv) > 1 else 300]]
g = engine.DeviceGraph(rowptr, col, w)
ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, heavy, hop, n_threads=0)
L = _lib.lib()
L.tlc_debug_dc_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
outs = []
for it in range(4):
    out, st = g.pd_pi_batch(torch.as_tensor(heavy).cuda(), hop)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 2)()
    L.tlc_debug_dc_stats(g._h, C.cast(buf, C.c_void_p), _lib.stream_ptr())
    o = out.cpu().numpy()
    nz = ref != 0
    rel = np.zeros_like(ref); rel[nz] = np.abs(o[nz] - ref[nz]) / np.abs(ref[nz])
    badrows = np.nonzero((rel.max(1) > 1e-8) | ((o == 0) != (ref == 0)).any(1))[0]
    print("run", it, "dc ok/fallback:", list(buf), "status equal:", np.array_equal(st.cpu().numpy(), rst), "bad rows:", len(badrows), badrows[:8],
          "max rel", rel.max())
    outs.append(o)
n_sz, m2 = g.sizes(len(heavy))
print("determinism:", [bool(np.array_equal(outs[0], x)) for x in outs[1:]])
d = np.nonzero((outs[0] != outs[1]).any(1))[0]
print("rows differing between runs:", d[:10], "their n/m/K:", [(int(n_sz[i]), int(m2[i] // 2), int(m2[i] // 2 - n_sz[i] + 1)) for i in d[:10]])
