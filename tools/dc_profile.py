"""Development aid: where tlc_pd_dc_kernel's slowest subgraph spends its cycles (thread 0's clock)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine, _lib
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = torch.as_tensor(wl["pi_pairs"]).cuda()
g.set_option("dc_inplace", 0)      # (the stamps are tlc_pd_dc_kernel's: in place, the tier kernel's own phase row overwrites them)
L = _lib.lib()
L.tlc_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
g.pd_pi_batch(pairs, 2)
L.tlc_debug_phase_profile(g._h, 1, None, 0, None)
g.pd_pi_batch(pairs, 2)
buf = (C.c_uint64 * 256)()        # (TLC_N_TIERS + 1) rows of 32; the library writes at most what is passed as capacity
L.tlc_debug_phase_profile(g._h, 0, C.cast(buf, C.c_void_p), 256, None)
a = np.array(list(buf), dtype=np.float64).reshape(8, 32)
for t, tn in ((2, "large"), (1, "medium")):
    r = a[t]
    print("dc kernel, tier %s: slowest subgraph n=%d K=%d total %.0f cycles: setup %.0f | MSF %.0f | labels %.0f | renumber+move %.0f | Boruvka rounds %d | levels %d"
          % (tn, r[28], r[29], r[22], r[16], r[17], r[18], r[19], r[20], r[21]))
