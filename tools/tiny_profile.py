"""Development aid: cycles per phase of tlc_pd_tiny_kernel (per wavefront, mean)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import engine, _lib
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = torch.as_tensor(wl["pi_pairs"]).cuda()
L = _lib.lib()
L.tlc_debug_phase_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
g.pd_pi_batch(pairs, 2)
L.tlc_debug_phase_profile(g._h, 1, None, 0, None)
g.pd_pi_batch(pairs, 2)
buf = (C.c_uint64 * 256)()        # (TLC_N_TIERS + 1) rows of 32; the library writes at most what is passed as capacity
L.tlc_debug_phase_profile(g._h, 0, C.cast(buf, C.c_void_p), 256, None)
a = np.array(list(buf), dtype=np.float64).reshape(8, 32)
r = a[5]
names = ["load slot", "filtration", "sort+asc pass", "sort+desc pass", "tree+cycle swap", "image", "store"]
print("tiny kernel: %d wavefronts; mean cycles per wavefront:" % r[14], {nm: int(r[k] / max(r[14], 1)) for k, nm in enumerate(names)}, "total", int(r[:7].sum() / max(r[14], 1)))
