"""Development aid: cycles per phase of tlc_xlane_kernel (per wavefront, mean) on the bench's batch."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from tlc_gnn_amd import engine, _lib
wl = bench.build_workload(0)
g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"])
pairs = torch.as_tensor(wl["pi_pairs"]).cuda()
L = _lib.lib()
for cut in ([int(a) for a in sys.argv[1:]] or [24]):
    g.set_option("xl_cut", cut)
    g.pd_pi_batch(pairs, 2)
    L.tlc_debug_phase_profile(g._h, 1, None, 0, None)
    g.pd_pi_batch(pairs, 2)
    buf = (C.c_uint64 * 320)()
    L.tlc_debug_phase_profile(g._h, 0, C.cast(buf, C.c_void_p), 320, None)
    a = np.array(list(buf), dtype=np.float64).reshape(10, 32)
    r = a[9]
    names = ["head", "intersection", "compaction", "record heads", "round 0", "segments", "heavy x heavy", "verdict"]
    nw = max(r[15], 1)
    print("cut %d: %s; %d wavefronts; mean cycles per wavefront:" % (cut, g.xl_stats(), r[15]), {nm: int(r[k] / nw) for k, nm in enumerate(names)},
          "total", int(r[:8].sum() / nw))
