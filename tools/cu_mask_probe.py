"""Is the extraction bound by something inside a CU or by something the whole chip shares?  One batch alone on a stream that may use
all CUs, every other CU, or the first half / quarter of the mask (hipExtStreamCreateWithCUMask): development aid."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from tlc_gnn_amd import engine
import bench
hip = C.CDLL("libamdhip64.so")
def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
g.pd_pi_batch(pairs, 2)
torch.cuda.synchronize()
masks = {"all CUs": [0xffffffff] * 8, "every other CU": [0x55555555] * 8, "first half of the mask": [0xffffffff] * 4 + [0] * 4,
         "every fourth CU": [0x11111111] * 8, "first quarter": [0xffffffff] * 2 + [0] * 6}
for xg in (0, 2048):
    g.set_option("x_grid", xg)
    for name, m in masks.items():
        st = masked_stream(m)
        with torch.cuda.stream(st):
            g.set_timing(True)
            ts = []
            for _ in range(4):
                g.pd_pi_batch(pairs, 2)
                st.synchronize()
                t = g.timings()
                ts.append(t)
            g.set_timing(False)
        keys = [k for k in ts[-1] if ts[-1][k] >= 0]
        print("x_grid=%d  %-24s " % (xg, name) + "  ".join("%s %.3f" % (k, float(np.median([t[k] for t in ts[1:]]))) for k in keys), flush=True)
