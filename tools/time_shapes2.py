"""Per-kernel times of the PD/PI batch on the other BASELINE shapes (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
for name in ("PPI", "Photo", "Computers"):
    n, e, k, hop, _ = synth.shaped_graph(name)
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    g = engine.DeviceGraph(rowptr, col, w)
    p = torch.from_numpy(np.ascontiguousarray(e, dtype=np.int32)).cuda()
    out = torch.empty((len(e), 25), dtype=torch.float64, device="cuda"); st = torch.empty(len(e), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.pd_pi_batch(p, hop, out=out, status=st)
    g.set_timing(True)
    torch.cuda.synchronize(); t0 = time.time()
    g.pd_pi_batch(p, hop, out=out, status=st)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(name, "%.3f ms" % (dt * 1e3), {k: round(v, 3) for k, v in g.timings().items() if v >= 0})
    g.close()
