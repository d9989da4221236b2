"""Throughput on uniformly random pairs of the PubMed-shaped graph (the negative sweep of loaddatas.py:44-53 is mostly
pairs with d(u,v) > hop: exact zero rows).  Development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e, k)
g = engine.DeviceGraph(rowptr, col, w)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for k in range(2, len(sys.argv) - 1, 2):                  # python tools/time_sweep.py E [option value ...]
    g.set_option(sys.argv[k], int(sys.argv[k + 1]))
rs = np.random.RandomState(3)
pairs = torch.as_tensor(rs.randint(0, n, size=(E, 2)).astype(np.int32)).cuda()
out = torch.empty((E, 25), dtype=torch.float64, device="cuda")
st = torch.empty(E, dtype=torch.uint8, device="cuda")
g.set_timing(True)
for _ in range(2):
    g.pd_pi_batch(pairs, 2, out=out, status=st)
torch.cuda.synchronize()
t0 = time.time()
g.pd_pi_batch(pairs, 2, out=out, status=st)
torch.cuda.synchronize()
dt = time.time() - t0
nz = int((out.abs().sum(1) > 0).sum())
print("pairs %d in %.2f ms = %.3e pairs/s; non-zero rows %d (%.2f %%); status hist %s" % (E, dt * 1e3, E / dt, nz, 100.0 * nz / E,
      torch.bincount(st.long(), minlength=6).tolist()))
print(g.stats(), {k: round(v, 3) for k, v in g.timings().items()})
