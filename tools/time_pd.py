"""Quick timing of the PD/PI batch on the PubMed-shaped graph (development aid; bench.py is the contract)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth

n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e, k)
g = engine.DeviceGraph(rowptr, col, w)
rs = np.random.RandomState(7)
pairs = torch.as_tensor(e[rs.permutation(len(e))[:37676]].astype(np.int32)).cuda()
out = torch.empty((len(pairs), 25), dtype=torch.float64, device="cuda")
st = torch.empty(len(pairs), dtype=torch.uint8, device="cuda")
for _ in range(3):
    g.pd_pi_batch(pairs, 2, out=out, status=st)
torch.cuda.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.time()
for _ in range(K):
    g.pd_pi_batch(pairs, 2, out=out, status=st)
torch.cuda.synchronize()
dt = (time.time() - t0) / K
print("ms per batch %.3f  PI/s %.3e" % (dt * 1e3, len(pairs) / dt), g.stats())
