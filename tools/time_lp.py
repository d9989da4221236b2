"""Device time of the LP leg's kernels, one process (development aid): python tools/time_lp.py"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops
W = bench.build_workload(0)
n = W["n"]; dev = torch.device("cuda")
torch.manual_seed(1234)
rs = np.random.RandomState(1)
E = 75352
pairs = torch.from_numpy(rs.randint(0, n, size=(E, 2)).astype(np.int32)).to(dev)
emb = torch.randn(n, 16, device=dev) * 0.3
pi = torch.rand(E, 25, dtype=torch.float64, device=dev)
l1w = torch.randn(25, 41, device=dev) * 0.2; l1b = torch.randn(25, device=dev) * 0.1
l2w = torch.randn(1, 25, device=dev) * 0.2; l2b = torch.randn(1, device=dev) * 0.1
out = torch.empty(E, device=dev)
def t(fn, reps=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("lp_decode %d pairs: %.2f us per call (back to back)" % (E, t(lambda: ops.lp_decode(pairs, emb, pi, l1w, l1b, l2w, l2b, out=out))))
print("checksum %.9f" % float(out.double().sum()))
