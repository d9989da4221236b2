"""The PubMed feature projection alone (19 717 x 500 @ 500 x 100), 100 launches -- for profiler passes."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import ops
torch.manual_seed(0)
M, K, N = 19717, 500, 100
A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(5): ops.gemm(A, B, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): ops.gemm(A, B, out=out)
b.record(); torch.cuda.synchronize()
print("gemm %d x %d x %d: %.2f us per launch" % (M, K, N, a.elapsed_time(b) * 10))
