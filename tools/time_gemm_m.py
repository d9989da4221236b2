"""tlc_gemm_f32 at K=500, N=100 over M (development aid: prologue vs per-tile cost of the bf16x3 kernel)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import ops
torch.manual_seed(0)
for M in (4096, 8192, 19717, 40000, 80000):
    a = torch.rand(M, 500, device="cuda") * 0.2
    b = torch.randn(500, 100, device="cuda") * 0.1
    bias = torch.zeros(100, device="cuda")
    out = torch.empty(M, 100, device="cuda")
    for _ in range(5):
        ops.gemm(a, b, bias, relu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.gemm(a, b, bias, relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    ref = torch.relu(a.double() @ b.double())
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print("M=%6d  %.1f us per call   max err / max |C| = %.2e" % (M, e0.elapsed_time(e1) * 1e3 / 50, err))
