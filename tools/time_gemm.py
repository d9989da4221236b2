"""Times tlc_gemm_f32 on the PubMed encoder shapes against torch.mm (rocBLAS) -- diagnostic."""
import os, sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import ops

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.manual_seed(0)
tag = "gemm16"
for (M, K, N) in [(19717, 500, 100), (19717, 100, 16), (2708, 1433, 100), (13752, 767, 100), (13752, 768, 100), (7650, 745, 64), (5000, 37, 7), (1, 3, 1), (81, 64, 128)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); bias = torch.randn(N, device="cuda")
    out = ops.gemm(A, B, bias, relu=True)
    ref = torch.relu(A.double() @ B.double() + bias.double())
    err = ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    us = t(lambda: ops.gemm(A, B, bias, relu=True))
    us_t = t(lambda: torch.relu(torch.addmm(bias, A, B)))
    print("%-9s M=%6d K=%5d N=%4d  tlc %.1f us  torch %.1f us  relerr %.2e  (%.1f TFLOP/s)" % (tag, M, K, N, us, us_t, err, 2.0 * M * K * N / us / 1e6))
