"""development: tlc_w2_partial_matching on B random diagram pairs of n points each (m = n targets)"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import ops
B = 8192
for n in (12, 25, 50, 100, 200):
    rs = np.random.RandomState(n)
    b0 = rs.rand(B * n); X = np.stack([b0, b0 + rs.uniform(-0.1, 0.6, size=B * n)], 1)
    b1 = rs.rand(B * n); Y = np.stack([b1, b1 + rs.uniform(0, 0.7, size=B * n)], 1)
    off = torch.arange(0, B * n + 1, n, dtype=torch.int64).cuda()
    Xd, Yd = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
    for _ in range(2): ops.w2_partial_matching(off, Xd, off, Yd, order=2, max_points=n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.w2_partial_matching(off, Xd, off, Yd, order=2, max_points=n)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("n = %3d: %.3f ms for %d problems = %.2f us per problem-wave at 8 per SIMD; n^2 = %d" % (n, ms, B, ms * 1e3 / (B / 1024.0) , n * n))
