"""Runs the sparse feature projection 50 times per density so that a -DTLC_SQ_DEBUG build prints its per-wavefront cycle breakdown."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops
wl = bench.build_workload(0)
x = torch.from_numpy(wl["x"]).cuda().contiguous()
torch.manual_seed(0)
w1 = torch.randn(x.shape[1], 100, device="cuda") * 0.05
out = torch.empty((x.shape[0], 100), device="cuda")
for dens in (None, 0.01):
    xd = x if dens is None else (torch.rand(x.shape, device="cuda") < dens).float() * torch.rand(x.shape, device="cuda")
    xs = ops.SparseRows(xd)
    print("density %.3f" % xs.density, flush=True)
    for _ in range(50):
        ops.sparse_gemm(xs, w1, out=out)
    torch.cuda.synchronize()
