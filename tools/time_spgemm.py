"""Device time of the sparse feature projection (tlc_spgemm_csr_dense_f32) on the bench workload's features, replayed from a graph of
20 calls (a Python call costs more than the kernel), next to the dense MFMA kernel.  Used with tools/gpu_spgemm_diag.sh."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops


def tg(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


wl = bench.build_workload(0)
x = torch.from_numpy(wl["x"]).cuda().contiguous()
torch.manual_seed(0)
w1 = torch.randn(x.shape[1], 100, device="cuda") * 0.05
xs = ops.SparseRows(x)
out = torch.empty((x.shape[0], 100), device="cuda")
dense = torch.empty_like(out)
print("sparse %.1f us | dense %.1f us | density %.3f" % (tg(lambda: ops.sparse_gemm(xs, w1, out=out)), tg(lambda: ops.gemm(x, w1, out=dense)), xs.density))
if "--densities" in sys.argv:
    for dens in (0.01, 0.02, 0.05, 0.2):
        xd = (torch.rand(x.shape, device="cuda") < dens).float() * torch.rand(x.shape, device="cuda")
        xsd = ops.SparseRows(xd)
        print("  density %.2f: sparse %.1f us" % (xsd.density, tg(lambda: ops.sparse_gemm(xsd, w1, out=out))))
