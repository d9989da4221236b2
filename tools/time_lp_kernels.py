"""Times every kernel of the LP leg on the bench workload on its own (events on the current stream), and the whole leg --
diagnostic for the round-5 work on the leg (decode on the MFMA, conv2's projection in conv1's aggregate, the k=16 aggregate)."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops

def t(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def tg(fn, n=20, reps=10):
    """the same through a captured graph of n calls: what the device takes when the host is not in the way (a Python call of a
    wrapper is ~10 us: kernels shorter than that read as 10 us through t())"""
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
n = wl["n"]
te = wl["train_edges"]
ei = torch.from_numpy(np.concatenate([te, te[:, ::-1]]).T.copy()).long().cuda()
rp, col, val = ops.gcn_norm_csr(ei, n)
x = torch.from_numpy(wl["x"]).cuda().contiguous()
torch.manual_seed(0)
w1 = torch.randn(x.shape[1], 100, device="cuda") * 0.05; b1 = torch.randn(100, device="cuda") * 0.1
w2 = torch.randn(100, 16, device="cuda") * 0.1; b2 = torch.randn(16, device="cuda") * 0.1
l1w = torch.randn(25, 41, device="cuda") * 0.3; l1b = torch.randn(25, device="cuda") * 0.1
l2w = torch.randn(1, 25, device="cuda") * 0.3; l2b = torch.randn(1, device="cuda") * 0.1
pairs = torch.from_numpy(np.concatenate([wl["pi_pairs"].astype(np.int64), wl["neg"]]).astype(np.int32)).cuda()
E = pairs.shape[0]
pi64 = torch.rand((E, 25), device="cuda", dtype=torch.float64); pi32 = pi64.float()
xw = torch.empty((n, 100), device="cuda"); h = torch.empty_like(xw); hw = torch.empty((n, 16), device="cuda"); emb = torch.empty_like(hw)
prob = torch.empty(E, device="cuda"); prob2 = torch.empty(E, device="cuda")
print("feature gemm      %.1f us" % t(lambda: ops.gemm(x, w1, out=xw)))
print("spmm k=100        %.1f us" % t(lambda: ops.spmm(rp, col, val, xw, bias=b1, relu=True, out=h)))
print("gemm2 100->16     %.1f us" % t(lambda: ops.gemm(h, w2, out=hw)))
print("spmm k=16         %.1f us" % t(lambda: ops.spmm(rp, col, val, hw, bias=b2, relu=True, renorm=True, out=emb)))
print("decode f64 table  %.1f us" % t(lambda: ops.lp_decode(pairs, emb, pi64, l1w, l1b, l2w, l2b, out=prob)))
print("decode f32 table  %.1f us" % t(lambda: ops.lp_decode(pairs, emb, pi32, l1w, l1b, l2w, l2b, out=prob2)))
print("decode f32 == f64:", bool(torch.equal(prob, prob2)))
emb1 = torch.empty_like(emb)
print("gcn2_encode       %.1f us" % t(lambda: ops.gcn2_encode(rp, col, val, x, w1, b1, w2, b2, relu=True, renorm=True, out=emb1)))
print("encode fused vs four kernels: max abs diff %.2e" % float((emb1 - emb).abs().max()))
def leg():
    ops.gcn2_encode(rp, col, val, x, w1, b1, w2, b2, relu=True, renorm=True, out=emb1)
    ops.lp_decode(pairs, emb1, pi32, l1w, l1b, l2w, l2b, out=prob)
print("whole leg         %.1f us" % t(leg))
# decode against plain torch
d = (emb[pairs[:, 0].long()] - emb[pairs[:, 1].long()]).pow(2)
hh = torch.nn.functional.leaky_relu(torch.cat([d, pi32], 1) @ l1w.T + l1b, 0.2)
ref = 1.0 / (torch.exp(torch.clamp((hh @ l2w.T + l2b).abs().squeeze(1), 0, 40) - 2.0) + 1.0)
print("decode vs torch: max abs diff %.2e" % float((ref - prob2).abs().max()))
xs = ops.SparseRows(x)
xws = torch.empty_like(xw)
print("sparse feature gemm (density %.3f)  %.1f us  (from a graph: %.1f us; the dense kernel from a graph: %.1f us)"
      % (xs.density, t(lambda: ops.sparse_gemm(xs, w1, out=xws)), tg(lambda: ops.sparse_gemm(xs, w1, out=xws)), tg(lambda: ops.gemm(x, w1, out=xw))))
ops.gemm(x, w1, out=xw)
print("sparse vs dense product: max abs diff %.2e (max |.| %.2e)" % (float((xw - xws).abs().max()), float(xw.abs().max())))
for dens in (0.02, 0.05, 0.2, 0.3):
    xd = x * 0 + (torch.rand(x.shape, device="cuda") < dens).float() * torch.rand(x.shape, device="cuda")
    xsd = ops.SparseRows(xd)
    print("  density %.2f: sparse %.1f us (graph %.1f us)" % (xsd.density, t(lambda: ops.sparse_gemm(xsd, w1, out=xws)), tg(lambda: ops.sparse_gemm(xsd, w1, out=xws))))
