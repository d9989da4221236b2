"""The main extraction's time on subsets of the bench batch split by the smaller ball's size (<= 64: the pairs x_sweep_ball takes;
> 64: the binned pairs that sweep the members' rows) -- which part bounds tlc_extract_kernel<64>?"""
import sys
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
n = W["n"]; rowptr, col = W["rowptr"], W["col"]
A = sp.csr_matrix((np.ones(len(col), dtype=np.int8), col, rowptr), shape=(n, n))
A = ((A + sp.identity(n, dtype=np.int8, format="csr")) > 0).astype(np.int32)
bsz = np.diff(((A @ A) > 0).tocsr().indptr)
P = W["pi_pairs"]
mn = np.minimum(bsz[P[:, 0]], bsz[P[:, 1]])
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
g.set_timing(True)
def run(name, sel, opt):
    pairs = torch.as_tensor(np.ascontiguousarray(P[sel])).cuda()
    E = len(pairs)
    out = torch.empty((E, 25), dtype=torch.float64, device="cuda"); st = torch.empty(E, dtype=torch.uint8, device="cuda")
    g.set_option("ball_edges", opt)
    acc = {}
    lat = []
    for rep in range(8):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.pd_pi_batch(pairs, 2, out=out, status=st); e1.record(); torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
        if rep >= 2:
            for k, x in g.timings().items():
                acc.setdefault(k, []).append(x)
    print("%-28s ball_edges=%d pairs %6d latency %.3f ms | " % (name, opt, E, float(np.median(lat))) + "  ".join("%s %.3f" % (k.replace("pd_tier_", "").replace("vicinity_", ""), float(np.median(x))) for k, x in acc.items()))
for opt in (0, 1):
    run("all", np.ones(len(P), bool), opt)
    run("min ball <= 64", mn <= 64, opt)
    run("min ball > 64", mn > 64, opt)
    run("64 < min ball <= 128", (mn > 64) & (mn <= 128), opt)
    run("min ball > 128", mn > 128, opt)
