"""Times tlc_spmm_csr_f32 on the bench graph's normalised CSR, full and with hub rows capped -- diagnostic."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

wl = bench.build_workload(0)
n = wl["n"]
te = wl["train_edges"]
ei = torch.from_numpy(np.concatenate([te, te[:, ::-1]]).T.copy()).long().cuda()
rp, col, val = ops.gcn_norm_csr(ei, n)
for cap in (None, 64, 16, 4):
    rpn, coln, valn = rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    if cap is not None:
        deg = np.minimum(np.diff(rpn), cap)
        nrp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        idx = np.concatenate([np.arange(rpn[i], rpn[i] + deg[i]) for i in range(n)])
        rpn, coln, valn = nrp, coln[idx], valn[idx]
    r, c, v = [torch.from_numpy(a).cuda() for a in (rpn, coln, valn)]
    for k in (100, 16):
        x = torch.randn(n, k, device="cuda"); y = torch.empty_like(x)
        us = t(lambda: ops.spmm(r, c, v, x, out=y))
        ref = torch.sparse_csr_tensor(r.long(), c.long(), v, (n, n)) @ x
        print("cap %s k=%d nnz=%d: %.1f us  maxerr %.2e" % (cap, k, len(coln), us, (y - ref).abs().max().item()))
