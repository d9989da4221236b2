"""Sixty calls of Net.encode's one-call form (sparse features) and of the decode on the bench workload, for
`rocprofv3 --kernel-trace --stats -- python3 tools/time_lp_trace.py`: per-kernel durations of the LP leg without the host in the way."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tlc_gnn_amd import ops
wl = bench.build_workload(0)
n = wl["n"]; te = wl["train_edges"]
ei = torch.from_numpy(np.concatenate([te, te[:, ::-1]]).T.copy()).long().cuda()
rp, col, val = ops.gcn_norm_csr(ei, n)
x = torch.from_numpy(wl["x"]).cuda().contiguous()
torch.manual_seed(0)
w1 = torch.randn(x.shape[1], 100, device="cuda") * 0.05; b1 = torch.randn(100, device="cuda") * 0.1
w2 = torch.randn(100, 16, device="cuda") * 0.1; b2 = torch.randn(16, device="cuda") * 0.1
l1w = torch.randn(25, 41, device="cuda") * 0.3; l1b = torch.randn(25, device="cuda") * 0.1
l2w = torch.randn(1, 25, device="cuda") * 0.3; l2b = torch.randn(1, device="cuda") * 0.1
pairs = torch.from_numpy(np.concatenate([wl["pi_pairs"].astype(np.int64), wl["neg"]]).astype(np.int32)).cuda()
pi32 = torch.rand((pairs.shape[0], 25), device="cuda")
xs = ops.SparseRows(x)
emb = torch.empty((n, 16), device="cuda"); prob = torch.empty(pairs.shape[0], device="cuda")
for _ in range(60):
    ops.gcn2_encode(rp, col, val, x, w1, b1, w2, b2, relu=True, renorm=True, x_sparse=xs, out=emb)
    ops.lp_decode(pairs, emb, pi32, l1w, l1b, l2w, l2b, out=prob)
torch.cuda.synchronize()
