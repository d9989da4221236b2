"""Times the steps of ops.gat_tiles on the HIV-shaped batch -- development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import synth, ops
from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GATConv
dev = torch.device("cuda:0")
e_all, f, node_offs, edge_offs = synth.hiv_shaped_molecules(41127, 1234)
glob = e_all.astype(np.int64) + np.repeat(node_offs[:-1], np.diff(edge_offs))[:, None]
both = np.concatenate([glob, glob[:, ::-1]])
n = int(node_offs[-1]); loops = np.arange(n)
ei = torch.from_numpy(np.concatenate([both, np.stack([loops, loops], 1)]).T.copy()).to(dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r
ms, (rowptr, col) = t(lambda: GATConv.csr_by_target(ei, n)); print("csr_by_target %.3f ms" % ms)
ms, tiles = t(lambda: ops.gat_tiles(rowptr, col, n)); print("gat_tiles     %.3f ms  (%d tiles)" % (ms, tiles.numel() - 1))
