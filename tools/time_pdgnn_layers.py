"""The PDGNN forward of bench.py's pdgnn block with and without the tiled layers (GraphBatch(tiled=...)) -- development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import synth
from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GraphBatch
dev = torch.device("cuda:0")
n_graphs = 41127
e_all, f, node_offs, edge_offs = synth.hiv_shaped_molecules(n_graphs, 1234)
glob = e_all.astype(np.int64) + np.repeat(node_offs[:-1], np.diff(edge_offs))[:, None]
both = np.concatenate([glob, glob[:, ::-1]])
order = np.argsort(np.searchsorted(node_offs[1:], both[:, 0], side="right"), kind="stable")
both = both[order]
eptr = np.concatenate([[0], np.cumsum(2 * np.diff(edge_offs))]).astype(np.int64)
n_tot = int(node_offs[-1])
loops = np.arange(n_tot)
ei = torch.from_numpy(np.concatenate([both, np.stack([loops, loops], 1)]).T.copy()).to(dev)
x = torch.from_numpy(f.astype(np.float32)).view(-1, 1).to(dev)
torch.manual_seed(1234)
model = Teacher_Model(type='GAT').eval().to(dev)
gptr = torch.from_numpy(node_offs).to(dev); d_eptr = torch.from_numpy(eptr).to(dev)
def med_ms(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
outs = {}
with torch.no_grad():
    for tiled in (False, True):
        t0 = time.perf_counter(); gb = GraphBatch(ei, n_tot, tiled=tiled); torch.cuda.synchronize(); tb = (time.perf_counter() - t0) * 1e3
        outs[tiled] = model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=gptr, edge_ptr=d_eptr, csr=gb)
        ms = med_ms(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=gptr, edge_ptr=d_eptr, csr=gb))
        print("tiled=%s: GraphBatch %.2f ms (first build), forward %.3f ms, tiles %s" % (tiled, tb, ms, None if gb.tiles is None else gb.tiles.numel() - 1))
a, b = [o[0] if isinstance(o, (tuple, list)) else o for o in (outs[False], outs[True])]
print("images: max abs diff %.3e (max |image| %.3e)" % (float((a - b).abs().max()), float(a.abs().max())))
