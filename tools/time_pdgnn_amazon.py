"""bench.py's pdgnn_amazon block alone (PDGNN forward on the hop-1 vicinities of the Amazon-shaped graphs) -- development aid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.pdgnn_amazon_aux(torch, torch.device("cuda:0")), indent=1))

# where one forward's wall clock goes (Photo): every part alone, host clock around a synchronised call
import time
import numpy as np
from tlc_gnn_amd import synth, ops, engine
from tlc_gnn_amd.Knowledge_Distillation.data_utils_LP import Vicinities
from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GraphBatch, GATConv
dev = torch.device("cuda:0")
n, edges, kappa, hop, _ = synth.shaped_graph("Photo")
ricci = np.concatenate([np.concatenate([edges, kappa[:, None]], 1), np.concatenate([edges[:, ::-1], kappa[:, None]], 1)]).tolist()
vic = Vicinities(edges, ricci)
pairs = edges[np.random.RandomState(1234).permutation(len(edges))[:4096]]
b = vic.batch(pairs, hop, node_cap=512, edge_cap=8192)
model = Teacher_Model(type='GAT').eval().to(dev)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r
node_ptr, edge_ptr = b["node_ptr"], b["edge_ptr"]
def prep():
    n_tot = int(node_ptr[-1])
    e = b["edges"].long() + node_ptr[b["pair_of_edge"]].view(-1, 1)
    loops = torch.arange(n_tot, device=e.device)
    return torch.cat([e.t(), torch.stack([loops, loops])], dim=1), b["f"].to(torch.float32).view(-1, 1)
ms, (ei, x) = t(prep); print("edge_index prep        %.3f ms" % ms)
nn = x.shape[0]
ms, (rowptr, col) = t(lambda: GATConv.csr_by_target(ei, nn)); print("csr_by_target          %.3f ms" % ms)
ms, tiles = t(lambda: ops.gat_tiles(rowptr, col, nn)); print("gat_tiles              %.3f ms" % ms)
with torch.no_grad():
    for tiled in (True, False):
        gb = GraphBatch(ei, nn, tiled=tiled)
        ms, h = t(lambda: model.DIM0_Model(x, ei, csr=gb)); print("4 layers tiled=%-5s    %.3f ms" % (tiled, ms))
        ms, _ = t(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=node_ptr, edge_ptr=edge_ptr, csr=gb))
        print("model, held GraphBatch %.3f ms" % ms)
    ms, _ = t(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=node_ptr, edge_ptr=edge_ptr)); print("model, per-call batch  %.3f ms" % ms)
