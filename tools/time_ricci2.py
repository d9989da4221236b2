import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tlc_gnn_amd import engine, synth
n, e, k, hop, _ = synth.shaped_graph("PubMed")
rowptr, col, w = synth.edges_to_csr(n, e)
deg = np.diff(rowptr)
prod = (deg[e[:, 0]] + 1) * (deg[e[:, 1]] + 1)
hub = e[(prod > 8192) | (deg[e[:, 0]] + deg[e[:, 1]] + 2 > 256)]
print("deg max", deg.max(), "hub edges", len(hub), "max prod", prod.max(), "max support", (deg[e[:,0]]+deg[e[:,1]]+2).max())
engine.ollivier_ricci_sinkhorn(rowptr, col, hub[:8])
for sub, mi in ((hub[:1], 0), (hub[:1], 1000), (hub[:256], 0), (hub[:256], 10), (hub[:256], 1000), (hub, 0), (hub, 1000)):
    torch.cuda.synchronize(); t0 = time.time()
    kap, it = engine.ollivier_ricci_sinkhorn(rowptr, col, sub, max_iter=mi, want_iters=True)
    print("%5d edges max_iter %4d: %8.2f ms (iterations max %d, products max %d)" % (len(sub), mi, (time.time() - t0) * 1e3, it.max(), ((deg[sub[:,0]]+1)*(deg[sub[:,1]]+1)).max()))
