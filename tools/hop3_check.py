"""hop 3 and 4 through the ball-list extraction: timings with the extraction on / off on a Cora-shaped and a PubMed-shaped graph
(run under rocprofv3 --kernel-trace --stats to see which kernels ran) -- development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
for shape, scale, hop in (("Cora", 1.0, 3), ("PubMed", 1.0, 3)):
    n, edges, kappa, _, _ = synth.shaped_graph(shape, scale=scale)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = torch.as_tensor(edges[np.random.RandomState(1).permutation(len(edges))[:8000]].astype(np.int32)).cuda()
    res = {}
    for ex in (1, 0):
        g.set_option("extract", ex)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out, st = g.pd_pi_batch(pairs, hop); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[ex] = (out.clone(), st.clone())
        print("%s hop %d extract=%d: %s ms  tiers %s" % (shape, hop, ex, " ".join("%.2f" % (t * 1e3) for t in ts), g.stats()))
    print("   rows equal to 1e-12:", bool((res[0][0] - res[1][0]).abs().max() <= 1e-12), "status equal:", bool(torch.equal(res[0][1], res[1][1])))
    g.close()
