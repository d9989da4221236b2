"""PD/PI batch time and tier split on the other dataset shapes (Cora, CiteSeer, Photo, Computers) -- development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
for name in sys.argv[1:] or ["Cora", "Photo", "Computers"]:
    n, e, k, hop, _ = synth.shaped_graph(name)
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    g = engine.DeviceGraph(rowptr, col, w)
    rs = np.random.RandomState(3)
    sel = e[rs.permutation(len(e))[:min(len(e), 20000)]].astype(np.int32)
    pairs = torch.as_tensor(sel).cuda()
    g.pd_pi_batch(pairs, hop)
    torch.cuda.synchronize()
    g.set_timing(True)
    t0 = time.perf_counter()
    out, st = g.pd_pi_batch(pairs, hop)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nn, m2 = g.sizes(len(sel))
    print(name, "n=%d m=%d hop=%d pairs=%d: %.2f ms (%.2f M PI/s)" % (n, len(e), hop, len(sel), dt * 1e3, len(sel) / dt / 1e6),
          {k2: v for k2, v in g.stats().items() if k2.startswith("tier")}, "max n/m:", nn.max(), m2.max() // 2,
          {k2: round(v, 2) for k2, v in g.timings().items() if v >= 0}, "dc (ran, gave back):", g.dc_stats())
    K = (m2 // 2 - nn + 1)
    big = (m2 // 2 > 256) | (nn > 128)
    print("   MEDIUM-sized: %d, of them Pos edges >= 320: %d; Pos edges of the ten largest: %s" % (big.sum(), (big & (K >= 320)).sum(), np.sort(K[big])[-10:].tolist()))
