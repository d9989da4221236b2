"""Exploration (development aid): which constructed graphs make tlc_pd_dc_kernel give a subgraph back to the serial walk."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from tlc_gnn_amd import engine, synth
from oracle import oracle

def hub_graph(L, K, rs, mode):
    e = set((0, k) for k in range(1, L + 1))
    while len(e) < L + K:
        a, b = rs.randint(1, L + 1, size=2)
        if a != b: e.add((min(a, b), max(a, b)))
    e = np.array(sorted(e), dtype=np.int64)
    if mode[0] == "dec":
        kappa = np.round(rs.uniform(-0.5, 0.9, size=len(e)), mode[1])
    elif mode[0] == "grid":                     # weights 1 + j * step, j small: distances collide up to rounding
        kappa = rs.randint(0, mode[2], size=len(e)) * mode[1]
    elif mode[0] == "two":                      # two weight values only, runs of equal keys below the 64 limit through jitter on few
        kappa = np.where(rs.rand(len(e)) < 0.5, 0.0, 0.5) + (rs.rand(len(e)) < mode[1]) * rs.uniform(0, 1e-9, size=len(e))
    return L + 1, e, kappa

found = 0
for seed in range(40):
    rs = np.random.RandomState(100 + seed)
    mode = [("dec", 2), ("dec", 3), ("grid", 1e-7, 50), ("grid", 1e-9, 1000), ("grid", 1e-12, 100000), ("two", 0.3), ("grid", 2.0 ** -40, 4096), ("dec", 1)][seed % 8]
    L, K = int(rs.choice([600, 800, 1100])), int(rs.choice([200, 400, 800]))
    n, e, kappa = hub_graph(L, K, rs, mode)
    rowptr, col, w = synth.edges_to_csr(n, e, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = np.array([[0, k] for k in range(1, 9)], dtype=np.int32)
    out, st = g.pd_pi_batch(torch.as_tensor(pairs).cuda(), 2)
    ran, back = g.dc_stats()
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, 2, n_threads=0)
    o = out.cpu().numpy()
    nz = ref != 0
    err = (np.abs(o[nz] - ref[nz]) / np.abs(ref[nz])).max() if nz.any() else 0.0
    found += back > 0
    print("seed %d mode %s L=%d K=%d: dc ran %d, gave back %d; status ok %s; err %.1e" % (seed, mode, L, K, ran, back, np.array_equal(st.cpu().numpy(), rst), err))
    g.close()
print("graphs with a give-back:", found)
