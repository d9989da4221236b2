"""development: the lane-per-pair extraction (xl_cut) on the dataset shapes where nearly every pair is TINY (hop 1), all positives of the
shape in one stream-ordered batch and in pipelined batches."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tlc_gnn_amd import engine, synth
for name in sys.argv[1:] or ["Photo", "Computers"]:
    n, e, k, hop, _ = synth.shaped_graph(name)
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = torch.as_tensor(e.astype(np.int32)).cuda()
    E = len(e)
    outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
    sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
    ref = None
    for cut in (0, 16, 24, 0, 24):
        g.set_option("xl_cut", cut)
        g.set_option("xl_pipelined", 1 if cut else 0)
        g.pd_pi_batch(pairs, hop, out=outs[0], status=sts[0])
        lat = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.pd_pi_batch(pairs, hop, out=outs[0], status=sts[0]); torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        if ref is None: ref = outs[0].clone()
        same = bool(torch.equal(outs[0], ref))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k2 in range(12):
            g.pd_pi_batch(pairs, hop, out=outs[k2 % 3], status=sts[k2 % 3], async_=True)
        g.join(); torch.cuda.synchronize()
        pip = (time.perf_counter() - t0) / 12
        print("%s hop %d, %d pairs, xl_cut=%2d: one batch %.3f ms (%.1f M/s), pipelined %.3f ms (%.1f M/s)  rows equal %s  xl %s" % (
            name, hop, E, cut, np.median(lat) * 1e3, E / np.median(lat) / 1e6, pip * 1e3, E / pip / 1e6, same, g.xl_stats()))
