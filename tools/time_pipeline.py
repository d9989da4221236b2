"""A large batch as sub-batches alternating over two handles / two streams (development aid): does chunk k+1's COUNT overlap
chunk k's tier kernels?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
for name in ("Computers", "Photo"):
    n, e, k, hop, _ = synth.shaped_graph(name)
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    gs = [engine.DeviceGraph(rowptr, col, w) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    p = torch.from_numpy(np.ascontiguousarray(e, dtype=np.int32)).cuda()
    E = len(e)
    out = torch.empty((E, 25), dtype=torch.float64, device="cuda"); st = torch.empty(E, dtype=torch.uint8, device="cuda")
    ref = torch.empty_like(out)
    for _ in range(3):
        gs[0].pd_pi_batch(p, hop, out=ref, status=st)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        gs[0].pd_pi_batch(p, hop, out=ref, status=st)
    torch.cuda.synchronize(); base = (time.time() - t0) / 5
    for parts in (2, 4, 8):
        bounds = np.linspace(0, E, parts + 1).astype(int)
        def run():
            cur = torch.cuda.current_stream()
            for s_ in streams:
                s_.wait_stream(cur)
            for i in range(parts):
                lo, hi = int(bounds[i]), int(bounds[i + 1])
                with torch.cuda.stream(streams[i % 2]):
                    gs[i % 2].pd_pi_batch(p[lo:hi], hop, out=out[lo:hi], status=st[lo:hi])
            for s_ in streams:
                cur.wait_stream(s_)
        for _ in range(3):
            run()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5):
            run()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print("%s: one call %.3f ms; %d parts over two handles %.3f ms (%.1f M images/s); equal %s" % (name, base * 1e3, parts, dt * 1e3, E / dt / 1e6, bool((out == ref).all())))
    for g in gs:
        g.close()
