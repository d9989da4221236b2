"""Timing of the stand-alone PI raster (tlc_pi_raster) on synthetic diagram batches (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine

rs = np.random.RandomState(3)
B = 37676


def run(name, k, wide=False):
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(k)])).cuda()
    tot = int(k.sum())
    g = torch.Generator(device="cuda").manual_seed(1)
    pts = torch.rand((tot, 2), generator=g, device="cuda", dtype=torch.float64)
    pts[:, 1] = pts[:, 0] + pts[:, 1] * (1 - pts[:, 0])
    if wide:
        pts = pts * 3.0 - 1.0
        pts[:, 1] = pts[:, 0] + (pts[:, 1] - pts[:, 0]).abs()
    for _ in range(3):
        engine.pi_raster(offs, pts, 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        engine.pi_raster(offs, pts, 5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = 16.0 * tot + 200.0 * len(k) + 8.0 * (len(k) + 1)
    print("%-36s %8.1f us  %7.1f GB/s of algorithmic bytes (%d diagrams, %d points, max %d)" % (name, us, byt / us / 1e3, len(k), tot, k.max()))


geo = np.minimum(np.maximum(1, rs.geometric(1 / 41.0, size=B)), 500).astype(np.int64)
big = geo.copy()
big[rs.randint(0, B, 70)] = rs.randint(600, 4096, 70)
run("uniform 48 points", np.full(B, 48, dtype=np.int64))
run("uniform 16 points", np.full(B, 16, dtype=np.int64))
run("geometric mean 41 (<= 500)", geo)
run("geometric + 70 of 600..4096", big)
run("geometric + 70 big, [-1,2]^2 (erfc)", big, wide=True)
run("8 diagrams of 100 000 points", np.full(8, 100000, dtype=np.int64))
