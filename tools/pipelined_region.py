"""K pipelined image batches of the bench workload (tlc_pd_pi_batch_async, joined once) for rocprofv3 --kernel-trace: how the
chunks' kernels overlap.  python tools/pipelined_region.py K [option value ...]"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
for k in range(2, len(sys.argv) - 1, 2):
    g.set_option(sys.argv[k], int(sys.argv[k + 1]))
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(pairs, 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
    e1.record(); torch.cuda.synchronize()
    print("region %d: %.4f ms per batch" % (rep, e0.elapsed_time(e1) / K))
