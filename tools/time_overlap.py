"""Two batches in flight: alternate two graph handles (own workspace each) on two streams (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, synth

import bench
W = bench.build_workload(0)                      # the bench's own graph and batch
rowptr, col, w = W["rowptr"], W["col"], W["w"]
NH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
gs = [engine.DeviceGraph(rowptr, col, w) for _ in range(NH)]
streams = [torch.cuda.Stream() for _ in range(NH)]
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
outs = [torch.empty((len(pairs), 25), dtype=torch.float64, device="cuda") for _ in range(NH)]
sts = [torch.empty(len(pairs), dtype=torch.uint8, device="cuda") for _ in range(NH)]
for i in range(NH):
    for _ in range(3):
        with torch.cuda.stream(streams[i]):
            gs[i].pd_pi_batch(pairs, 2, out=outs[i], status=sts[i])
torch.cuda.synchronize()
# baseline: one handle, back to back
t0 = time.time()
for _ in range(K):
    gs[0].pd_pi_batch(pairs, 2, out=outs[0], status=sts[0])
torch.cuda.synchronize()
base = (time.time() - t0) / K
ref = outs[0].clone()
t0 = time.time()
for s in range(K):
    i = s % NH
    with torch.cuda.stream(streams[i]):
        gs[i].pd_pi_batch(pairs, 2, out=outs[i], status=sts[i])
torch.cuda.synchronize()
ov = (time.time() - t0) / K
same = all(bool((o == ref).all()) for o in outs)
print("handles %d: one at a time %.3f ms/batch (%.2f M PI/s); alternating %.3f ms/batch (%.2f M PI/s); outputs equal: %s; HWQ=%s"
      % (NH, base * 1e3, len(pairs) / base / 1e6, ov * 1e3, len(pairs) / ov / 1e6, same, os.environ.get("GPU_MAX_HW_QUEUES")))
