"""Development aid: where the submitting thread spends a pipelined chunk (TLC_HOST_TRACE=1 prints per chunk), for one option value.
python tools/host_trace.py <option> <value> [K]"""
import os, sys
os.environ["TLC_HOST_TRACE"] = "1"
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
opt, val = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
g.set_option(opt, val)
for rep in range(2):
    torch.cuda.synchronize()
    sys.stderr.write("---- %s=%d rep %d\n" % (opt, val, rep))
    for k in range(K):
        g.pd_pi_batch(pairs, 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
    torch.cuda.synchronize()
