"""One-rank RCCL smoke on the one-GPU box: the handle (and its eleven streams) first, then an NCCL process group of size 1, a few
collectives, and pipelined image batches beside them (development aid: the order bench.py uses at N > 1)."""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, ".")
from tlc_gnn_amd import engine
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t); dist.barrier()
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
def region(K=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(pairs, 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
region(8)
a = region()
gathered = torch.empty(2 * (1 << 20), device="cuda")[: 1 << 20]
dist.all_gather_into_tensor(gathered, t)
b = region()
print("one-rank nccl group: all_reduce ok (%.0f), pipelined batch %.3f / %.3f ms (%.1f M images/s)" % (float(t[0]), a, b, E / min(a, b) / 1e3))
dist.destroy_process_group()
