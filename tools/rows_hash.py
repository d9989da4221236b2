"""sha256 of the bench batch's image rows and status bytes (stream-ordered call): two libraries give the same bits?"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
out, st = g.pd_pi_batch(pairs, 2)
torch.cuda.synchronize()
print("rows", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], "status", hashlib.sha256(st.cpu().numpy().tobytes()).hexdigest()[:16])
