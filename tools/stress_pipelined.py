"""Stress of the pipelined image batches: R rounds of three batches in flight (one per workspace, the third one's second half submitted
by the join), rows compared bit for bit with the stream-ordered call after every round; variants with other work on the caller's stream
between submission and join (a large copy; the collectives of a one-rank RCCL group).  Prints every mismatch with its rows and sizes.
python tools/stress_pipelined.py [rounds] [shape] [scale] [n_pairs]"""
import os, socket, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tlc_gnn_amd import engine, synth
from tlc_gnn_amd import dist as tdist
R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
shape = sys.argv[2] if len(sys.argv) > 2 else "PubMed"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
NP = int(sys.argv[4]) if len(sys.argv) > 4 else 12000
n, edges, kappa, hop, _ = synth.shaped_graph(shape, scale=scale)
rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
g = engine.DeviceGraph(rowptr, col, w, device=0)
pairs = torch.as_tensor(np.ascontiguousarray(edges[:NP], dtype=np.int32)).cuda()
E = len(pairs)
want, want_st = g.pd_pi_batch(pairs, hop)
torch.cuda.synchronize()
nn, mm = g.sizes(E)
print("tiers", g.stats())
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)

def run(name, between):
    bad_rounds = 0
    for rnd in range(R):
        for k in range(3):
            outs[k].fill_(-1.0); sts[k].fill_(255)                      # (a skipped row shows too)
        for k in range(3):
            g.pd_pi_batch(pairs, hop, out=outs[k], status=sts[k], async_=True)
        between()
        g.join()
        torch.cuda.synchronize()
        for k in range(3):
            if not (torch.equal(outs[k], want) and torch.equal(sts[k], want_st)):
                bad = torch.nonzero((outs[k] != want).any(dim=1) | (sts[k] != want_st)).view(-1)
                idx = bad[:6].cpu().numpy()
                bad_rounds += 1
                print("%s round %d buffer %d: %d rows differ; first %s n %s m2 %s max|diff| %.3e status %s / %s" % (
                    name, rnd, k, bad.numel(), idx.tolist(), nn[idx].tolist(), mm[idx].tolist(), float((outs[k] - want).abs().max()),
                    sts[k][bad[:6]].tolist(), want_st[bad[:6]].tolist()))
    print("%s: %d rounds, %d with a mismatch" % (name, R, bad_rounds))

run("plain", lambda: None)
run("copy on the caller's stream", lambda: big2.copy_(big))
import torch.distributed as dist
s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
parts = tdist.shard_pairs_interleaved(tdist.ball_bound(rowptr, col, hop)[pairs.cpu().numpy()].min(1), 1)
pr = tdist.PaddedRows(E, 1, 0)
def coll():
    tdist.gather_shards_indexed(want, parts, always_collective=True)
    pr.send("rows", 25, want).copy_(want)
    pr.gather("rows")
run("one-rank RCCL collectives", coll)
dist.destroy_process_group()
g.close()
