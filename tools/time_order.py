"""Does the ORDER of the pairs inside a batch matter?  The fixed 37 676-pair batch as given, randomly permuted, and sorted by the smaller
ball bound (development aid; pipelined batches, one process)."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine, dist as tdist
import bench
K = 40
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
base = np.asarray(W["pi_pairs"])
E = len(base)
rs = np.random.RandomState(5)
cost = tdist.pair_cost(tdist.ball_bound(W["rowptr"], W["col"], W["hop"]), base)
variants = {"as given": base, "permuted": base[rs.permutation(E)], "permuted 2": base[rs.permutation(E)],
            "heaviest first": base[np.argsort(-cost, kind="stable")], "lightest first": base[np.argsort(cost, kind="stable")],
            "sorted in 1024s": np.concatenate([base[k:k + 1024][np.argsort(-cost[k:k + 1024], kind="stable")] for k in range(0, E, 1024)]),
            "sorted in 4096s": np.concatenate([base[k:k + 4096][np.argsort(-cost[k:k + 4096], kind="stable")] for k in range(0, E, 4096)])}
rot = [torch.from_numpy(b).cuda() for b in bench.rotated_batches(W, 8, seed=4321)]
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
def region(batches):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(batches[k % len(batches)], 2, out=outs[k % 3], status=sts[k % 3], async_=True)
    g.join()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
dev = {k: torch.as_tensor(np.ascontiguousarray(v)).cuda() for k, v in variants.items()}
for v in dev.values():
    region([v])
res = {k: [] for k in dev}
res["rotating"] = []
for rep in range(3):
    for k, v in dev.items():
        res[k].append(region([v]))
    res["rotating"].append(region(rot))
g.set_option("tier_mask", 0)
front = {}
for rep in range(3):
    for k in ("as given", "heaviest first", "lightest first"):
        front.setdefault(k, []).append(region([dev[k]]))
g.set_option("tier_mask", 255)
for k, v in front.items():
    print("first halves only (tier_mask 0)  %-16s %.4f ms   runs %s" % (k, np.median(v), " ".join("%.3f" % x for x in v)))
for k, v in res.items():
    print("%-16s %.4f ms (%.2f M/s)   runs %s" % (k, np.median(v), E / np.median(v) / 1e3, " ".join("%.3f" % x for x in v)))
print("first pairs as given:", base[:6].tolist())
