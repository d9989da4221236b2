"""Several handle-option settings timed inside ONE process, alternating (boxes differ by several per cent): pipelined image batches,
fixed and rotating.  python tools/sweep_options.py "n_ws=4" "n_ws=4,gate_ticks=0" ...   (the empty string "" = defaults)
Defaults restored between settings are those in DEFAULTS below."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd import engine
import bench
DEFAULTS = {"tier_mask": 255, "n_ws": 3, "ball_edges": 1, "fast_split": 1, "dc_inplace": 1, "heavy": 1, "tiny": 1, "extract": 1}
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
K, REPS = 40, 3
W = bench.build_workload(0)
g = engine.DeviceGraph(W["rowptr"], W["col"], W["w"])
pairs = torch.as_tensor(W["pi_pairs"]).cuda()
E = len(pairs)
rot = [torch.from_numpy(b).cuda() for b in bench.rotated_batches(W, 8, seed=4321)]
outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(5)]
sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(5)]
def region(batches):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(K):
        g.pd_pi_batch(batches[k % len(batches)], 2, out=outs[k % 5], status=sts[k % 5], async_=True)
    g.join()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
def apply(cfg):
    for k, v in DEFAULTS.items():
        g.set_option(k, cfg.get(k, v))
for cfg in cfgs:
    apply(cfg); region([pairs]); region(rot)
res = [[] for _ in cfgs]
for rep in range(REPS):
    for i, cfg in enumerate(cfgs):
        apply(cfg)
        res[i].append((region([pairs]), region(rot)))
for cfg, r in zip(cfgs, res):
    a = np.array(r)
    print("%-40s fixed %.4f ms (%.2f M/s)  rotating %.4f ms (%.2f M/s)   [fixed runs: %s]" % (
        ",".join("%s=%d" % kv for kv in cfg.items()) or "defaults", np.median(a[:, 0]), E / np.median(a[:, 0]) / 1e3,
        np.median(a[:, 1]), E / np.median(a[:, 1]) / 1e3, " ".join("%.3f" % x for x in a[:, 0])))
