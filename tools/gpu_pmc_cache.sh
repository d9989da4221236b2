#!/bin/bash
# L2 / vector-cache counters per kernel of the image batch (development aid): hit rates and request counts
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_cache; mkdir -p gpurun_out/prof_cache
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/prof_cache/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_cache/a.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d gpurun_out/prof_cache/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_cache/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/prof_cache/c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_cache/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(int)
for f in glob.glob("gpurun_out/prof_cache/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "tlc_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("TCC_HIT_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_VMEM_RD"): calls[(k, r["Counter_Name"])] += 1
out = {}
for k, v in agg.items():
    n = max([calls[(k, c)] for c in ("TCC_HIT_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_VMEM_RD")] + [1])
    out[k] = {c: v[c] / n for c in v}
    out[k]["dispatches"] = n
json.dump(out, open("gpurun_out/pmc_cache.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:12]:
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print("%-60s per dispatch: L2 req %.2e hit %.1f%% | EA rd %.2e | TCP->TCC rd %.2e wr %.2e | L1 acc %.2e | vmem rd %.2e wr %.2e salu %.2e smem %.2e" % (
        k, v.get("TCC_REQ_sum", 0), 100 * h / max(h + m, 1), v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCP_TCC_READ_REQ_sum", 0), v.get("TCP_TCC_WRITE_REQ_sum", 0),
        v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_SMEM", 0)))
PY
grep -i "error\|invalid\|not found" gpurun_out/prof_cache/*.log | head -5
