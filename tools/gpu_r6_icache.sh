#!/bin/bash
# instruction-cache counters per kernel of a pipelined region (own pmc pass; names the GPU does not have are skipped by trying them in turn)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_ic; mkdir -p gpurun_out/prof_ic
rocprofv3 --list-avail > gpurun_out/prof_ic/avail.txt 2>&1
grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_CACHE\)[A-Z_]*" gpurun_out/prof_ic/avail.txt | sort -u > gpurun_out/r06_icache_counters.txt
cat gpurun_out/r06_icache_counters.txt
run() { d=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof_ic/$d -- python3 tools/pipelined_region.py 6 > gpurun_out/prof_ic/$d.log 2>&1 || echo "pass $d failed: $(tail -2 gpurun_out/prof_ic/$d.log)"; }
run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES
run c SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU
python3 - <<'PY' | tee gpurun_out/r06_icache.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/prof_ic/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:58]
        if "tlc_" not in k or "ball_" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
names = sorted({c for v in agg.values() for c in v})
print("counters:", names)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = max(calls[k].values())
    print("%-58s " % k + "  ".join("%s=%.3g" % (c, v[c] / n) for c in names if c in v))
PY
