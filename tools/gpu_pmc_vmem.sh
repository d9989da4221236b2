#!/bin/bash
# what the extraction waits for: vector-memory / scalar-memory instruction counts and the busy cycles of the texture-address and L1
# units per kernel (own passes, no trace domains).  Counter names that this GPU does not have are reported by rocprofv3 and skipped.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_vmem; mkdir -p gpurun_out/prof_vmem
rocprofv3 --list-avail > gpurun_out/prof_vmem/avail.txt 2>&1
run() { d=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof_vmem/$d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_vmem/$d.log 2>&1 || echo "pass $d failed"; }
run a TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/prof_vmem/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tlc_" not in k: continue
        agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k[:60]][r["Counter_Name"]] += 1
out = {}
for k, v in agg.items():
    n = max(calls[k].values())
    out[k] = {c: v[c] / n for c in sorted(v)}; out[k]["dispatches"] = n
json.dump(out, open("gpurun_out/pmc_vmem.json", "w"), indent=1)
for k in out:
    if "extract_kernel<64" in k: print(k, json.dumps(out[k]))
PY
