#!/bin/bash
# LDS bank-conflict share per kernel (development aid): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE cycles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_lds; mkdir -p gpurun_out/prof_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof_lds/p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_lds/log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/prof_lds/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:64]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:14]:
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%-64s active %12.0f conflict %12.0f  %.1f%%" % (k, a, c, 100 * c / max(a, 1)))
PY
