#!/bin/bash
# rocprofv3 summaries of the bench command (kernel trace + stats, then PMC passes on their own)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r01}
mkdir -p gpurun_out/prof
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
cat gpurun_out/bench_$R.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/kt.log 2>&1
cp $(find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_write.log 2>&1
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for name, pat in (("FETCH_SIZE", "gpurun_out/prof/pmc_fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/prof/pmc_write/**/*counter_collection.csv")):
    files = glob.glob(pat, recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != name:
                continue
            k = row["Kernel_Name"]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(out, open("gpurun_out/%s_pmc_raw.json" % os.environ.get("R", "r01"), "w"), indent=1)
for name in out:
    for k, v in sorted(out[name].items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print(name, k[:60], v)
PY
head -12 gpurun_out/${R}_bench_kernel_stats.csv
