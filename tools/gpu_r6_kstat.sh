#!/bin/bash
# average kernel durations of single batches (alone) and of a pipelined region for option settings: bash tools/gpu_r6_kstat.sh "opt val" "opt val" ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r06_kstat.txt
for cfg in "$@"; do
  for mode in single_batches.py "pipelined_region.py 24"; do
    rm -rf gpurun_out/prof_ks; mkdir -p gpurun_out/prof_ks
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks/kt -- python3 tools/$mode $cfg > gpurun_out/prof_ks/kt.log 2>&1
    f=$(find gpurun_out/prof_ks/kt -name "*kernel_stats.csv" | head -1)
    echo "== $mode [$cfg]" >> gpurun_out/r06_kstat.txt
    grep -E "region|latency" gpurun_out/prof_ks/kt.log >> gpurun_out/r06_kstat.txt
    python3 - $f >> gpurun_out/r06_kstat.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tlc_' in r['Name'] and 'ball_' not in r['Name']:
        print("  %-62s calls %4s avg %8.1f us  min %8.1f  max %8.1f" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  done
done
cat gpurun_out/r06_kstat.txt
