#!/bin/bash
# kernel durations of the long list (tools/time_strong_list.py): which kernel bounds it
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_sl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sl -- python3 tools/time_strong_list.py > gpurun_out/strong_list.log 2>&1
grep chunk_pairs gpurun_out/strong_list.log | head -3
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_sl/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-80s calls %5s avg %10.1f us max %10.1f  %5s%%" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
