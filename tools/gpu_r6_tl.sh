#!/bin/bash
# kernel-trace timeline of a pipelined region with the given options: bash tools/gpu_r6_tl.sh <tag> [option value ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/prof_tl; mkdir -p gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl/kt -- python3 tools/pipelined_region.py 24 "$@" > gpurun_out/prof_tl/kt.log 2>&1
f=$(find gpurun_out/prof_tl/kt -name "*kernel_trace.csv" | head -1)
grep region gpurun_out/prof_tl/kt.log
python3 tools/queue_occupancy.py $f > gpurun_out/r06_queue_$tag.txt 2>&1
python3 - $f <<'PY' > gpurun_out/r06_tlwin_$tag.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tlc_extract_kernel<64, true>' in r['Kernel_Name']]
i0 = idx[-12]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:idx[-8] + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    print("%9.1f %9.1f  %7.1f  q=%-3s wg=%-6s %s" % (s, e, e - s, r.get('Queue_Id', '?'), r.get('Grid_Size', '?'), r['Kernel_Name'][:64]))
PY
cat gpurun_out/r06_queue_$tag.txt | head -30; head -60 gpurun_out/r06_tlwin_$tag.txt
