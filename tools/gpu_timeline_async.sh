#!/bin/bash
# kernel-trace timeline of pipelined batches (tools/time_async.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_tla; mkdir -p gpurun_out/prof_tla
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tla/kt -- python3 tools/time_async.py 12 > gpurun_out/prof_tla/kt.log 2>&1
python3 - <<'PY' > gpurun_out/timeline_async.txt
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_tla/kt/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tlc_extract_kernel<64' in r['Kernel_Name']]
# the async loops are the 2nd and 4th groups of 12: take COUNT launches 3+12+5 .. +3 (inside the first async loop)
k = 3 + 12 + 12 + 12 + 5      # (inside the SECOND async loop: the first one grows the later workspaces)
i0 = idx[k]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[max(i0 - 6, 0):idx[k + 4] + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    print("%9.1f %9.1f  %7.1f  q=%s  %s" % (s, e, e - s, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
PY
cat gpurun_out/timeline_async.txt
