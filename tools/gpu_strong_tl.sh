#!/bin/bash
# kernel timeline of the long list's batches (one chunk each): where do the 26 / 33 ms runs differ
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_sl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sl -- python3 tools/time_strong_list.py 0 0 0 > gpurun_out/strong_list.log 2>&1
grep chunk_pairs gpurun_out/strong_list.log
python3 - <<'PY' > gpurun_out/strong_tl.txt
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_sl/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tlc_classify_kernel' in r['Kernel_Name']]
for a, b in zip(idx[-12:], idx[-11:] + [len(rows)]):
    t0 = int(rows[a]['Start_Timestamp'])
    print("---- batch")
    for r in rows[a:b]:
        s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
        if e - s > 300: print("%9.1f %9.1f  %8.1f  q=%s  %s" % (s, e, e - s, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
PY
tail -60 gpurun_out/strong_tl.txt
