#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_gp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gp -- python3 tools/time_gemm_probe.py > gpurun_out/gemm_probe.log 2>&1
grep "M=\|rror" gpurun_out/gemm_probe.log | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_gp/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:58], r.get('Grid_Size', r.get('Grid_Size_X', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v); print("%-60s grid %-8s n=%3d  median %7.1f us  min %7.1f" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
PY
