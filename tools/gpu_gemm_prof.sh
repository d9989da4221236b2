#!/bin/bash
# kernel durations (rocprofv3 --stats) of tools/time_gemm_m.py: GPU time of the projection kernel per M, without the host's launch floor
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_gm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gm -- python3 tools/time_gemm_m.py > gpurun_out/gemm_m.log 2>&1
grep "M=" gpurun_out/gemm_m.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_gm/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']:
        d[(r['Kernel_Name'][:60], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', '?'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: kv[0]):
    v = sorted(v)
    print("%-62s grid %-8s n=%3d  median %7.1f us  min %7.1f" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
PY
