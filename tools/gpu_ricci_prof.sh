#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_ricci
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ricci -- python3 tools/time_ricci.py > gpurun_out/prof_ricci.log 2>&1
f=$(find gpurun_out/prof_ricci -name "*kernel_stats.csv" | head -1)
head -6 "$f" | cut -c1-200
grep -v amdgpu.ids gpurun_out/prof_ricci.log | tail -4
