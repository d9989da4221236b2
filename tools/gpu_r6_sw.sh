#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 0 1; do
timeout -k 10 200 python tools/time_sweep.py 1048576 ball_bits $v 2>&1 | grep -v amdgpu.ids | head -1
done > gpurun_out/r06_sweep_bb.txt
cat gpurun_out/r06_sweep_bb.txt
