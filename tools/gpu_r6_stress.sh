#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python tools/stress_pipelined.py 3000 2>&1 | grep -v amdgpu | grep -i "rounds\|mismatch" | tee gpurun_out/r06_stress.txt
