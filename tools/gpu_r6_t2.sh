#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest_gpu.log; tail -4 gpurun_out/r06_pytest_gpu.log
timeout -k 10 300 python tools/ab_option.py ball_bits 1 1 40 2>&1 | grep -v amdgpu.ids
