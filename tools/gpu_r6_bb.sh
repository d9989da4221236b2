#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_extract.py -x -q -m gpu > gpurun_out/r06_pytest_bb.log 2>&1
echo "rc $?" >> gpurun_out/r06_pytest_bb.log; tail -4 gpurun_out/r06_pytest_bb.log
timeout -k 10 300 python tools/ab_option.py ball_bits 0 1 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_ab_ball_bits.txt
timeout -k 10 300 python tools/ab_option.py ball_bits 0 1 40 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_ab_ball_bits.txt
cat gpurun_out/r06_ab_ball_bits.txt
