#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_pd_parity.py tests/test_gpu_tiers.py -x -q -m gpu 2>&1 | tail -4
timeout -k 10 400 python tools/stress_pipelined.py 600 2>&1 | grep -v amdgpu | tail -4
