#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_pd_parity.py tests/test_gpu_variants.py tests/test_gpu_dropins.py -x -q -m gpu > gpurun_out/r06_pytest_b.log 2>&1
echo "rc $?" >> gpurun_out/r06_pytest_b.log; tail -5 gpurun_out/r06_pytest_b.log
