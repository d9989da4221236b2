#!/bin/bash
# first GPU contact: tests of the PD path
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit" | head -4 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt
timeout 900 python -m pytest tests/test_gpu_pd_parity.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/pytest_pd.log
cat gpurun_out/pytest_pd.log
