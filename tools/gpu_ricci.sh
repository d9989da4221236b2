#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ricci.py -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/pytest_ricci.log && \
timeout -k 10 300 python tools/time_ricci.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_ricci.log
