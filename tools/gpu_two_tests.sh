#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_ricci.py -x -q -m gpu -k "degenerate" 2>&1 | tail -25 | tee gpurun_out/pytest_two.log
