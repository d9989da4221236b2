#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py -x -q -m gpu --durations=4 2>&1 | tail -30 | tee gpurun_out/pytest_sweep.log
