#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pd_parity.py -x -q -m gpu -k "${1:-power_of_two}" --durations=5 2>&1 | tail -15 | tee gpurun_out/pytest_one.log
