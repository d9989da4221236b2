#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_tiers.py -x -q -m gpu 2>&1 | tail -4
