#!/bin/bash
# one parity test (or a -k selection) on the GPU box: bash tools/gpu_one_test.sh <pytest args>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest -m gpu -x -q "$@" > gpurun_out/pytest_one.log 2>&1; echo "pytest rc=$?"
grep -v "^  File\|^Extension modules" gpurun_out/pytest_one.log | tail -60
