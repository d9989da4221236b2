#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lp_forward.py tests/test_gpu_pdgnn.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_pdgnn.log && \
bash tools/gpu_prof_pdgnn.sh
