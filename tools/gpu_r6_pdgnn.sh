#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pdgnn.py tests/test_gpu_lp_forward.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06_pytest_pdgnn.log
timeout -k 10 300 python tools/time_pdgnn_layers.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_pdgnn_layers.txt
