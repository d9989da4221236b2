#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_r6_libs.sh 2 A B H64 H128
cp tools/ab_libs/B.so tlc-gnn_amd/libtlcgnn_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_pd_parity.py tests/test_gpu_tiers.py -x -q -m gpu 2>&1 | tail -3
