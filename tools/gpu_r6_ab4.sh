#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python tools/sweep_options.py "" "tier_mask=0" "tier_mask=0,early_back=2" "tier_mask=0,early_back=2,n_ws=4" "fuse_mask=16" "n_ws=4" 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sweep1.txt
cat gpurun_out/r06_sweep1.txt
