#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python tools/ab_option.py fuse_mask 0 16 40 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_ab_fuse_mid.txt || exit 1
done
cat gpurun_out/r06_ab_fuse_mid.txt
timeout -k 10 200 python tools/size_hist.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_size_hist.txt; cat gpurun_out/r06_size_hist.txt
