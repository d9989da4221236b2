#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python tools/time_strong_list.py fuse_mask 0 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_strong_fuse.txt
timeout -k 10 400 python tools/time_strong_list.py fuse_mask 0 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_strong_fuse.txt
cat gpurun_out/r06_strong_fuse.txt
