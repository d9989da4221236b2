#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout -k 10 120 python tools/time_overlap.py 2 20 && timeout -k 10 120 python tools/time_overlap.py 3 21 && \
  GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python tools/time_overlap.py 2 20 && GPU_MAX_HW_QUEUES=16 timeout -k 10 120 python tools/time_overlap.py 3 21 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/overlap.log
