#!/bin/bash
# several environment settings on ONE box (separate processes, in turn, twice): bash tools/gpu_env_ab3.sh "A=1" "B=2" ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$v : "; timeout -k 10 200 env $v python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/env_ab3.txt
