#!/bin/bash
# A/B of an environment setting on ONE box (separate processes, alternating): bash tools/gpu_env_ab.sh "VAR=value" [reps]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in $(seq 1 ${2:-2}); do
  echo -n "(none)   : "; timeout -k 10 200 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  echo -n "$1 : "; timeout -k 10 200 env $1 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
done | tee gpurun_out/env_ab.txt
