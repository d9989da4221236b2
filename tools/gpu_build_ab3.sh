#!/bin/bash
# A/B of several compile-time settings on ONE box: each built and timed in turn, twice.  bash tools/gpu_build_ab3.sh "<EXTRA 1>" "<EXTRA 2>" ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 EXTRA="$v" > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $v"; exit 1; }
    echo -n "EXTRA='$v' : "; timeout -k 10 200 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/build_ab3.txt
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 > /dev/null 2>&1
