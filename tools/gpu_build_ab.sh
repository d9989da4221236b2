#!/bin/bash
# A/B of a compile-time definition on ONE box: build A, time, build B, time, twice.  bash tools/gpu_build_ab.sh "<EXTRA of A>" "<EXTRA of B>"
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$1" "$2"; do
    make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 EXTRA="$v" > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $v"; exit 1; }
    echo "EXTRA='$v'"; timeout -k 10 200 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/build_ab.txt
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 > /dev/null 2>&1
