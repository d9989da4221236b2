#!/bin/bash
# A/B of two prebuilt libraries on ONE box: tools/ab_libs/A.so and B.so (built in the container) take turns as
# tlc-gnn_amd/libtlcgnn_hip.so; pipelined batches timed by tools/ab_option.py.  bash tools/gpu_so_ab.sh [reps [option valueA valueB]]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/keep.so
for rep in $(seq 1 ${1:-2}); do
  for v in A B; do
    cp tools/ab_libs/$v.so tlc-gnn_amd/libtlcgnn_hip.so
    echo -n "$v: "; timeout -k 10 200 python tools/ab_option.py ${2:-ball_edges} ${3:-1} ${4:-1} 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/so_ab.txt
cp /tmp/keep.so tlc-gnn_amd/libtlcgnn_hip.so
