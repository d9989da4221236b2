#!/bin/bash
# several prebuilt libraries (tools/ab_libs/<name>.so) in turn on one box: bash tools/gpu_r6_libs.sh <rounds> name1 name2 ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/keep.so
rounds=$1; shift
for rep in $(seq 1 $rounds); do
  for v in "$@"; do
    cp tools/ab_libs/$v.so tlc-gnn_amd/libtlcgnn_hip.so
    echo -n "$v: "; timeout -k 10 200 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/r06_libs.txt
cp /tmp/keep.so tlc-gnn_amd/libtlcgnn_hip.so
