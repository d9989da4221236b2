#!/bin/bash
# the tier kernels' stages compiled out from the end (-DTLC_STOP_AFTER=k, libraries tools/ab_libs/S<k>.so built in the container): what a
# pipelined batch and one batch alone cost with stages 1..k only -- two rounds over the libraries, on one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/keep.so
for rep in 1 2; do
  for k in 10 9 8 7 6 5 4 3 2 1; do
    cp tools/ab_libs/S$k.so tlc-gnn_amd/libtlcgnn_hip.so
    echo -n "stop_after=$k: "; timeout -k 10 200 python tools/ab_option.py ball_edges 1 1 30 2>&1 | grep -v amdgpu | head -1
  done
done | tee gpurun_out/r06_knockout.txt
cp /tmp/keep.so tlc-gnn_amd/libtlcgnn_hip.so
