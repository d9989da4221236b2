#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/keep.so
for v in A B; do cp tools/ab_libs/$v.so tlc-gnn_amd/libtlcgnn_hip.so; echo -n "$v: "; python tools/rows_hash.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r06_rows_hash.txt
cp /tmp/keep.so tlc-gnn_amd/libtlcgnn_hip.so
bash tools/gpu_r6_soab.sh
