#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PAIR_TIMES=1 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
for v in ${VARIANTS:-0}; do
echo "=== TLC_X_VARIANT=$v"
TLC_X_VARIANT=$v timeout -k 10 300 python tools/pair_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pair_times_v$v.log
done
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
