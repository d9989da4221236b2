#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PAIR_TIMES=1 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
timeout -k 10 300 python tools/pair_times.py > gpurun_out/pair_times.log 2>&1; echo "rc=$?"
cat gpurun_out/pair_times.log
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
