#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PHASE_DEBUG=1 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
timeout -k 10 300 python tools/dc_profile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_dc_profile.txt
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
