#!/bin/bash
# A/B of a compile-time variant of the library on ONE box: the shipped build, then a rebuild with EXTRA="$1" (tools/ab_option.py's
# pipelined region on the bench batch for both; option $2 with values $3 $4 is what ab_option alternates -- any harmless one).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
OPT=${2:-ball_edges}; A=${3:-1}; B=${4:-1}
echo "=== shipped build"; timeout -k 10 300 python tools/ab_option.py $OPT $A $B 30 2>&1 | grep -v amdgpu.ids | tail -2
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 EXTRA="$1" > gpurun_out/make_ab.log 2>&1; echo "make rc=$?"
echo "=== EXTRA=$1"; timeout -k 10 300 python tools/ab_option.py $OPT $A $B 30 2>&1 | grep -v amdgpu.ids | tail -2
echo "=== shipped build again"; cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so; timeout -k 10 300 python tools/ab_option.py $OPT $A $B 30 2>&1 | grep -v amdgpu.ids | tail -2
