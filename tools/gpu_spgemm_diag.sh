#!/bin/bash
# Where the sparse feature projection's time goes: the shipped build, then rebuilds of lp_forward.hip with -DTLC_SQ_DIAG=1 (no rows:
# launch + staging of the weight slice), =2 (no staging), =3 (neither: the launch alone).  Restores the shipped library at the end.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
echo "=== shipped"; timeout -k 10 200 python tools/time_spgemm.py --densities 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  touch tlc-gnn_amd/csrc/lp_forward.hip
  make -C tlc-gnn_amd/csrc EXTRA="$v" > gpurun_out/make_diag.log 2>&1 || { echo "make failed"; tail -5 gpurun_out/make_diag.log; break; }
  echo "=== $v"; timeout -k 10 200 python tools/time_spgemm.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
