#!/bin/bash
# strong list: size statistics, the timing of the production build, then the per-phase cycle shares with the walk's own clocks
# (PHASE_DEBUG=2 build on the box; the production library is put back afterwards)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/strong_stats.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/strong_stats.txt
python tools/time_strong_list.py 0 0 0 2>&1 | grep chunk_pairs
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PHASE_DEBUG=2 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
timeout -k 10 300 python tools/phase_profile.py "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/phase_profile2.txt
grep -A16 "tier medhi\|tier medium" gpurun_out/phase_profile2.txt
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
