#!/bin/bash
# A/B of the priority class (= hardware queue pool) of the MID / MEDIUM side streams: sync and pipelined batch times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "1 1" "2 1" "0 1" "2 0" "1 2" "2 2"; do
  set -- $cfg
  echo "MID=$1 MEDIUM=$2" 
  TLC_MID_PRIO=$1 TLC_MEDIUM_PRIO=$2 timeout -k 10 120 python tools/time_async.py 40 2>&1 | grep -v amdgpu.ids || exit 1
done | tee gpurun_out/prio_ab.txt
