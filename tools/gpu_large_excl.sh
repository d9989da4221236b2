#!/bin/bash
# A/B: the LARGE tier kernel / its dc kernel asking for a whole CU's LDS (bit 0 / bit 1 of TLC_LARGE_EXCL) or only for what they use
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
export TLC_MID_PRIO=0
for e in 3 0 1 2; do
  echo "TLC_LARGE_EXCL=$e"
  TLC_LARGE_EXCL=$e TLC_HOST_TRACE=1 timeout -k 10 120 python tools/time_async.py 40 2>&1 | grep -v "amdgpu.ids\|tlc host" || exit 1
done | tee gpurun_out/large_excl.txt
