#!/bin/bash
# round 6: MEDIUM / MID tiers with the cycle swap kept in the tier kernel (option fuse_mask) against the hand-off + swap kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 2 16 18 146; do
  echo "== fuse_mask 0 vs $v" >> gpurun_out/r06_ab_fuse.txt
  timeout -k 10 300 python tools/ab_option.py fuse_mask 0 $v 40 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_ab_fuse.txt || exit 1
done
cat gpurun_out/r06_ab_fuse.txt
