#!/bin/bash
# round 6: which stream a workspace's MEDIUM list takes (option med_alt)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "$@"; do
  echo "== med_alt 0 vs $v" >> gpurun_out/r06_ab_med_alt.txt
  timeout -k 10 300 python tools/ab_option.py med_alt 0 $v 40 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_ab_med_alt.txt || exit 1
done
cat gpurun_out/r06_ab_med_alt.txt
