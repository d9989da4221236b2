#!/bin/bash
# the first-layer projection: bf16x3 kernel vs the f32 MFMA kernel (TLC_GEMM_F32_ONLY=1), kernel durations from rocprofv3 --stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1; do
  rm -rf gpurun_out/prof_gemm$v
  if [ $v = 1 ]; then export TLC_GEMM_F32_ONLY=1; else unset TLC_GEMM_F32_ONLY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gemm$v -- python3 tools/time_gemm.py > gpurun_out/gemm_ab_$v.log 2>&1
  echo "TLC_GEMM_F32_ONLY=$v"; grep "M= 19717 K=  500" gpurun_out/gemm_ab_$v.log
  python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/prof_gemm$v/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Name']:
        print("   %-80s calls %5s avg %8.1f us min %8.1f" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
