#!/bin/bash
# TLC_DC_MIN_POS_SHARED (divide and conquer inside the 256-thread tier kernel) on the PubMed batch and on the strong-scaling list
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 320 200 128 1000000 320; do
  make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 EXTRA="-DTLC_DC_MIN_POS_SHARED=$v" > gpurun_out/sweep_build.log 2>&1 || { echo "build failed"; exit 1; }
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TLC_DC_MIN_POS_SHARED=$v  value %.2f M  rotating %.2f M  latency %.4f ms  strong list %.2f M images/s (%.1f ms)' % (d['value']/1e6, d['rotated_batches']['value']/1e6, d['pi_latency_ms'], d['strong_scaling']['images_per_sec']/1e6, d['strong_scaling']['seconds']*1e3))"
done | tee gpurun_out/dc_shared_sweep.txt
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 > /dev/null 2>&1
