#!/bin/bash
# MEDIUM tier kernel (256 threads) at 4 (128 VGPRs, spills) vs 3 (168 VGPRs) wavefronts per SIMD: bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
for w in ${WPES:-4 3}; do
  rm -f tlc-gnn_amd/csrc/build/pd_pipeline.o; make -C tlc-gnn_amd/csrc -j16 M_WPE=$w > /dev/null 2>&1
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline > gpurun_out/bench_m$w.json 2> gpurun_out/bench_m$w.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_m$w.json').read().strip().splitlines()[-1])
print('M_WPE=$w', round(d['value']/1e6,2), round(d['rotated_batches']['value']/1e6,2), d.get('pi_latency_ms'), d.get('kernel_ms'))
PY
  done
done
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
