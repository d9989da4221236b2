#!/bin/bash
# extraction kernel under different register caps (waves per SIMD): bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
for w in ${WPES:-1 4 5}; do
  rm -f tlc-gnn_amd/csrc/build/extract.o; make -C tlc-gnn_amd/csrc -j16 X_WPE=$w > /dev/null 2>&1
  timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline > gpurun_out/bench_w$w.json 2> gpurun_out/bench_w$w.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_w$w.json').read().strip().splitlines()[-1])
print('X_WPE=$w', round(d['value']/1e6,2), d.get('pi_latency_ms'), d.get('kernel_ms',{}).get('vicinity_count'))
PY
done
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
