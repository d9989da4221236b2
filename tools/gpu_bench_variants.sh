#!/bin/bash
# bench.py's image leg under variations of the timed region (kernel events on/off, K): development aid
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "" "--no-kernel-events" "--steps 40" "--steps 40 --no-kernel-events" "--sync-batches"; do
  timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline $v > gpurun_out/bench_v.json 2> gpurun_out/bench_v.err || exit 1
  python - "$v" <<PY
import json, sys
d=json.loads(open('gpurun_out/bench_v.json').read().strip().splitlines()[-1])
print("%-36s value %.2f M  pi %.4f ms/step  latency %.4f  roofline kernel %s" % (sys.argv[1], d['value']/1e6, d['pi_ms_per_step'], d['pi_latency_ms'], d['roofline'].get('kernel')))
PY
done | tee gpurun_out/bench_variants.txt
