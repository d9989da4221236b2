#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), d.get('pi_latency_ms'), d.get('kernel_ms'))
PY
timeout -k 10 120 python tools/time_async.py 30 2>&1 | grep -v amdgpu.ids
