#!/bin/bash
# round 3: the ball-list extraction -- its tests, the PD parity file, bench line, per-pair stamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_pd_parity.py tests/test_gpu_variants.py tests/test_gpu_dropins.py tests/test_gpu_sweep.py -q -m gpu > gpurun_out/pytest_extract.log 2>&1; echo "pytest rc=$?"
grep -v "^  File\|^Extension" gpurun_out/pytest_extract.log | tail -8
timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline > gpurun_out/bench_x1.json 2> gpurun_out/bench_x1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_x1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('pi_latency_ms'), d.get('kernel_ms'))
PY
if [ -n "$PAIRT" ]; then VARIANTS="0" bash tools/gpu_r3_pairtimes.sh; fi
