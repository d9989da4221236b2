#!/bin/bash
# round 6, first GPU contact: the whole GPU suite, then the driver's bench command
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest_gpu.log
tail -5 gpurun_out/r06_pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench0.json 2> gpurun_out/r06_bench0.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench0.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','lp_ms_per_step') if k in d})
PY
