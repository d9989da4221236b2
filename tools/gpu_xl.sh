#!/bin/bash
# lane-per-pair extraction: parity tests, phase profile, timeline, quick bench (development aid)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/xl
timeout -k 10 600 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_extract.py -m gpu -x -q -k "lane or extraction or tiny or async" > gpurun_out/xl/tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/xl/tests.log
python tools/xlane_profile.py 24 32 2>&1 | grep cut
bash tools/gpu_timeline.sh > gpurun_out/xl/tl.log 2>&1
grep -n "xlane\|extract_kernel<64\|classify\|scan_bin\|tiny" gpurun_out/timeline_now.txt | head -12
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --no-traffic > gpurun_out/xl/bench.json 2> gpurun_out/xl/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/xl/bench.json'))
print({k:d[k] for k in ('value','pi_ms_per_step','pi_latency_ms','timed_outputs_equal')}, d['rotated_batches']['value'])
PY
