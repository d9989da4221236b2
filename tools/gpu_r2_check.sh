#!/bin/bash
# round-2 check: GPU parity tests, smoke, the bench line, and the self-launching 2-rank rehearsal on ONE GPU (gloo)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err | cut -c1-400
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
    for k in ("value", "pi_ms_per_step", "pi_ms_per_step_median", "pi_latency_ms", "lp_ms_per_step", "lp_forward_edges_per_sec", "ms_per_step"):
        print(k, d.get(k))
    print("rotated", d.get("rotated_batches")); print("roofline", d.get("roofline")); print("strong", d.get("strong_scaling"))
    print("kernel_ms", d.get("kernel_ms")); print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --dist-backend gloo --single-device > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "2rank rc=$?"; tail -5 gpurun_out/bench_2rank.err | cut -c1-300; cut -c1-900 gpurun_out/bench_2rank.json
