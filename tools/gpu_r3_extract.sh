#!/bin/bash
# round 3: the ball-list extraction -- its tests, the PD parity file, A/B bench lines, per-pair stamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_pd_parity.py tests/test_gpu_variants.py tests/test_gpu_dropins.py tests/test_gpu_sweep.py -q -m gpu > gpurun_out/pytest_extract.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_extract.log
for mode in 1 0; do
TLC_EXTRACT=$mode timeout -k 10 300 python bench.py --no-sweep > gpurun_out/bench_x$mode.json 2> gpurun_out/bench_x$mode.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_x$mode.json').read().strip().splitlines()[-1])
print('extract=$mode', {k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('pi_latency_ms'), d.get('kernel_ms'))
PY
done
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PAIR_TIMES=1 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
timeout -k 10 300 python tools/pair_times.py > gpurun_out/pair_times.log 2>&1; echo "rc=$?"
cat gpurun_out/pair_times.log
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
