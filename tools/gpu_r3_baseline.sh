#!/bin/bash
# round 3, first call: GPU tests of the tree as it stands, a baseline bench line, and the per-phase cycle split of COUNT
# (PHASE_DEBUG build made on the box, then the production library restored)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --no-sweep > gpurun_out/bench_r3_base.json 2> gpurun_out/bench_r3_base.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r3_base.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step') if k in d}, d.get('pi_latency_ms'), d.get('kernel_ms'))
PY
cp tlc-gnn_amd/libtlcgnn_hip.so /tmp/prod.so
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 PHASE_DEBUG=1 > gpurun_out/make_dbg.log 2>&1; echo "make rc=$?"
timeout -k 10 300 python tools/phase_profile.py > gpurun_out/phase_profile.log 2>&1; echo "phase rc=$?"
grep -E "COUNT pass|early" gpurun_out/phase_profile.log
cp /tmp/prod.so tlc-gnn_amd/libtlcgnn_hip.so
