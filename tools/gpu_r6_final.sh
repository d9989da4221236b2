#!/bin/bash
# round 6 profiles of the final build: bench line, kernel stats + timeline, HBM counters, instruction counts, issue counters, queue occupancy
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_profile_bench.sh r06 > gpurun_out/r06_profile.log 2>&1; echo "profile rc $?"
bash tools/gpu_pmc_insts.sh > gpurun_out/r06_pmc_insts_per_kernel.txt 2>&1; echo "insts rc $?"
bash tools/gpu_pmc_issue.sh > gpurun_out/r06_pmc_issue.log 2>&1; cp gpurun_out/pmc_issue_now.json gpurun_out/r06_pmc_issue.json; echo "issue rc $?"
bash tools/gpu_r6_tl.sh final > gpurun_out/r06_tl.log 2>&1; echo "tl rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','lp_ms_per_step','pi_latency_ms') if k in d})
print(d.get('roofline'))
PY
