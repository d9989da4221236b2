#!/bin/bash
# the two HBM-traffic counter passes of tools/gpu_profile_bench.sh alone (A/B of a store policy): gpurun_out/<tag>_pmc_traffic.json
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-ab}
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof/pmc_write.log 2>&1
python tools/pmc_to_traffic.py $R > gpurun_out/${R}_traffic.log 2>&1; cp profiles/pmc_traffic.json gpurun_out/${R}_pmc_traffic.json
cat gpurun_out/${R}_traffic.log
