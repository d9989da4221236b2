#!/bin/bash
# rocprofv3 summaries of the bench command (kernel trace + stats, then the PMC passes on their own)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r01}
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
cat gpurun_out/bench_$R.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep > gpurun_out/prof/kt.log 2>&1
cp $(find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
python tools/timeline.py gpurun_out/prof/kt 6 > gpurun_out/${R}_bench_timeline.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof/pmc_write.log 2>&1
mkdir -p profiles && python tools/pmc_to_traffic.py $R > gpurun_out/${R}_traffic.log 2>&1; cp profiles/pmc_traffic.json gpurun_out/${R}_pmc_traffic.json; cp profiles/${R}_pmc_raw.json gpurun_out/
head -14 gpurun_out/${R}_bench_kernel_stats.csv
