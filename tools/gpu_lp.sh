#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lp_forward.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_lp.log && \
python bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -3 gpurun_out/bench_now.err
