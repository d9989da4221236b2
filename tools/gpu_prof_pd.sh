#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pd_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/pytest_pd.log
python tools/time_pd.py 20 > gpurun_out/time_pd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pd -- python tools/time_pd.py 5 > gpurun_out/prof_pd.log 2>&1
cat gpurun_out/pytest_pd.log gpurun_out/time_pd.log
find gpurun_out/prof_pd -name "*kernel_stats*" | head -1 | xargs cat | head -20
