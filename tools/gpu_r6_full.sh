#!/bin/bash
# round 6: the whole GPU suite, smoke(), then the profiles of the final build
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r06_pytest_gpu.log
tail -4 gpurun_out/r06_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_r6_final.sh
