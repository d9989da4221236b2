#!/bin/bash
# HUGE tier with the cycle swap's tables in LDS: parity tests, then the long list (three HUGE vicinities) with and without
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pd_parity.py tests/test_gpu_tiers.py -m gpu -x -q > gpurun_out/huge_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/huge_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_strong_tl.sh | tail -14
echo "---- TLC_HUGE_LDS=0"
TLC_HUGE_LDS=0 python tools/time_strong_list.py 0 0 2>&1 | grep chunk_pairs
