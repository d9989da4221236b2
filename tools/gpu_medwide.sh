#!/bin/bash
# the compact / wide MEDIUM configurations: tier tests, then the long list and the headline batch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/medwide
timeout -k 10 900 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_extract.py tests/test_gpu_pd_parity.py -m gpu -x -q > gpurun_out/medwide/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/medwide/tests.log
[ $rc -eq 0 ] || exit $rc
python tools/time_strong_list.py 0 0 0 2>&1 | grep chunk_pairs
python tools/ab_option.py x_chunk_div 4096 4096 30 2>&1 | grep x_chunk_div
