#!/bin/bash
# parity of the extraction tests, then in-process A/B of the main pass's chunking: x_chunk_div 4096 (about one chunk per resident
# wavefront) against the listed divisors
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/chunk
timeout -k 10 600 python -m pytest tests/test_gpu_extract.py tests/test_gpu_pd_parity.py -m gpu -x -q > gpurun_out/chunk/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/chunk/tests.log
[ $rc -eq 0 ] || exit $rc
for v in "$@"; do python tools/ab_option.py x_chunk_div 4096 $v 30 2>&1 | grep x_chunk_div; done
