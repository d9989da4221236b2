#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/xl
timeout -k 10 600 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_extract.py -m gpu -x -q -k "lane or extraction or tiny or async" > gpurun_out/xl/tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/xl/tests.log
for v in "$@"; do python tools/ab_option.py xl_cut 0 $v 30 2>&1 | grep xl_cut; done
