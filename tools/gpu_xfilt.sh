#!/bin/bash
# extraction from the ball lists for tlc_vicinity_filtration / TLC_INCLUDE_ROOTS: the whole GPU suite, the PDGNN vicinity batch, the headline
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/xfilt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/xfilt/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/xfilt/tests.log
[ $rc -eq 0 ] || exit $rc
python tools/time_vic_batch.py 2>&1 | grep -v amdgpu | tee gpurun_out/xfilt/vic_batch.txt
python tools/ab_option.py extract 1 0 20 2>&1 | grep "extract="
