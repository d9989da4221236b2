#!/bin/bash
# round 6: second half of the pending chunk submitted from inside the wait for a workspace (option early_back) + host traces
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_option.py early_back 0 1 40 > gpurun_out/r06_ab_early_back.txt 2>&1
cat gpurun_out/r06_ab_early_back.txt
timeout -k 10 120 python tools/host_trace.py early_back 0 16 > /dev/null 2> gpurun_out/r06_host_trace0.txt
timeout -k 10 120 python tools/host_trace.py early_back 1 16 > /dev/null 2> gpurun_out/r06_host_trace1.txt
tail -20 gpurun_out/r06_host_trace0.txt
