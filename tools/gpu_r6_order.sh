#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python tools/order_probe.py 40 2>&1 | grep -v amdgpu | tee gpurun_out/r06_order_probe.txt
