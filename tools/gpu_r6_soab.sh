#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_so_ab.sh 3 > gpurun_out/r06_so_ab.txt 2>&1
cat gpurun_out/r06_so_ab.txt
bash tools/gpu_pmc_insts.sh > gpurun_out/r06_pmc_insts_b.txt 2>&1; head -12 gpurun_out/r06_pmc_insts_b.txt
