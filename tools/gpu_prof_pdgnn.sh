#!/bin/bash
# kernel stats of the PDGNN auxiliary (development aid)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gnn; mkdir -p gpurun_out/prof_gnn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gnn/kt -- python3 tools/time_pdgnn.py > gpurun_out/prof_gnn/kt.log 2>&1
tail -2 gpurun_out/prof_gnn/kt.log
head -14 $(find gpurun_out/prof_gnn/kt -name "*kernel_stats.csv" | head -1) | cut -c1-150
