#!/bin/bash
# kernel-trace timeline of a few bench steps (development aid)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_tl; mkdir -p gpurun_out/prof_tl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tl/kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sweep > gpurun_out/prof_tl/kt.log 2>&1
python3 tools/timeline.py gpurun_out/prof_tl/kt 5 > gpurun_out/timeline_now.txt
head -24 gpurun_out/timeline_now.txt
