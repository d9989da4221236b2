#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_raster
timeout -k 10 300 python -m pytest tests/test_gpu_pd_parity.py -x -q -m gpu -k "raster" 2>&1 | tail -15 | tee gpurun_out/pytest_raster.log && \
timeout -k 10 200 python tools/time_raster.py 2>&1 | tee gpurun_out/time_raster.log && \
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_raster -- python3 tools/time_raster.py > gpurun_out/prof_raster.log 2>&1 && \
python - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_raster/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pi_raster" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# 23 launches per case (3 warm-up + 20 timed)
for c in range(len(d) // 23):
    seg = d[c * 23 + 3:(c + 1) * 23]
    print("case %d: kernel avg %.1f us  min %.1f us" % (c, sum(seg) / len(seg), min(seg)))
P
