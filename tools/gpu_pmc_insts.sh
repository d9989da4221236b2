#!/bin/bash
# wave-instructions per kernel of the image batch (VALU / SALU / LDS / waves), one PMC pass over a few pipelined batches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_insts; mkdir -p gpurun_out/prof_insts
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_insts/a -- python3 tools/pipelined_region.py 4 > gpurun_out/prof_insts/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(int)
for f in glob.glob("gpurun_out/prof_insts/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
tot = collections.defaultdict(float)
print("%-66s %6s %10s %10s %10s %9s   (per dispatch, millions of wave-instructions)" % ("kernel", "calls", "VALU", "SALU", "LDS", "waves"))
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]["SQ_INSTS_VALU"] + kv[1]["SQ_INSTS_SALU"])):
    c = max(calls[k], 1)
    if "tlc_" not in k: continue
    print("%-66s %6d %10.3f %10.3f %10.3f %9.0f" % (k, c, v["SQ_INSTS_VALU"] / c / 1e6, v["SQ_INSTS_SALU"] / c / 1e6, v["SQ_INSTS_LDS"] / c / 1e6, v["SQ_WAVES"] / c))
PY
