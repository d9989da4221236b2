#!/bin/bash
# kernel trace + SQ counters of the PubMed feature projection alone (tools/time_gemm_one.py) -- diagnostic
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_g1; mkdir -p gpurun_out/prof_g1
python3 tools/time_gemm_one.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g1/kt -- python3 tools/time_gemm_one.py > gpurun_out/prof_g1/kt.log 2>&1
grep -i gemm $(find gpurun_out/prof_g1/kt -name "*kernel_stats.csv" | head -1) | cut -c1-200
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_g1/pmc -- python3 tools/time_gemm_one.py > gpurun_out/prof_g1/pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_g1/pmc2 -- python3 tools/time_gemm_one.py > gpurun_out/prof_g1/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_g1/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    if "SQ_BUSY_CU_CYCLES" in d:
        d["mfma_pipe_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])
        d["wait_share"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]; d["issue_stall_share"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
        d["busy_cycles_per_cu"] = d["SQ_BUSY_CU_CYCLES"] / 256.0
    print(k, json.dumps(d, indent=1))
PY
