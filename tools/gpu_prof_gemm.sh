#!/bin/bash
# kernel trace + SQ counters of the feature GEMM (tools/time_gemm.py) -- diagnostic
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gemm; mkdir -p gpurun_out/prof_gemm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gemm/kt -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/kt.log 2>&1
grep gemm $(find gpurun_out/prof_gemm/kt -name "*kernel_stats.csv" | head -1) | cut -c1-220
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_gemm/pmc -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_gemm/pmc2 -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc", "pmc2"):
    for f in glob.glob("gpurun_out/prof_gemm/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gemm16" in r["Kernel_Name"] and "Li7E" in r["Kernel_Name"] or "gemm16_f32_kernel<7" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k)
            for c, xs in sorted(v.items()):
                print("   %-32s n=%d mean=%.4g" % (c, len(xs), sum(xs) / len(xs)))
PY
