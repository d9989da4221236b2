#!/bin/bash
# kernel trace + SQ counters of the feature GEMM (tools/time_gemm.py) -- diagnostic
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_gemm; mkdir -p gpurun_out/prof_gemm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gemm/kt -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/kt.log 2>&1
grep gemm $(find gpurun_out/prof_gemm/kt -name "*kernel_stats.csv" | head -1) | cut -c1-220
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_gemm/pmc -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/pmc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_gemm/pmc2 -- python3 tools/time_gemm.py > gpurun_out/prof_gemm/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = {"_note": "rocprofv3 over tools/time_gemm.py: kernel-trace durations, then two --pmc passes (SQ counters summed over the chip, "
                "mean per dispatch). MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); "
                "v_mfma_f32_16x16x4_f32 holds the pipe 32 cycles, so BUSY/32 = MFMA instructions issued."}
for f in glob.glob("gpurun_out/prof_gemm/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm16" in r["Name"]:
            res.setdefault("kernel_trace", {})[r["Name"][:70]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
for d in ("pmc", "pmc2"):
    for f in glob.glob("gpurun_out/prof_gemm/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gemm16_f32_kernel<7, true" in r["Kernel_Name"]:
                acc["gemm16_f32_kernel<7,true,2,32> (mean over M=19717 K=500 and M=13752 K=768, N=100)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, xs in sorted(v.items()):
                res.setdefault(k, {})[c] = sum(xs) / len(xs)
k = "gemm16_f32_kernel<7,true,2,32> (mean over M=19717 K=500 and M=13752 K=768, N=100)"
if k in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res[k] and "SQ_BUSY_CU_CYCLES" in res[k]:
    res[k]["mfma_pipe_utilisation"] = res[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * res[k]["SQ_BUSY_CU_CYCLES"])
json.dump(res, open("gpurun_out/gemm_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
