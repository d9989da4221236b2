#!/bin/bash
# issue-slot and LDS utilisation of the PD kernels (reported beside the HBM roofline fraction, which is tiny by
# construction for cache-resident, latency-bound kernels).  Counters in their own passes, kernel-trace off (gpurun rule).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_issue; mkdir -p gpurun_out/prof_issue
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_issue/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_issue/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/prof_issue/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_issue/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_issue/c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/prof_issue/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/prof_issue/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not ("tlc_" in k or "gemm16" in k or "spmm" in k or "lp_decode" in k):
            continue
        agg[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k[:70]][r["Counter_Name"]] += 1
out = {"_note": "rocprofv3 --pmc over `bench.py --steps 3 --warmup 1 --no-sweep`, three passes (4 SQ counters each); sums over all "
                "dispatches of a kernel. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave "
                "(MI355X_MICROARCH.md). wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers), "
                "issue_stall_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active_share = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; "
                "lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    d = {c: v[c] for c in sorted(v)}
    d["dispatches"] = max(calls[k].values())
    d["wait_share"] = v.get("SQ_WAIT_ANY", 0) / wc
    d["issue_stall_share"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
    d["active_share"] = v.get("SQ_ACTIVE_INST_ANY", 0) / wc
    d["active_lds_share"] = v.get("SQ_ACTIVE_INST_LDS", 0) / wc
    d["active_valu_share"] = v.get("SQ_ACTIVE_INST_VALU", 0) / wc
    d["lds_conflict_share"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    out[k] = d
json.dump(out, open("gpurun_out/pmc_issue_now.json", "w"), indent=1)
for k, d in list(out.items())[1:12]:
    print("%-70s wait %.2f stall %.2f active %.2f (lds %.2f valu %.2f) lds-conflict %.2f" % (k, d["wait_share"], d["issue_stall_share"], d["active_share"], d["active_lds_share"], d["active_valu_share"], d["lds_conflict_share"]))
PY
