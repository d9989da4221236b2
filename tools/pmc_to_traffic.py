"""Fold the raw rocprofv3 PMC collection (FETCH_SIZE / WRITE_SIZE, separate passes) into profiles/pmc_traffic.json:
HBM bytes per launch for the kernels bench.py names."""
import collections, csv, glob, json, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
NAMES = [("tlc_pd_tier_kernel<2048", "pd_tier_large"), ("tlc_pd_dc_kernel<2048", "pd_tier_large"), ("tlc_pd_tiny", "pd_tiny"), ("tlc_pd_tier_kernel<512", "pd_tier_medium"),
         ("tlc_pd_tier_kernel<128", "pd_tier_mid"), ("tlc_pd_swap_kernel<512", "pd_swap_medium"),
         ("tlc_pd_swap_kernel<128", "pd_swap_mid"),
         ("tlc_pd_tier_kernel<64", "pd_tier_small"), ("tlc_pd_tier_kernel<0", "pd_tier_huge"),
         ("tlc_extract_kernel<512", "vicinity_count_early"), ("tlc_extract_kernel<64", "vicinity_count"),
         ("tlc_classify", "classify"), ("tlc_vicinity_kernel<true", "vicinity_fill"), ("tlc_vicinity_kernel<false, 512", "vicinity_count_early"),
         ("tlc_vicinity_kernel<false", "vicinity_count"),
         ("tlc_scan_", "scan_bin"), ("gemm16_f32_kernel", "gemm_f32"), ("spmm_csr", "spmm_csr"), ("lp_decode", "lp_decode")]
raw = {}
vals = {}
for cname, pat in (("FETCH_SIZE", "gpurun_out/prof/pmc_fetch/**/*counter_collection.csv"),
                   ("WRITE_SIZE", "gpurun_out/prof/pmc_write/**/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != cname:
                continue
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    vals[cname] = agg
    raw[cname] = {k: {"sum_KB": sum(v), "launches": len(v), "median_KB": sorted(v)[len(v) // 2]} for k, v in agg.items()}
json.dump(raw, open("profiles/%s_pmc_raw.json" % tag, "w"), indent=1)
# batches in the PMC pass = launches of the scan kernel (exactly one per batch)
steps = max([len(v) for k, v in vals["FETCH_SIZE"].items() if "tlc_scan_bin" in k] + [1])
out = collections.defaultdict(lambda: {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches_per_step": 0})
for cname in raw:
    for k, v in raw[cname].items():
        for pat, nm in NAMES:
            if pat in k:
                if v["launches"] * 2 < steps:  # set-up only (e.g. the FILL fallback of the first batch on a fresh handle)
                    break
                # mean over the batches of the pass (a kernel launched twice per batch with different list sizes -- the two halves
                # of the MEDIUM tier -- has no meaningful median dispatch)
                per_step = max(1, int(round(v["launches"] / float(steps))))
                out[nm][cname + "_KB"] += v["sum_KB"] / float(steps)
                if cname == "FETCH_SIZE":
                    out[nm]["launches_per_step"] += per_step
                break
import datetime
res = {"_collected": "%s code (%s)" % (tag, datetime.date.today().isoformat()),
       "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 --no-sweep` "
                "(raw per-kernel sums and medians in profiles/%s_pmc_raw.json); bytes per batch = (FETCH_SIZE + WRITE_SIZE) * 1024 "
                "summed over the pass / batches in the pass (one of them is the 75 352-pair set-up batch of the decode table). "
                "gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams and is "
                "uncalibrated for the 4/8-byte gathers these kernels issue, so the read side is a lower bound; the working set (CSR 1.1 MB, "
                "arena ~35 MB) sits in L2 / Infinity Cache." % tag}
for nm, v in out.items():
    res[nm] = (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0
    res.setdefault("_detail", {})[nm] = v
json.dump(res, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
