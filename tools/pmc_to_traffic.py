"""Fold the raw rocprofv3 PMC collection (FETCH_SIZE / WRITE_SIZE, separate passes) into HBM bytes per batch for the kernels
bench.py names.  As a script: python tools/pmc_to_traffic.py <tag> reads gpurun_out/prof/pmc_{fetch,write} and writes
profiles/pmc_traffic.json + profiles/<tag>_pmc_raw.json; bench.py imports fold() for the measurement inside its own run."""
import collections, csv, glob, json, sys

NAMES = [("tlc_pd_tier_kernel<2048", "pd_tier_large"), ("tlc_pd_dc_kernel<2048", "pd_tier_large"), ("tlc_pd_tiny", "pd_tiny"), ("tlc_pd_tier_kernel<512", "pd_tier_medium"),
         ("tlc_pd_tier_kernel<128", "pd_tier_mid"), ("tlc_pd_swap_kernel<512", "pd_swap_medium"),
         ("tlc_pd_swap_kernel<128", "pd_swap_mid"),
         ("tlc_pd_tier_kernel<64", "pd_tier_small"), ("tlc_pd_tier_kernel<0", "pd_tier_huge"),
         ("tlc_extract_kernel<512", "vicinity_count_early"), ("tlc_extract_kernel<64", "vicinity_count"),
         ("tlc_classify", "classify"), ("tlc_vicinity_kernel<true", "vicinity_fill"), ("tlc_vicinity_kernel<false, 512", "vicinity_count_early"),
         ("tlc_vicinity_kernel<false", "vicinity_count"),
         ("tlc_scan_", "scan_bin"), ("gemm16_f32_kernel", "gemm_f32"), ("gemm_bres_f32_kernel", "gemm_f32"), ("spmm_csr", "spmm_csr"),
         ("spmm_w2_kernel", "spmm_w2"), ("spmm16_kernel", "spmm16"), ("lp_decode", "lp_decode")]


def fold(fetch_dir, write_dir):
    """-> (bytes per batch by bench.py kernel name, detail, raw).  Bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 summed over the pass /
    batches in the pass (= launches of the scan kernel, one per batch)."""
    raw, vals = {}, {}
    for cname, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        agg = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != cname:
                    continue
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        vals[cname] = agg
        raw[cname] = {k: {"sum_KB": sum(v), "launches": len(v), "median_KB": sorted(v)[len(v) // 2]} for k, v in agg.items()}
    steps = max([len(v) for k, v in vals["FETCH_SIZE"].items() if "tlc_scan_bin" in k] + [1])
    out = collections.defaultdict(lambda: {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches_per_step": 0})
    for cname in raw:
        for k, v in raw[cname].items():
            for pat, nm in NAMES:
                if pat in k:
                    if v["launches"] * 2 < steps:  # set-up only (e.g. the FILL fallback of the first batch on a fresh handle)
                        break
                    # mean over the batches of the pass (a kernel launched twice per batch with different list sizes has no
                    # meaningful median dispatch)
                    per_step = max(1, int(round(v["launches"] / float(steps))))
                    out[nm][cname + "_KB"] += v["sum_KB"] / float(steps)
                    if cname == "FETCH_SIZE":
                        out[nm]["launches_per_step"] += per_step
                    break
    res = {nm: (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0 for nm, v in out.items()}
    return res, dict(out), raw


NOTE = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 --no-sweep`; bytes per batch = "
        "(FETCH_SIZE + WRITE_SIZE) * 1024 summed over the pass / batches in the pass (one of them is the 75 352-pair set-up batch of the decode "
        "table). gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams and is "
        "uncalibrated for the 4/8-byte gathers these kernels issue, so the read side is a lower bound; the working set (CSR 1.1 MB, arena "
        "~35 MB) sits in L2 / Infinity Cache.")

if __name__ == "__main__":
    import datetime
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    res, detail, raw = fold("gpurun_out/prof/pmc_fetch", "gpurun_out/prof/pmc_write")
    json.dump(raw, open("profiles/%s_pmc_raw.json" % tag, "w"), indent=1)
    doc = {"_collected": "%s code (%s)" % (tag, datetime.date.today().isoformat()), "_note": NOTE + " Raw per-kernel sums and medians: profiles/%s_pmc_raw.json." % tag}
    doc.update(res)
    doc["_detail"] = detail
    json.dump(doc, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res, indent=1))
