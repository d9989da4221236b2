"""Per-kernel resource usage (VGPRs, scratch, spills, occupancy, static LDS) of one .hip file: development aid.
python tools/resource_usage.py tlc-gnn_amd/csrc/pd_pipeline.hip [filter-substring ...] [-- extra hipcc flags]"""
import re, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--"); extra = args[k + 1:]; args = args[:k]
src, filt = args[0], args[1:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in err.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", l)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1); rows[cur][k.strip()] = v.strip()
keys = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "vspill"),
        ("SGPRs Spill", "sspill"), ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds")]
for name, v in rows.items():
    if filt and not any(f in name for f in filt):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    print("%-78s %s" % (dem[:78], "  ".join("%s=%s" % (b, v.get(a, "?")) for a, b in keys)))
