"""Source lines of the scratch loads / stores (register spills) of the kernels of one .hip file: development aid.
python tools/spill_sites.py tlc-gnn_amd/csrc/pd_pipeline.hip [kernel-name-substring ...]"""
import collections, os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); filt = sys.argv[2:]
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                "-gline-tables-only", "-save-temps", "-c", src, "-o", os.path.join(d, "x.o"), "-I", os.path.dirname(src)],
               cwd=d, capture_output=True)
asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
cur, loc = None, None
hits = collections.defaultdict(collections.Counter)
for l in open(os.path.join(d, asm)):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
    if m:
        loc = int(m.group(1))
    if cur and re.match(r"\s+scratch_(load|store)", l):
        hits[cur][(loc, "ld" if "scratch_load" in l else "st")] += 1
for k, v in hits.items():
    if filt and not any(f in k for f in filt):
        continue
    print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90])
    print("   " + "  ".join("%d:%s x%d" % (ln, kind, c) for (ln, kind), c in sorted(v.items())))
