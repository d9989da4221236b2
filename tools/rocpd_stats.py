"""Per-kernel durations (and the gaps inside a repeated launch sequence) from a rocprofv3 run's sqlite output (`*_results.db`,
the default output format of this image's rocprofv3 when --output-format is not given).  usage: rocpd_stats.py DB [substring]"""
import collections, re, sqlite3, statistics, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:90]

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
d = collections.defaultdict(list)
for n, s, e in rows:
    d[short(n)].append((e - s) / 1e3)
print("%-92s %6s %9s %9s %9s %11s" % ("kernel", "calls", "median us", "min us", "max us", "total ms"))
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    print("%-92s %6d %9.2f %9.2f %9.2f %11.3f" % (n, len(v), statistics.median(v), min(v), max(v), sum(v) / 1e3))
