"""Hardware-queue occupancy and per-batch latency of a pipelined region from a rocprofv3 --kernel-trace CSV of tools/pipelined_region.py:
busy fraction of every queue with its kernels, and for every batch which kernel ended last.  python tools/queue_occupancy.py <kernel_trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:60]) for r in rows)
by = collections.defaultdict(list)
for k in ks:
    by[k[3]].append(k)
fast = by['void tlc_extract_kernel<64, true>(TlcVicParams)']
nb = len(fast)
t0, t1 = fast[-20][0], fast[-4][0]
busy = collections.defaultdict(float); names = collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, q, n in ks:
    a, b = max(s, t0), min(e, t1)
    if b > a:
        busy[q] += b - a; names[q][n] += b - a
print("window %.3f ms = 16 batches, %.4f ms per batch" % ((t1 - t0) / 1e6, (t1 - t0) / 16e6))
for q in sorted(busy, key=lambda x: -busy[x]):
    print("queue %-3s busy %.2f: " % (q, busy[q] / (t1 - t0)) + "; ".join("%s %.3f" % (n[:44], v / (t1 - t0)) for n, v in sorted(names[q].items(), key=lambda x: -x[1])[:5]))
per = [n for n in by if len(by[n]) == nb]
for b in range(nb - 20, nb - 8):
    s = fast[b][0]
    items = sorted((by[n][b][1], n, by[n][b][0], by[n][b][2]) for n in per)
    e = items[-1]
    print("batch %d latency %.3f ms; last: %s (queue %s, ran %.0f us, started +%.0f us)" % (b, (e[0] - s) / 1e6, e[1][:44], e[3], (e[0] - e[2]) / 1e3, (e[2] - s) / 1e3))
b = nb - 15
s = fast[b][0]
for n in sorted(per, key=lambda n: by[n][b][0]):
    k = by[n][b]
    print("  %-60s q%-3s start +%7.1f  dur %7.1f  end +%7.1f" % (n, k[2], (k[0] - s) / 1e3, (k[1] - k[0]) / 1e3, (k[1] - s) / 1e3))
