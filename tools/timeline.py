"""Print the kernel timeline of the last PD/PI batch from a rocprofv3 kernel-trace CSV (development aid)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if ('tlc_vicinity_kernel<false, 64' in r['Kernel_Name'] or 'tlc_extract_kernel<64' in r['Kernel_Name'])]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -2   # which COUNT launch starts the window (default: second to last)
i0 = idx[k]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[max(i0 - 8, 0):idx[k + 1] + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    print("%9.1f %9.1f  %7.1f  q=%s  %s" % (s, e, e - s, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
