"""Every kernel of a rocprofv3 kernel-trace CSV inside a time window of the LAST pipelined region: start, end, duration, queue.
python tools/timeline_window.py DIR [from_us] [to_us]   (times relative to the 4th-from-last main extraction's start)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tlc_extract_kernel<64, false>' in r['Kernel_Name'] or 'tlc_extract_kernel<64>' in r['Kernel_Name']]
t0 = int(rows[idx[-6]]['Start_Timestamp'])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -300.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
for r in rows:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    if e < lo or s > hi:
        continue
    print("%9.1f %9.1f  %7.1f  q=%-2s  %s" % (s, e, e - s, r.get('Queue_Id', '?'), r['Kernel_Name'][:64]))
