"""Print the merged kernel / copy timeline of the last pipelined host batch in a rocprofv3 trace directory."""
import csv, sys
d = sys.argv[1].rstrip('/') + '/'
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
rows = list(csv.DictReader(open(d + 't_memory_copy_trace.csv')))
k = list(csv.DictReader(open(d + 't_kernel_trace.csv')))
t0 = min(int(r['Start_Timestamp']) for r in rows)
ev = []
for r in rows:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    ev.append((s, e, r['Direction'][12:], ''))
for r in k:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    ev.append((s, e, r['Kernel_Name'][:44], r['Queue_Id']))
ev.sort()
big = [x for x in ev if 'HOST_TO' in x[2] and x[1] - x[0] > 250000]
end = big[-1][0]
start = end - span_ms * 1e6
for s, e, n, q in ev:
    if start <= s <= end + 3e6 and (e - s > 25000 or 'HOST' in n or 'DEVICE' in n or 'sort_hist' in n or 'stats_final' in n or 'copy_link' in n):
        print(f"{(s - start) / 1e3:8.1f} {(e - start) / 1e3:8.1f} {(e - s) / 1e3:7.1f}us q{q} {n}")
