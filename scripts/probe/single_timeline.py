"""Kernels and copies of the LAST single-sweep call in a rocprofv3 trace directory (prefix t), on one time axis (us)."""
import csv, sys
d = sys.argv[1].rstrip('/') + '/'
import glob
kt = glob.glob(d + '**/t_kernel_trace.csv', recursive=True)[0]
mt = glob.glob(d + '**/t_memory_copy_trace.csv', recursive=True)
k = list(csv.DictReader(open(kt)))
m = list(csv.DictReader(open(mt[0]))) if mt else []
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:50], r['Queue_Id']) for r in k]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'][12:] + ' ' + r.get('Size', ''), 'dma') for r in m]
ev.sort()
# last call = events after the last k_sort_hist start (minus the upload copies just before it)
starts = [i for i, e in enumerate(ev) if 'k_sort_hist' in e[2]]
i0 = starts[-1]
while i0 > 0 and ev[i0 - 1][0] > ev[starts[-1]][0] - 200000: i0 -= 1
t0 = ev[i0][0]
n = 0
for s, e, name, q in ev[i0:]:
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}  q{q:>4} {name}")
    n += 1
print("events", n)
