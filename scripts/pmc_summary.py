"""Per-kernel mean counter values per launch from rocprofv3 --pmc output (one or more *_counter_collection.csv).

usage: python scripts/pmc_summary.py <dir-or-csv> [...] [--filter substring]
"""
import csv, glob, os, sys
from collections import defaultdict
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = ""
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    args.remove(flt)
files = []
for a in args:
    files += [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if flt and flt not in k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))
for k in sorted(acc):
    print(k[:110])
    for c in sorted(acc[k]):
        n = len(disp[k][c])
        print("    %-36s %18.1f   (launches %d)" % (c, acc[k][c] / max(n, 1), n))
