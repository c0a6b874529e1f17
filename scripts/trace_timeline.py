"""Print the kernel timeline of one bench step from a rocprofv3 --kernel-trace csv (start / end / duration in us, HW queue).

usage: python scripts/trace_timeline.py <dir with b_kernel_trace.csv> [step index from the end, default 2]
"""
import csv, re, sys
from pathlib import Path
d = Path(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = next(d.rglob("*kernel_trace.csv"))
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")[:44]
idx = [i for i, r in enumerate(rows) if "k_sort_hist" in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
end = 0
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if "fillBuffer" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"]:
        continue
    end = max(end, e)
    print(f"{s:8.0f} {e:8.0f} {e - s:7.0f} q{r['Queue_Id']:>2} {short(r['Kernel_Name'])}")
print(f"step span {end:.0f} us; next step starts at {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.0f} us")
