"""Turn gpurun_out/evidence (scripts/collect_profiles.sh) into the committed files under profiles/.

usage: python scripts/make_profiles.py [round-tag, default r01]
"""
import csv, json, re, shutil, sys
from collections import defaultdict
from pathlib import Path
root = Path(__file__).resolve().parent.parent
ev = root / "gpurun_out" / "evidence"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r01"
traffic_only = "--traffic-only" in sys.argv      # on the GPU box, before the default bench run reads hbm_traffic.json
prof = root / "profiles"
sys.path.insert(0, str(root))
from bench import DEFAULT_FRAMES as default_frames   # noqa: E402  (bench.py only parses arguments under __main__)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


per = defaultdict(lambda: defaultdict(list))
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    with open(ev / which / "b_counter_collection.csv", newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                per[short(row["Kernel_Name"])][counter].append(float(row["Counter_Value"]))
with open(prof / f"{tag}_pmc_fetch_write_per_kernel.csv", "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    for k in sorted(per):
        f, w = per[k]["FETCH_SIZE"], per[k]["WRITE_SIZE"]
        fh.write('"%s",%d,%.1f,%.1f\n' % (k, len(f), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))
# the timed region of roofline.avg_launch_ms: every k_beams tier, k_power and the list builders between them, per step
steps = 3   # --steps 2 --warmup 1
fetch = sum(sum(v["FETCH_SIZE"]) for k, v in per.items() if k.startswith(("k_beams", "k_power", "k_list"))) / steps
write = sum(sum(v["WRITE_SIZE"]) for k, v in per.items() if k.startswith(("k_beams", "k_power", "k_list"))) / steps
rec = {
    "round": int(tag[1:]), "frames": default_frames,
    "kernel": "per-beam kernels of one step: k_beams (all capacity tiers), k_power, k_list_* (the region of roofline.avg_launch_ms)",
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "bytes_per_launch": (fetch + write) * 1024,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`, summed over the "
            "kernels of the timed region and averaged over the 3 steps; raw counter values (KB). MI355X_MICROARCH.md: FETCH_SIZE reads "
            "exactly 1/2 of the bytes of a WIDE (16 B/lane) coalesced stream on gfx950; this kernel reads 4-byte row fields and "
            "64-byte table records, for which the counter is uncalibrated, so no correction is applied.",
}
(prof / "hbm_traffic.json").write_text(json.dumps(rec, indent=1))
if traffic_only:
    sys.exit(0)
line = [l for l in (ev / "bench_default.json").read_text().splitlines() if l.startswith("{")][-1]
bench = json.loads(line)
(prof / f"{tag}_bench_default.json").write_text(line + "\n")
shutil.copy(ev / "stats" / "b_kernel_stats.csv", prof / f"{tag}_rocprofv3_kernel_stats_bench_default.csv")
# SQ counters of the per-beam kernels (per launch)
import subprocess
sq = subprocess.run([sys.executable, str(root / "scripts" / "pmc_summary.py"), str(ev / "sq_a"), str(ev / "sq_b"), "--filter", "k_"],
                    capture_output=True, text=True).stdout
keep_blocks, cur = [], []
for ln in sq.splitlines():
    if not ln.startswith(" "):
        if cur and (cur[0].startswith("void k_beams") or cur[0].startswith("void k_power")):
            keep_blocks += cur
        cur = [ln]
    else:
        cur.append(ln)
if cur and (cur[0].startswith("void k_beams") or cur[0].startswith("void k_power")):
    keep_blocks += cur
if keep_blocks:
    (prof / f"{tag}_pmc_sq_per_beam_kernels.txt").write_text(
        "SQ counters per launch (mean over the launches of `bench.py --steps 2 --warmup 1`, 128-frame batch), two rocprofv3 --pmc passes\n"
        "VALU utilisation = SQ_ACTIVE_INST_VALU * 4 / (kernel cycles * 1024 SIMDs); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)\n"
        + "\n".join(keep_blocks) + "\n")
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step")}), bench["roofline"]["avg_launch_ms"], rec["bytes_per_launch"] / 1e9, "GB per step")
rows = list(csv.DictReader(open(ev / "stats" / "b_kernel_stats.csv")))
tot = 0.0
for r in rows:
    if "rocclr" in r["Name"] or "at::" in r["Name"]:
        continue
    ms = float(r["TotalDurationNs"]) / 1e6 / 12
    tot += ms
    print("%-50s %7.3f ms/step" % (short(r["Name"])[:50], ms))
print("sum %.3f ms/step" % tot)
