"""Turn gpurun_out/evidence (scripts/collect_profiles.sh) into the committed files under profiles/.

usage: python scripts/make_profiles.py [round-tag, default r02]
"""
import csv, json, re, shutil, subprocess, sys
from collections import defaultdict
from pathlib import Path
root = Path(__file__).resolve().parent.parent
ev = root / "gpurun_out" / "evidence"
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
prof = root / "profiles"
REGION = ("k_beams", "k_power", "k_tier")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


# ---- bench lines --------------------------------------------------------------------------------------------------
lines = {}
for w in ("C2", "C2far", "C1", "C4", "C3"):
    f = ev / f"bench_{w}.json"
    if f.exists():
        got = [l for l in f.read_text().splitlines() if l.startswith("{")]
        if got:
            lines[w] = json.loads(got[-1])
            (prof / f"{tag}_bench_{w}.json").write_text(got[-1] + "\n")
bench = lines.get("C2")
# ---- kernel stats + timelines ----------------------------------------------------------------------------------------
shutil.copy(ev / "stats" / "b_kernel_stats.csv", prof / f"{tag}_rocprofv3_kernel_stats_bench_default.csv")
for name in ("timeline", "timeline_serial"):
    if (ev / f"{name}.txt").exists():
        shutil.copy(ev / f"{name}.txt", prof / f"{tag}_{name}_one_step.txt")
# ---- HBM-side bytes per kernel ----------------------------------------------------------------------------------------
per = defaultdict(lambda: defaultdict(list))
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    with open(ev / which / "b_counter_collection.csv", newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                per[short(row["Kernel_Name"])][counter].append(float(row["Counter_Value"]))
steps = 3   # --steps 2 --warmup 1
with open(prof / f"{tag}_pmc_fetch_write_per_kernel.csv", "w") as fh:
    fh.write("kernel,launches_per_step,FETCH_SIZE_KB_per_step,WRITE_SIZE_KB_per_step\n")
    for k in sorted(per):
        f, w = per[k]["FETCH_SIZE"], per[k]["WRITE_SIZE"]
        fh.write('"%s",%.1f,%.1f,%.1f\n' % (k, len(f) / steps, sum(f) / steps, sum(w) / steps))
fetch = sum(sum(v["FETCH_SIZE"]) for k, v in per.items() if k.startswith(REGION)) / steps
write = sum(sum(v["WRITE_SIZE"]) for k, v in per.items() if k.startswith(REGION)) / steps
rec = {"round": int(tag[1:]), "frames": 256, "workload": "C2",
       "kernel": "per-beam region of one step: k_beams* (scan pass and later tiers), k_power_plan, k_power*, k_tier_* (the region of roofline.avg_launch_ms)",
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "bytes_per_launch": (fetch + write) * 1024,
       "whole_step_bytes": 1024 * sum(sum(v["FETCH_SIZE"]) + sum(v["WRITE_SIZE"]) for v in per.values()) / steps,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`, summed over the kernels of "
               "the timed region and averaged over the 3 steps; raw counter values (KB).  MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of the "
               "bytes of a WIDE (16 B/lane) coalesced stream on gfx950; these kernels read 4-byte row fields, 8-byte queue planes and 64-byte table "
               "records, for which the counter is uncalibrated, so no correction is applied."}
(prof / "hbm_traffic.json").write_text(json.dumps(rec, indent=1))
# ---- SQ counters of the per-beam kernels (per launch) --------------------------------------------------------------------
sq = subprocess.run([sys.executable, str(root / "scripts" / "pmc_summary.py"), str(ev / "sq_a"), str(ev / "sq_b"), "--filter", "k_"],
                    capture_output=True, text=True).stdout
keep, cur = [], []
for ln in sq.splitlines() + [""]:
    if not ln.startswith(" "):
        if cur and cur[0].startswith(("void k_beams", "void k_power")):
            keep += cur
        cur = [ln]
    else:
        cur.append(ln)
(prof / f"{tag}_pmc_sq_per_beam_kernels.txt").write_text(
    "SQ counters per launch (mean over the launches of `bench.py --steps 2 --warmup 1`, 256-frame batch), two rocprofv3 --pmc passes\n"
    "lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); SQ_*_CYCLES and SQ_ACTIVE_INST_* count quad-cycles\n" + "\n".join(keep) + "\n")
# ---- resource usage -------------------------------------------------------------------------------------------------
res = subprocess.run([sys.executable, str(root / "scripts" / "kernel_resources.py"), "snowgpu_kernels.hip", "float"], capture_output=True, text=True).stdout
(prof / f"{tag}_kernel_resource_usage.txt").write_text("hipcc -Rpass-analysis=kernel-resource-usage, gfx950, float32-row instantiations (scripts/kernel_resources.py)\n" + res)
for extra in ("stream_c5.json",):
    if (ev / extra).exists():
        shutil.copy(ev / extra, prof / f"{tag}_{extra}")
if (root / "gpurun_out" / "fullsize_parity.jsonl").exists():
    shutil.copy(root / "gpurun_out" / "fullsize_parity.jsonl", prof / f"{tag}_fullsize_parity.jsonl")
if bench:
    print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "value_pcie_inclusive")}), bench["roofline"]["avg_launch_ms"], rec["bytes_per_launch"] / 1e9, "GB per step (region)")
for w, d in lines.items():
    print(w, round(d["value"] / 1e9, 3), "G points/s", round(d["ms_per_step"], 2), "ms/step", d["config"]["beams_per_capacity_tier"])
