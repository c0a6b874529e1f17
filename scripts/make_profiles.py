"""Turn gpurun_out/evidence (scripts/collect_profiles.sh) into the committed files under profiles/.

usage: python scripts/make_profiles.py [round-tag, default r02]
"""
import csv, json, re, shutil, subprocess, sys
from collections import defaultdict
from pathlib import Path
root = Path(__file__).resolve().parent.parent
ev = root / "gpurun_out" / "evidence"
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
prof = root / "profiles"
REGION = ("k_beams", "k_power", "k_tier")


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


# ---- bench lines --------------------------------------------------------------------------------------------------
lines = {}
for w in ("C2", "C2far", "C1", "C4", "C3", "C2fire", "C2_device_tables"):
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
# ---- HBM-side bytes per kernel (dumped by bench.py's own counter passes) + counter calibration ------------------------
for name in ("pmc_fetch_write_per_kernel.csv", "pmc_calibration.json", "pmc_calibration.txt", "pipeline_trace.txt", "pipeline_trace_packed.txt", "pipeline_timeline.txt",
             "single_sweep_timeline.txt", "packed_threads.json"):
    if (ev / name).exists():
        shutil.copy(ev / name, prof / f"{tag}_{name}")
if bench and bench.get("roofline", {}).get("traffic"):
    rl = bench["roofline"]
    rec = {"round": int(tag[1:]), "frames": 256, "workload": "C2",
           "kernel": "per-beam region of one step: k_beams* (scan pass and later tiers), k_power_plan, k_power*, k_tier_* (the region of roofline.avg_launch_ms)",
           "bytes_per_launch": rl["traffic"], "detail": rl.get("traffic_detail"),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child passes of `bench.py --steps 2 --warmup 1`, summed over the kernels "
                   "of the timed region, averaged over the 3 steps, counters x 1024 B x calibration factor (FETCH_SIZE x 2.0 on gfx950: "
                   f"r03_pmc_calibration.json, kernels of known byte counts)"}
    (prof / "hbm_traffic.json").write_text(json.dumps(rec, indent=1))
for w in ("C5",):
    f = ev / f"bench_{w}.json"
    if f.exists():
        got = [l for l in f.read_text().splitlines() if l.startswith("{")]
        if got:
            (prof / f"{tag}_bench_{w}.json").write_text(got[-1] + "\n")
# ---- SQ counters of the per-beam kernels (per launch) --------------------------------------------------------------------
sq = subprocess.run([sys.executable, str(root / "scripts" / "pmc_summary.py"), str(ev / "sq"), "--filter", "k_"],
                    capture_output=True, text=True).stdout
keep, cur = [], []
for ln in sq.splitlines() + [""]:
    if not ln.startswith(" "):
        if cur and cur[0].startswith(("void k_beams", "void k_power", "void k_rows", "void k_tier_scan_direct")):
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
