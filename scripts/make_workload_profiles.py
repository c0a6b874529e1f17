"""Turn gpurun_out/wl/<W> (scripts/collect_workload_profiles.sh) into profiles/<tag>_<W>_{bench.json,kernel_stats.csv,timeline.txt,pmc.txt}.

usage: python scripts/make_workload_profiles.py <tag> [W ...]
pmc.txt: per kernel and launch -- bytes read (FETCH_SIZE x 1024 x 2, the gfx950 calibration of profiles/r03_pmc_calibration.json) and
written (WRITE_SIZE x 1024), VALU instructions per wave, lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), VALU issue
share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, waiting share = SQ_WAIT_ANY / SQ_WAVE_CYCLES.
"""
import csv, json, re, shutil, sys
from collections import defaultdict
from pathlib import Path
root = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
wls = sys.argv[2:] or ["C2far", "C1", "C4"]


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")


def counters(d):
    acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(set))
    for f in Path(d).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f, newline="")):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]].add(row["Dispatch_Id"])
    return {k: {c: v / max(len(cnt[k][c]), 1) for c, v in d.items()} for k, d in acc.items()}


for w in wls:
    src = root / "gpurun_out" / "wl" / w
    if not src.exists():
        continue
    line = [l for l in (src / "bench.json").read_text().splitlines() if l.startswith("{")]
    if line:
        (root / "profiles" / f"{tag}_{w}_bench.json").write_text(line[-1] + "\n")
    ks = next(src.rglob("*kernel_stats.csv"), None)
    if ks:
        shutil.copy(ks, root / "profiles" / f"{tag}_{w}_kernel_stats.csv")
    if (src / "timeline.txt").exists():
        shutil.copy(src / "timeline.txt", root / "profiles" / f"{tag}_{w}_timeline.txt")
    fe, wr, sq = counters(src / "fetch"), counters(src / "write"), counters(src / "sq")
    out = [f"{w}: counters per launch (mean over the launches of `bench.py --workload {w} --steps 2 --warmup 1`), three rocprofv3 --pmc passes",
           "kernel | read MB (FETCH_SIZE x 1024 x 2) | written MB | VALU insts / wave | lane utilisation | VALU issue share | waiting share | waves"]
    tot_r = tot_w = 0.0
    for k in sorted(set(fe) | set(wr) | set(sq)):
        r = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e6
        wv = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
        tot_r += r; tot_w += wv
        s = sq.get(k, {})
        ai, tc, wc, wa, iv, nw = (s.get(x, 0.0) for x in ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_WAVES"))
        out.append(f"{k[:48]:48s} {r:9.1f} {wv:9.1f} {iv / nw if nw else 0:9.0f} {tc / (64 * ai) if ai else 0:6.2f} {ai / wc if wc else 0:6.2f} {wa / wc if wc else 0:6.2f} {nw:10.0f}")
    out.append(f"sum over the kernels (one launch each; kernels that run several times per step count once): read {tot_r:.0f} MB, written {tot_w:.0f} MB")
    (root / "profiles" / f"{tag}_{w}_pmc.txt").write_text("\n".join(out) + "\n")
    print("\n".join(out))
