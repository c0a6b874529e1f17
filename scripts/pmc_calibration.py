#!/usr/bin/env python3
"""Reads the two counter passes of scripts/probe/pmc_calib (FETCH_SIZE, WRITE_SIZE) and prints counter bytes / true bytes per kernel.
usage: python scripts/pmc_calibration.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

BYTES = (32 << 20) * 20
TRUE = {   # kernel -> (bytes read, bytes written)
    "cal_rows_field_read": (BYTES, BYTES // 20), "cal_rows_copy": (BYTES, BYTES),
    "cal_stream<float>": (BYTES, BYTES), "cal_stream<double>": (BYTES, BYTES), "cal_stream<HIP_vector_type<float, 4u>>": (BYTES, BYTES),
    "cal_read_only<float>": (BYTES, 0), "cal_read_only<double>": (BYTES, 0), "cal_rec64_read": (BYTES, 0)}


def load(d, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] == counter:
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                acc[k] += float(row["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k, (rd, wr) in TRUE.items():
    f, w = fetch.get(k), write.get(k)
    out[k] = {"true_read_bytes": rd, "true_written_bytes": wr, "FETCH_SIZE_x1024": None if f is None else f * 1024,
              "WRITE_SIZE_x1024": None if w is None else w * 1024,
              "fetch_ratio": None if f is None else f * 1024 / rd, "write_ratio": None if (w is None or wr == 0) else w * 1024 / wr}
    print(f"{k:44s} fetch ratio {out[k]['fetch_ratio']}  write ratio {out[k]['write_ratio']}")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
