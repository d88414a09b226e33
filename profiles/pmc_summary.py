#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes: mean / sum per counter and kernel, as JSON.

usage: pmc_summary.py OUT.json CANDIDATES_PER_LAUNCH "<command that was profiled>" DIR [DIR ...]   (each DIR = one --pmc pass, *_counter_collection.csv inside)
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
streaming read (MI355X_MICROARCH.md, HBM): `hbm_bytes_corrected` = 2*FETCH_SIZE + WRITE_SIZE, per dispatch.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, cpl, cmd, dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snesimage_amd import _ffi  # noqa: E402
version = _ffi.load().snesimage_version().decode()
res = {"command": cmd, "candidates_per_launch": cpl, "library": version, "source_hash": version.split("src:")[1].strip() if "src:" in version else None, "note": "PMC collection serialises kernels: every figure is for the kernel running alone", "kernels": {}}
for k, cs in sorted(acc.items()):
    e = {c: {"sum": v[0], "mean": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_corrected_per_dispatch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
top = sorted(res["kernels"].items(), key=lambda kv: -max((v["sum"] for c, v in kv[1].items() if isinstance(v, dict)), default=0))[:12]
for k, e in top:
    print(k[:60], {c: round(v["mean"], 3) for c, v in e.items() if isinstance(v, dict)})
