#!/usr/bin/env python3
"""Table of profiles/r4_pmc_sq.sh: per kernel, mean duration (serialised: the kernel alone on the chip) and where its waves' cycles go.
usage: r4_sq_table.py sq.json [kernel_trace.csv]"""
import csv
import json
import sys
from collections import defaultdict

d = json.load(open(sys.argv[1]))
dur = defaultdict(lambda: [0.0, 0])
if len(sys.argv) > 2 and sys.argv[2]:
    for row in csv.DictReader(open(sys.argv[2])):
        n = row["Kernel_Name"].split("(")[0]
        dur[n][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        dur[n][1] += 1
print("%-44s %5s %8s %7s | %6s %6s %6s %6s | %8s %7s %7s %6s | %8s %8s %5s" % ("kernel (alone on the chip)", "n", "us", "waves", "wait%", "stall%", "issue%", "valu%", "VALU/wv", "vmrd/wv", "vmwr/wv", "lds/wv", "fetchKB", "writeKB", "L2hit"))
rows = []
for k, e in d["kernels"].items():
    if "SQ_WAVE_CYCLES" not in e:
        continue
    g = lambda c: e.get(c, {}).get("mean", 0.0)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    wv = max(g("SQ_WAVES"), 1.0)
    us = dur[k][0] / dur[k][1] if dur[k][1] else 0.0
    hit = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    rows.append((us * e["SQ_WAVE_CYCLES"]["dispatches"], "%-44s %5d %8.1f %7.0f | %6.1f %6.1f %6.1f %6.1f | %8.0f %7.1f %7.1f %6.1f | %8.0f %8.0f %5.2f" % (
        k.replace("snes::", "")[:44], e["SQ_WAVE_CYCLES"]["dispatches"], us, wv, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
        100 * g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_INSTS_VALU") / wv, g("SQ_INSTS_VMEM_RD") / wv, g("SQ_INSTS_VMEM_WR") / wv, g("SQ_INSTS_LDS") / wv, 2 * g("FETCH_SIZE"), g("WRITE_SIZE"), hit)))
for _, line in sorted(rows, reverse=True)[:40]:
    print(line)
