#!/usr/bin/env python3
"""Copy the evidence run's bench lines / proxies / stats / PMC summaries from gpurun_out/r2final into profiles/ (r2_*)
and fill the placeholders of BASELINE.md's round-2 table from them.  Run from the repo root after profiles/r2_final.sh."""
import glob
import json
import os
import shutil
import sys

src = "gpurun_out/r2final"


def copy_pmc():
    for c in ("rgb", "perceptual", "dither", "images"):
        p = os.path.join(src, "pmc_" + c, "pmc.json")
        if os.path.exists(p):
            d = json.load(open(p))
            if c == "images":
                d["calls"], d["candidates_per_call"] = 8, 128 * 64
            json.dump(d, open(os.path.join("profiles", "r2_pmc_%s.json" % c), "w"), indent=1)


if "--pmc-only" in sys.argv:  # r2_final.sh: the bench lines that follow take traffic and VALU counts from these summaries
    copy_pmc()
    sys.exit(0)


def line(name):
    return json.loads(open(os.path.join(src, name)).read().strip().splitlines()[-1])


for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "shard_proxy_*.json")):
    shutil.copy(f, os.path.join("profiles", "r2_" + os.path.basename(f)))
for f in glob.glob(os.path.join(src, "kernel_stats_*.txt")):
    shutil.copy(f, os.path.join("profiles", "r2_rocprofv3_" + os.path.basename(f)))
copy_pmc()


def M(v):
    return "%.2f M" % (v / 1e6)


rgb, perc, dith, img = line("bench_rgb.json"), line("bench_perceptual.json"), line("bench_dither.json"), line("bench_images.json")
rep = {
    "R2_RGB_FRAC": "%.3f" % rgb["roofline"]["frac"], "R2_RGB_PIPE": "%.3f" % rgb["roofline"]["pipeline_frac"], "R2_RGB": M(rgb["value"]),
    "R2_CPU16": "%.0f" % rgb["cpu_baseline"]["value"], "R2_CPU1": "%.1f" % rgb["cpu_baseline"]["single_thread"],
    "R2_B1024": M(line("bench_rgb_b1024.json")["value"]), "R2_B8192": M(line("bench_rgb_b8192.json")["value"]),
    "R2_REF64": "%.3f M" % (rgb["reference_batch"]["value"] / 1e6),
    "R2_PERC_FRAC": "%.3f" % perc["roofline"]["frac"], "R2_PERC_PIPE": "%.3f" % perc["roofline"]["pipeline_frac"], "R2_PERC": M(perc["value"]),
    "R2_DITH_FRAC": "%.3f" % dith["roofline"]["frac"], "R2_DITH_PIPE": "%.3f" % dith["roofline"]["pipeline_frac"], "R2_DITH": M(dith["value"]),
    "R2_IMG256": M(line("bench_images_b256.json")["value"]), "R2_IMGPERC": M(line("bench_images_perceptual.json")["value"]),
    "R2_IMG_PIPE": "%.3f" % img["roofline"]["frac"], "R2_INIT": "%.2f" % img["config"]["init_seconds"], "R2_IMG": M(img["value"]),
}
rows = {}
for f in ("shard_proxy_rgb.json", "shard_proxy_rgb_large.json"):
    for r in json.load(open(os.path.join(src, f)))["rows"]:
        rows.setdefault(r["candidates_per_call"], {})[r["G"]] = r
proxy = ""
for n in sorted(rows):
    r = rows[n]
    proxy += "| %d | %.3f | %.3f | %.3f | %.3f | %.2f / %.2f / %.2f |\n" % (n, r[1]["ms_per_call"], r[2]["ms_per_call"], r[4]["ms_per_call"], r[8]["ms_per_call"],
                                                                       r[2]["efficiency_bound"], r[4]["efficiency_bound"], r[8]["efficiency_bound"])
rep["PROXY_ROWS\n"] = proxy
text = open("BASELINE.md").read()
for k in sorted(rep, key=len, reverse=True):
    text = text.replace(k, rep[k])
open("BASELINE.md", "w").write(text)
print({k: v for k, v in rep.items() if not k.startswith("PROXY")})
