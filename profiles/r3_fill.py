#!/usr/bin/env python3
"""Copy the evidence run's artefacts (profiles/r3_final.sh -> gpurun_out/r3final) into profiles/ as r3_*.
Run from the repo root after the run; prints the figures BASELINE.md's round-3 table quotes."""
import glob
import json
import os
import shutil

src = "gpurun_out/r3final"
for c in ("rgb", "perceptual", "dither"):
    p = os.path.join(src, "pmc_" + c, "pmc.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", "r3_pmc_%s.json" % c))
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "slots_*.json")):
    shutil.copy(f, os.path.join("profiles", "r3_" + os.path.basename(f)))
shutil.copy(os.path.join(src, "shard_proxy.json"), "profiles/r3_shard_proxy.json")
if os.path.exists(os.path.join(src, "dither_perceptual.json")):
    shutil.copy(os.path.join(src, "dither_perceptual.json"), "profiles/r3_dither_perceptual.json")
for f in glob.glob(os.path.join(src, "kernel_stats_*.txt")):
    shutil.copy(f, os.path.join("profiles", "r3_rocprofv3_" + os.path.basename(f)))
for f in glob.glob(os.path.join(src, "timeline_*.txt")) + glob.glob(os.path.join(src, "steps_*.txt")) + glob.glob(os.path.join(src, "acceptance_*.txt")):
    shutil.copy(f, os.path.join("profiles", "r3_" + os.path.basename(f)))


def line(name):
    return json.loads(open(os.path.join(src, name)).read().strip().splitlines()[-1])


for n in ("bench_rgb.json", "bench_perceptual.json", "bench_dither.json", "bench_images.json"):
    d = line(n)
    r = d["roofline"]
    print(n, "%.3f M/s" % (d["value"] / 1e6), "roofline %.3f / %s" % (r["frac"], r.get("pipeline_frac")), "traffic/cand %s" % (r["traffic"] / r["candidates_per_launch"] if r.get("traffic") and r.get("candidates_per_launch") else None))
    if "reference_batch" in d:
        rb = d["reference_batch"]
        print("   reference loop: %.3f M useful/s at %.2f %% acceptance, wasted %.2f; from the k-means start %.3f M at %.1f %%" % (
            rb["value"] / 1e6, 100 * rb["acceptance"], rb["wasted_frac"], rb["from_kmeans_start"]["value"] / 1e6, 100 * rb["from_kmeans_start"]["acceptance"]))
