#!/usr/bin/env python3
"""Copy the evidence run's artefacts (profiles/r4_final.sh -> gpurun_out/r4final) into profiles/ as r4_*.
Run from the repo root after each part; prints the figures BASELINE.md's round-4 table quotes."""
import glob
import json
import os
import shutil

src = "gpurun_out/r4final"
for c in ("rgb", "perceptual", "dither", "slots"):
    p = os.path.join(src, "pmc_" + c, "pmc.json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", "r4_pmc_%s.json" % c))
for n in ("sq4096", "sq64"):
    p = os.path.join(src, n, "sq_table.txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", "r4_wave_cycles_%s.txt" % n[2:]))
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "slots_*.json")) + glob.glob(os.path.join(src, "shard_proxy_*.json")):
    shutil.copy(f, os.path.join("profiles", "r4_" + os.path.basename(f)))
for f in glob.glob(os.path.join(src, "kernel_stats_*.txt")):
    shutil.copy(f, os.path.join("profiles", "r4_rocprofv3_" + os.path.basename(f)))
for f in glob.glob(os.path.join(src, "timeline_*.txt")) + glob.glob(os.path.join(src, "acceptance_*.txt")):
    shutil.copy(f, os.path.join("profiles", "r4_" + os.path.basename(f)))


def line(name):
    p = os.path.join(src, name)
    return json.loads(open(p).read().strip().splitlines()[-1]) if os.path.exists(p) else None


for n in ("bench_rgb.json", "bench_perceptual.json", "bench_dither.json", "bench_images.json", "bench_rgb_batch64.json"):
    d = line(n)
    if not d:
        continue
    r = d["roofline"]
    print(n, "%.3f M/s, %.4f ms/step" % (d["value"] / 1e6, d["ms_per_step"]), "roofline %.3f / %s" % (r["frac"], r.get("pipeline_frac")),
          "traffic/cand %s" % (r["traffic"] / r["candidates_per_launch"] if r.get("traffic") and r.get("candidates_per_launch") else None), d.get("stale_pmc", ""))
    if "reference_batch" in d:
        rb = d["reference_batch"]
        print("   reference loop: %.3f M useful/s at %.2f %% acceptance, wasted %.2f; from the k-means start %.3f M at %.1f %%" % (
            rb["value"] / 1e6, 100 * rb["acceptance"], rb["wasted_frac"], rb["from_kmeans_start"]["value"] / 1e6, 100 * rb["from_kmeans_start"]["acceptance"]))
    if d.get("channel_calls"):
        print("   channel calls: %.3f M/s, %.3f ms per 32-candidate call" % (d["channel_calls"]["value"] / 1e6, d["channel_calls"]["ms_per_call"]))
    if d.get("other_configs"):
        print("   other configs:", {k: round(v.get("value", 0) / 1e6, 3) for k, v in d["other_configs"].items()})
