#!/usr/bin/env python3
"""Throughput of the reference's own optimizer loop (64 / 32 candidates per call, lib.rs:888-933) under speculative
multi-slot stepping (snesimage_run_slots), on the BASELINE image: from the k-means start and from a converged palette.

    python profiles/r4_slots.py [--config rgb|perceptual|dither|dither_perceptual] [--calls N] [--window W] [--converge SWEEPS]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="rgb")
    ap.add_argument("--calls", type=int, default=960)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--converge", type=int, default=2, help="sweeps of 4,096-candidate calls before the converged leg")
    ap.add_argument("--sync-every", type=int, default=0, help="converge phase: wait for the device every N calls (under rocprofv3 --pmc the host must not run ~10^5 launches ahead: DESIGN.md section 8)")
    args = ap.parse_args()
    import snesimage_amd as S
    from snesimage_amd.synth import synth_image
    flags = {"rgb": {}, "perceptual": {"perceptual": True}, "dither": {"dither": True}, "dither_perceptual": {"dither": True, "perceptual": True}}[args.config]
    g = S.OptimizedImage(synth_image(), 8, 15, **flags)
    g.initialize_tiles()
    g.recalculate_palettes()
    out = {"config": args.config, "window": args.window}

    def leg(name, first, state):
        g.slots_reserve(args.window or 64)  # untimed: the slot contexts of the largest window
        g.sync()
        t0 = time.perf_counter()
        _, st, stats = g.run_slots(args.calls, seed=1, first_step_id=first, state=state, window=args.window, want_log=False)
        g.sync()
        dt = time.perf_counter() - t0
        out[name] = {"calls": stats["calls"], "seconds": dt, "calls_per_s": stats["calls"] / dt, "useful_cand_per_s": stats["useful"] / dt,
                     "scored_cand_per_s": stats["scored"] / dt, "accepted": stats["accepted"], "acceptance": stats["accepted"] / stats["calls"],
                     "windows": stats["windows"], "wasted_frac": 1.0 - stats["useful"] / max(1, stats["scored"]), "error": g.error()}
        return st

    st = leg("from_kmeans_start", 0, (0, 0, 0, 0))
    if args.converge:
        sched = S.schedule(8, 15, 120 * args.converge)
        for j, (m, p, i, ch, _) in enumerate(sched):
            g.step_async(S.METHOD_RANDOM, p, i, 0, 5, 10 ** 7 + j, 4096)
            if args.sync_every and (j + 1) % args.sync_every == 0:
                g.sync()
        g.sync()
        leg("converged", args.calls, st)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
