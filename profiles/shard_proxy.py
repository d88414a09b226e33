#!/usr/bin/env python3
"""Single-GPU proxy for strong scaling (no 8-GPU node is available to the build).

Rank 0's share of a G-way sharded optimizer call — snesimage_step_begin(shard_rank=0, shard_count=G) + snesimage_step_commit —
is timed for G in {1, 2, 4, 8} at a fixed total number of candidates per call.  With t(G) the time per call,
    efficiency_bound(G) = t(1) / (G * t(G))
bounds the compute side of the strong-scaling efficiency from above: it contains everything a rank does itself (the
replicated work on B and the pack, its share of the candidates, the commit) and leaves out only the RCCL min-all-reduce
of n doubles (latency-bound, ~10-20 us over xGMI).  Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--totals", default="64,4096")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--config", choices=["rgb", "perceptual", "dither"], default="rgb")
    args = ap.parse_args()
    import torch

    import snesimage_amd as S
    from snesimage_amd.synth import synth_image

    dev = torch.device("cuda", 0)
    img = S.OptimizedImage(synth_image(), 8, 15, dither=args.config == "dither", perceptual=args.config == "perceptual", device=0)
    img.initialize_tiles()
    img.recalculate_palettes()
    stream = torch.cuda.Stream(dev)
    img.set_stream(stream.cuda_stream)
    slots = S.schedule(8, 15, 480)
    out = {"config": args.config, "note": __doc__.split("\n\n")[1].replace("\n", " "), "rows": []}
    for total in [int(t) for t in args.totals.split(",")]:
        buf = torch.empty(total, dtype=torch.float64, device=dev)
        steps = args.steps if total >= 1024 else args.steps * 4
        t1 = None
        for G in (1, 2, 4, 8):
            def run(lo, hi):
                for i in range(lo, hi):
                    _, p, idx, ch, _ = slots[i % len(slots)]
                    img.step_begin(S.METHOD_RANDOM, p, idx, ch, 1, i, total, 0, G, buf.data_ptr())
                    img.step_commit(buf.data_ptr())
            run(0, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(5, 5 + steps)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / steps
            t1 = t if G == 1 else t1
            out["rows"].append({"candidates_per_call": total, "G": G, "own_candidates": (total + G - 1) // G, "ms_per_call": t * 1e3,
                                "efficiency_bound": t1 / (G * t)})
    print(json.dumps(out))
    img.close()


if __name__ == "__main__":
    main()
