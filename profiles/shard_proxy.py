#!/usr/bin/env python3
"""Single-GPU proxy for strong scaling (no 8-GPU node is available to the build).

Rank 0's share of a G-way sharded optimizer call — snesimage_step_begin(shard_rank=0, shard_count=G) + snesimage_step_commit —
is timed for G in {1, 2, 4, 8} at a fixed total number of candidates per call.  With t(G) the time per call,
    efficiency_bound(G) = t(1) / (G * t(G))
bounds the compute side of the strong-scaling efficiency from above: it contains everything a rank does itself (the
replicated work on B and the pack, its share of the candidates, the commit) and leaves out only the RCCL min-all-reduce
of n doubles (latency-bound, ~10-20 us over xGMI).  Prints one JSON object.

--windows K1,K2,...: the same for the slot windows of the reference's loop (snesimage_slots_begin / _commit): a window of K
calls x 64 candidates, its calls dealt to G ranks in runs of six consecutive calls (own base images, own candidates).  Rank 0's share is
timed; a rank takes at most 64 calls of a window, so a single GPU works through K calls as K/64 windows in a row.
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
    ap.add_argument("--windows", default="", help="calls per window (64 candidates each) for the slot-window proxy, e.g. 64,128,256,512")
    ap.add_argument("--converge", type=int, default=20, help="--windows: sweeps of 4,096-candidate calls first (windows are for the regime where calls rarely accept)")
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
    if args.windows:
        sweep = S.schedule(8, 15, 120)
        for j in range(120 * args.converge):
            _, p, idx, _, _ = sweep[j % 120]
            img.step_async(S.METHOD_RANDOM, p, idx, 0, 5, 10 ** 7 + j, 4096)
        img.sync()
        img.slots_reserve(64)
        wbuf = torch.empty(1024 * 64, dtype=torch.float64, device=dev)
        out["window_rows"] = []
        for K in [int(t) for t in args.windows.split(",")]:  # (K <= 480: the random calls of steps 0..3; a window ends where the method changes)
            t1 = None
            for G in (1, 2, 4, 8):
                per_window = min(K, 64 * G)  # calls of one window over all ranks
                reps = (K + per_window - 1) // per_window  # windows a rank works through for K calls in all (the last one may be shorter)

                states = [S.schedule(8, 15, w * per_window + 1)[-1][1:] for w in range(reps)]  # every G walks the same K calls of the schedule

                def run_w(n, rank):
                    for i in range(n):
                        for w in range(reps):
                            taken, stride = img.slots_begin(min(per_window, K - w * per_window), 1, 10 ** 6 + i * 2048 + w * per_window, states[w], 0, rank, G, wbuf.data_ptr())
                            img.slots_commit(wbuf.data_ptr(), 0)
                per_rank = []
                for rank in range(G):  # the calls differ in cost (subpalettes differ in size): a window takes as long as its slowest rank
                    run_w(2, rank)
                    torch.cuda.synchronize()
                    n = max(4, 2048 // K)
                    t0 = time.perf_counter()
                    run_w(n, rank)
                    torch.cuda.synchronize()
                    per_rank.append((time.perf_counter() - t0) / n)
                t = max(per_rank)
                t1 = t if G == 1 else t1
                out["window_rows"].append({"calls_per_window_total": K, "G": G, "own_calls_per_window": (per_window + G - 1) // G, "windows_in_a_row": reps,
                                           "ms_for_K_calls_slowest_rank": t * 1e3, "ms_per_rank": [x * 1e3 for x in per_rank], "efficiency_bound": t1 / (G * t)})
    print(json.dumps(out))
    img.close()


if __name__ == "__main__":
    main()
