#!/usr/bin/env python3
"""What the per-kernel timing of bench.py's timed region costs the step: the headline's loop (4,096 random candidates per call) with
the library's event bracket on and off, interleaved.  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import snesimage_amd as S  # noqa: E402
from snesimage_amd.distributed import HipWindowScorer, sharded_step  # noqa: E402
from snesimage_amd.synth import synth_image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
device = torch.device("cuda:0")
image = S.OptimizedImage(synth_image(), 8, 15, device=0)
image.initialize_tiles()
image.recalculate_palettes()
scorer = HipWindowScorer(image, device)
slots = S.schedule(8, 15, 4000)


def run(lo, hi):
    for i in range(lo, hi):
        _, p, idx, ch, _ = slots[i % len(slots)]
        sharded_step(scorer, S.METHOD_RANDOM, p, idx, ch, 1, i, n)


run(0, 10)
out = {"candidates_per_call": n, "steps": steps, "ms_per_step": {"timing_on": [], "timing_off": []}}
at = 10
pattern = sys.argv[3] if len(sys.argv) > 3 else "101010"
out["pattern"] = pattern
for rep in range(1):
    for on in [ch == "1" for ch in pattern]:
        torch.cuda.synchronize()
        image.timing_enable(on)
        t0 = time.perf_counter()
        run(at, at + steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if on:
            image.timing_read()
        image.timing_enable(False)
        # (the same 200 slots every time: a slot's cost varies by +-5 % with the slot)
        out["ms_per_step"]["timing_on" if on else "timing_off"].append(round(1e3 * dt / steps, 4))
print(json.dumps(out))
