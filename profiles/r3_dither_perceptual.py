#!/usr/bin/env python3
"""--dither --perceptual-palettes: optimizer calls of N candidates on consecutive slots, group-sparse path (resumed runs,
B a call ahead) against the dense path (SNES_SPARSE=0).  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out = {"candidates_per_call": n, "calls": steps}
for name, env in (("dense", "0"), ("sparse", "1")):
    os.environ["SNES_SPARSE"] = env
    import snesimage_amd as S
    from snesimage_amd.synth import synth_image
    im = S.OptimizedImage(synth_image(0x5EED0000), 8, 15, dither=True, perceptual=True)
    im.initialize_tiles()
    im.recalculate_palettes()
    slots = S.schedule(8, 15, steps + 2)
    for i in range(2):
        im.step(S.METHOD_RANDOM, slots[i][1], slots[i][2], 0, 1, i, n)
    t0 = time.perf_counter()
    for i in range(2, steps + 2):
        im.step(S.METHOD_RANDOM, slots[i][1], slots[i][2], 0, 1, i, n)
    dt = time.perf_counter() - t0
    out[name] = {"candidates_per_s": n * steps / dt, "ms_per_call": 1e3 * dt / steps, "error": im.error(), "palette_crc": int(np.asarray(im.palette, np.uint64).sum())}
    im.close()
print(json.dumps(out))
