#!/usr/bin/env python3
"""Diagnosis of the rocprofv3 --pmc crash behind a slot window (round 3: SIGSEGV at the first snesimage_step_async after
snesimage_run_slots).  Same sequence as profiles/r3_slots.py, with progress markers on stderr and the process's memory map
written out before the phase that crashed, so that the frames of the tool's stack trace can be attributed to libraries.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 profiles/r4_pmc_diag.py OUTDIR [--steps-first]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mark(msg):
    sys.stderr.write("DIAG " + msg + "\n")
    sys.stderr.flush()


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    import snesimage_amd as S
    from snesimage_amd.synth import synth_image
    g = S.OptimizedImage(synth_image(), 8, 15)
    g.initialize_tiles()
    g.recalculate_palettes()
    mark("initialised")
    sched = S.schedule(8, 15, 120)
    if "--steps-first" in sys.argv:  # the large calls BEFORE any window: do they survive on their own?
        for j in range(6):
            m, p, i, ch, _ = sched[j]
            g.step_async(S.METHOD_RANDOM, p, i, 0, 5, 10 ** 7 + j, 4096)
        g.sync()
        mark("6 large calls before any window: ok")
    g.slots_reserve(64)
    g.sync()
    mark("slot contexts reserved")
    _, st, stats = g.run_slots(240, seed=1, first_step_id=0, state=(0, 0, 0, 0), window=0, want_log=False)
    g.sync()
    mark("run_slots: %d calls in %d launch sets" % (stats["calls"], stats["windows"]))
    with open("/proc/self/maps") as f, open(os.path.join(out, "maps.txt"), "w") as o:
        o.write(f.read())
    mark("maps written")
    for j in range(6):
        m, p, i, ch, _ = sched[j]
        mark("step_async %d ..." % j)
        g.step_async(S.METHOD_RANDOM, p, i, 0, 5, 10 ** 7 + j, 4096)
        g.sync()
        mark("step_async %d done" % j)
    _, st, stats = g.run_slots(240, seed=1, first_step_id=240, state=st, window=0, want_log=False)
    g.sync()
    mark("second run_slots: %d calls in %d launch sets" % (stats["calls"], stats["windows"]))
    g.close()
    mark("closed")


if __name__ == "__main__":
    main()
