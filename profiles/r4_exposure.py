#!/usr/bin/env python3
"""How far would error() move if upstream's unpinned arithmetic differed from this restatement?  (CPU only: the oracle.)

The reference's error() is ssimulacra2 0.5.1 + yuvxyb 0.4.2 + yuvxyb-math 0.1.1, none of which is on disk: the
oracle restates them and the product is held to the oracle.  This script does NOT touch the product.  It switches the oracle
between its restatement (variant 0) and what-if variants of the two places where the builder's recollection of upstream
differs from the restatement (oracle/snes_oracle.h: oracle_set_variant) and records, on the BASELINE image (synthetic
256x256, 8 x 15, k-means start) and the 64 random candidates of one optimizer call per slot:
  * the relative change of error() per candidate (max / median over the candidates),
  * whether the call's decision changes: the argmin candidate, and accept / reject against the incumbent.

    python profiles/r4_exposure.py [--slots N] > profiles/r4_unpinned_exposure.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = [
    (1, "zimg-style sRGB constants in yuvxyb's EOTF (alpha 1.0550107, beta 0.0030412825, linear below 12.92 beta)"),
    (2, "powf / cbrtf in binary32 as a fast-math crate would (exp2f(y log2f x); bit-hack seed + two Halley steps)"),
    (3, "both of the above"),
    (4, "exact powf / cbrtf moved by a pseudo-random relative amount <= 1e-6"),
]


def measure(slots, n_cand=64, seed=1):
    from oracle import oracle_py as O
    from snesimage_amd.synth import synth_image
    o = O.OracleImage(synth_image(), 8, 15, cache_source=True)
    o.initialize_tiles()
    o.recalculate_palettes()
    cands = {s: O.random_candidates(seed, 1000 + i, n_cand) for i, s in enumerate(slots)}
    base = {}
    o.set_variant(0)
    inc0 = o.error()
    for s in slots:
        base[s] = o.score_candidates(s[0], s[1], cands[s])
    rows = []
    for bits, what in VARIANTS:
        o.set_variant(bits)
        inc = o.error()
        rel_all, argmin_changed, decision_changed = [], 0, 0
        for s in slots:
            e = o.score_candidates(s[0], s[1], cands[s])
            rel_all.append(np.abs(e - base[s]) / np.abs(base[s]))
            k0, k1 = int(np.argmin(base[s])), int(np.argmin(e))
            argmin_changed += k0 != k1
            decision_changed += (k0 != k1) or ((base[s][k0] < inc0) != (e[k1] < inc))
        rel = np.concatenate(rel_all)
        rows.append({"variant": bits, "what": what, "incumbent_error_rel_change": abs(inc - inc0) / inc0,
                     "candidate_error_rel_change_max": float(rel.max()), "candidate_error_rel_change_median": float(np.median(rel)),
                     "calls": len(slots), "calls_whose_argmin_changed": int(argmin_changed), "calls_whose_decision_changed": int(decision_changed)})
    o.set_variant(0)
    assert o.error() == inc0
    return {"image": "synthetic 256x256 (seed 0x5EED0000), 8 subpalettes x 15, RGB distance, k-means start", "incumbent_error": inc0,
            "candidates_per_call": n_cand, "slots": [list(s) for s in slots], "bar": 1e-5, "rows": rows,
            "note": "oracle-only what-if variants (oracle_set_variant); the product computes variant 0 and nothing else"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=8)
    a = ap.parse_args()
    sl = [(p, (5 * p + 3) % 15) for p in range(a.slots)]
    print(json.dumps(measure(sl), indent=1))
