"""Parity of the HIP path against the CPU oracle, through the C ABI, on a real MI355X.

Bars (north_star): palette/tile indices bit-exact; SSIMULACRA2 error within 1e-5 relative.  The
kernels keep the oracle's binary32 operation order, so the tests hold the error to 1e-11 relative
(only the order of the binary64 pooling sums differs).
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_ERR = 1e-11  # far inside the 1e-5 the north star allows


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture(scope="module")
def S():
    import snesimage_amd
    return snesimage_amd


def pair(S, O, img, count, size, **kw):
    g = S.OptimizedImage(img, count, size, **kw)
    o = O.OracleImage(img, count, size, **kw)
    return g, o


def sync_state(g, o):
    g.tile_palettes = o.tile_palettes
    g.palette = o.palette
    g.palette_map = o.palette_map


# ---- deterministic math: device bits == oracle bits -------------------------------------------------
def test_device_math_bit_exact(S, O):
    rng = np.random.default_rng(11)
    x = rng.uniform(-30, 30, 20000).astype(np.float32)
    assert np.array_equal(S.debug_math(0, x), O.det_math(0, x))
    assert np.array_equal(S.debug_math(1, x), O.det_math(1, x))
    xe = np.concatenate([rng.uniform(-130, 0, 20000), [-0.0, -1e-30, -800.0]]).astype(np.float32)
    assert np.array_equal(S.debug_math(2, xe), O.det_math(2, xe))
    xc = np.concatenate([rng.uniform(0, 4, 20000), [0.0, 1e-20, 1.0, 8.0]]).astype(np.float32)
    assert np.array_equal(S.debug_math(3, xc), O.det_math(3, xc))
    ya, xa = rng.uniform(-5, 5, 20000).astype(np.float32), rng.uniform(-5, 5, 20000).astype(np.float32)
    ya[:4], xa[:4] = [0.0, 0.0, 1.0, -1.0], [1.0, -1.0, 0.0, 0.0]
    assert np.array_equal(S.debug_math(4, xa, ya), O.det_math(4, xa, ya))


def test_device_cbrt_equals_reference_sequence(S, O):
    """The device cube root (division-free evaluation + rounding-boundary guard, dmath.hpp) returns the float of the
    restated musl sequence for every argument: wide random sweep, signs, specials, and arguments whose root sits as
    close to a float rounding boundary as single precision allows (cubes of half-way points)."""
    rng = np.random.default_rng(13)
    parts = [np.exp(rng.uniform(np.log(1e-37), np.log(3e38), 1_500_000)), rng.uniform(0, 4, 1_500_000), rng.uniform(1, 8, 500_000),
             -rng.uniform(0, 50, 100_000)]
    f = rng.uniform(0.5, 2.0, 400_000).astype(np.float32)
    mid = f.astype(np.float64) + 0.5 * np.spacing(f).astype(np.float64)   # exactly between two floats
    for bump in (0.0, 1.0, -1.0, 2.0, -2.0):                                # the nearest few floats to mid^3
        c = (mid ** 3).astype(np.float32)
        parts.append(np.nextafter(c, np.float32(np.inf) if bump > 0 else -np.float32(np.inf)) if abs(bump) == 1.0 else
                     (np.nextafter(np.nextafter(c, np.float32(np.inf * np.sign(bump))), np.float32(np.inf * np.sign(bump))) if bump else c))
    parts.append(np.array([0.0, -0.0, 1e-45, 1e-39, -1e-40, 1.0, 8.0, 27.0, 3.4e38, np.inf, -np.inf], np.float64))
    x = np.concatenate([np.asarray(p, np.float64) for p in parts]).astype(np.float32)
    got, want = S.debug_math(3, x), O.det_math(3, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_device_lab_and_ciede_bit_exact(S, O):
    rng = np.random.default_rng(12)
    rgb = rng.integers(0, 256, size=(4000, 3)).astype(np.uint8)
    lab_dev = S.debug_math(6, rgb.astype(np.float32)).reshape(-1, 3)
    lab_ref = np.stack([O.srgb8_to_lab(c) for c in rgb])
    assert np.array_equal(lab_dev, lab_ref)
    a, b = lab_ref[:2000], lab_ref[2000:]
    d_dev = S.debug_math(5, a, b)
    d_ref = np.array([O.ciede2000(p, q) for p, q in zip(a, b)], np.float32)
    assert np.array_equal(d_dev, d_ref)
    grey = np.array([[50.0, 0.0, 0.0]], np.float32)  # zero-chroma branch
    assert np.array_equal(S.debug_math(5, grey, a[:1]), np.array([O.ciede2000(grey[0], a[0])], np.float32))


# ---- remap (optimize) -----------------------------------------------------------------------------
@pytest.mark.parametrize("count,size,flags", [(1, 15, {}), (8, 15, {}), (4, 7, {}), (8, 15, {"perceptual": True}),
                                              (8, 15, {"dither": True}), (2, 3, {"dither": True, "perceptual": True}),
                                              # entry-search shapes of the dither kernel: ragged tail only, one full group of 8,
                                              # group + tail, two groups, two groups + tail
                                              (4, 3, {"dither": True}), (3, 8, {"dither": True}), (2, 12, {"dither": True}),
                                              (10, 16, {"dither": True}), (2, 21, {"dither": True})])
def test_optimize_map_bit_exact(S, O, img256, img256_alpha, count, size, flags):
    for img in (img256, img256_alpha):
        g, o = pair(S, O, img, count, size, **flags)
        o.initialize_tiles()
        o.recalculate_palettes()
        sync_state(g, o)
        g.optimize()
        assert np.array_equal(g.palette_map, o.palette_map)
        assert rel(g.error(), o.error()) < REL_ERR
        g.optimize()  # idempotent
        assert np.array_equal(g.palette_map, o.palette_map)
        g.close()


# ---- k-means initialisers ---------------------------------------------------------------------------
@pytest.mark.parametrize("count,size,flags", [(1, 15, {}), (8, 15, {}), (4, 7, {"perceptual": True}), (2, 3, {"nes": True}),
                                              (2, 3, {"nes": True, "perceptual": True}), (8, 15, {"dither": True})])
def test_kmeans_initialisers_bit_exact(S, O, img256, img256_alpha, count, size, flags):
    for img in (img256, img256_alpha):
        g, o = pair(S, O, img, count, size, **flags)
        g.initialize_tiles()
        o.initialize_tiles()
        assert np.array_equal(g.tile_palettes, o.tile_palettes)
        assert np.array_equal(g.palette, o.palette)
        assert np.array_equal(g.palette_map, o.palette_map)
        g.recalculate_palettes()
        o.recalculate_palettes()
        assert np.array_equal(g.palette, o.palette)
        assert np.array_equal(g.palette_map, o.palette_map)
        assert g.as_json() == o.as_json()
        assert np.array_equal(g.as_rgba(), o.as_rgba())
        assert np.array_equal(g.palette_u16, o.palette_u16)
        g.close()


def test_kmeans_precondition_is_an_error(S, O):
    img = np.zeros((8, 256, 4), np.uint8)  # fully transparent: no points (the reference panics in cogset)
    g = S.OptimizedImage(img, 1, 3)
    with pytest.raises(S.SnesImageError) as e:
        g.initialize_tiles()
    assert e.value.code == -4
    o = O.OracleImage(img, 1, 3)
    with pytest.raises(RuntimeError):
        o.initialize_tiles()


# ---- candidate scoring ------------------------------------------------------------------------------
@pytest.mark.parametrize("flags,slot,n", [({}, (2, 3), 24), ({}, (7, 14), 9), ({"perceptual": True}, (0, 0), 6),
                                          ({"dither": True}, (5, 1), 6), ({"dither": True, "perceptual": True}, (1, 2), 3)])
def test_score_candidates_parity_with_maps(S, O, img256_alpha, flags, slot, n):
    from hipmem import DeviceArray
    g, o = pair(S, O, img256_alpha, 8, 15, **flags)
    o.initialize_tiles()
    o.recalculate_palettes()
    sync_state(g, o)
    cand = S.random_candidates(5, slot[0] * 100 + slot[1], n)
    cand[0] = o.palette[slot[0] * 15 + slot[1]]       # the incumbent colour itself
    cand[1] = o.palette[slot[0] * 15 + (slot[1] + 1) % 15]  # duplicate of a neighbouring entry: tie-break by index
    eo, mo = o.score_candidates(slot[0], slot[1], cand, want_maps=True)
    d_c = DeviceArray.from_numpy(cand)
    d_e = DeviceArray(n, np.float64, fill=0)
    d_m = DeviceArray((n, 256, 256), np.uint8, fill=0)
    g.score_candidates_device(slot[0], slot[1], d_c.ptr, n, d_e.ptr, d_m.ptr)
    g.sync()
    assert np.array_equal(d_m.numpy(), mo)                   # indices bit-exact
    assert rel(d_e.numpy(), eo) < REL_ERR
    d_r = DeviceArray((n, 256, 256), np.uint8, fill=7)       # the remap alone, in chunks that do not divide n
    g.set_chunk(4)
    g.remap_candidates_device(slot[0], slot[1], d_c.ptr, n, d_r.ptr)
    g.sync()
    g.set_chunk(1024)
    assert np.array_equal(d_r.numpy(), mo)
    assert rel(g.score_candidates(slot[0], slot[1], cand), eo) < REL_ERR
    assert rel(eo[0], o.error()) == 0.0 and rel(g.score_candidates(slot[0], slot[1], cand[:1])[0], g.error()) == 0.0
    assert np.array_equal(g.palette, o.palette) and np.array_equal(g.palette_map, o.palette_map)  # state untouched
    g.close()


def test_perceptual_remap_alone_equals_the_scoring_path_maps(S, O, img256_alpha):
    """`snesimage_remap_candidates_device` with --perceptual-palettes: B's map for everyone, then CIEDE2000 win tests over the
    slot's contested pixels only — pixels ruled out by the two sure "no"s of color.hpp (lightness; lightness and a-b plane),
    the rest queued and evaluated on full waves.  Against the maps the scoring path hands out (every pixel searched over its
    subpalette by the first-generation kernel) for 700 candidates, and against the oracle for a handful."""
    from hipmem import DeviceArray
    g, o = pair(S, O, img256_alpha, 8, 15, perceptual=True)
    o.initialize_tiles()
    o.recalculate_palettes()
    sync_state(g, o)
    n = 700
    for slot in [(3, 5), (0, 0)]:
        cand = S.random_candidates(11, slot[0] * 15 + slot[1], n)
        cand[0] = o.palette[slot[0] * 15 + slot[1]]
        cand[1] = o.palette[slot[0] * 15 + (slot[1] + 1) % 15]
        d_c = DeviceArray.from_numpy(cand)
        d_e = DeviceArray(n, np.float64, fill=0)
        d_m = DeviceArray((n, 256, 256), np.uint8, fill=0)
        d_r = DeviceArray((n, 256, 256), np.uint8, fill=7)
        g.score_candidates_device(slot[0], slot[1], d_c.ptr, n, d_e.ptr, d_m.ptr)
        g.remap_candidates_device(slot[0], slot[1], d_c.ptr, n, d_r.ptr)
        g.sync()
        a, b = d_m.numpy(), d_r.numpy()
        assert np.array_equal(a, b), (slot, int(np.argmax((a != b).reshape(n, -1).any(axis=1))))
        _, mo = o.score_candidates(slot[0], slot[1], cand[:4], want_maps=True)
        assert np.array_equal(b[:4], mo)
    g.close()


def test_chunking_does_not_change_results(S, O, img256):
    g, o = pair(S, O, img256, 8, 15)
    o.initialize_tiles()
    o.recalculate_palettes()
    sync_state(g, o)
    cand = S.random_candidates(9, 9, 70)
    g.set_chunk(256)
    a = g.score_candidates(4, 4, cand)
    g.set_chunk(7)
    b = g.score_candidates(4, 4, cand)
    g.set_chunk(64)
    c = g.score_candidates(4, 4, cand)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert rel(a[:5], o.score_candidates(4, 4, cand[:5])) < REL_ERR
    assert g.score_candidates(4, 4, cand[:0]).size == 0
    with pytest.raises(S.SnesImageError):
        g.score_candidates(8, 0, cand)
    with pytest.raises(S.SnesImageError):
        g.score_candidates(0, 15, cand)
    g.close()


@pytest.mark.parametrize("flags", [{}, {"perceptual": True}, {"dither": True}])
def test_launch_groups_and_lanes_do_not_change_results(S, img256_alpha, flags):
    """A list longer than a launch group runs as several groups dealt to the lanes (and, with --dither, every lane keeps the
    map of its best candidate across its groups): errors, the committed colour and the committed palette_map must not depend
    on the group size.  400 candidates on the group-sparse path with groups of 4096, 150 and 64."""
    ref = None
    for chunk in (4096, 150, 64):
        g = S.OptimizedImage(img256_alpha, 8, 15, **flags)
        g.initialize_tiles()
        g.recalculate_palettes()
        g.set_chunk(chunk)
        cand = S.random_candidates(21, 3, 400)
        errs = g.score_candidates(5, 2, cand)
        e, best = g.step(S.METHOD_RANDOM, 5, 2, 0, 21, 3, 400)
        got = (errs, e, best, g.palette_map.copy(), g.error())
        assert got[1] == min(got[0].min(), got[1]) and got[4] == got[1]
        if ref is None:
            ref = got
        else:
            assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        g.close()


@pytest.mark.parametrize("h,count,size", [(8, 1, 3), (16, 2, 3), (32, 2, 7), (64, 4, 7), (128, 8, 15)])
def test_small_heights(S, O, h, count, size):
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0010 + h, 256, h, 0)
    g, o = pair(S, O, img, count, size)
    g.initialize_tiles()
    o.initialize_tiles()
    g.recalculate_palettes()
    o.recalculate_palettes()
    assert np.array_equal(g.palette, o.palette) and np.array_equal(g.palette_map, o.palette_map)
    cand = S.random_candidates(2, h, 5)
    assert rel(g.score_candidates(count - 1, size - 1, cand), o.score_candidates(count - 1, size - 1, cand)) < REL_ERR
    g.close()


def test_single_entry_subpalette_and_max_palette(S, O, img256):
    g, o = pair(S, O, img256, 8, 1)
    o.initialize_tiles()
    sync_state(g, o)
    cand = S.random_candidates(3, 3, 4)
    eo, mo = o.score_candidates(3, 0, cand, want_maps=True)
    assert rel(g.score_candidates(3, 0, cand), eo) < REL_ERR
    g.close()
    g, o = pair(S, O, img256, 11, 23)  # 253 colours: the largest palette the packed index allows
    o.initialize_tiles()
    o.recalculate_palettes()
    sync_state(g, o)
    g.optimize()
    assert np.array_equal(g.palette_map, o.palette_map)
    cand = S.random_candidates(3, 4, 3)
    assert rel(g.score_candidates(10, 22, cand), o.score_candidates(10, 22, cand)) < REL_ERR
    g.close()
    with pytest.raises(S.SnesImageError):
        S.OptimizedImage(img256, 16, 16)


def test_quirk_component_32(S, O, img256):
    """Q3: a k-means centre >= 252 rounds to component 32, which wraps in as_rgba."""
    g, o = pair(S, O, img256, 2, 3)
    o.initialize_tiles()
    pal = o.palette
    pal[1] = [32, 5, 32]
    o.palette = pal
    o.optimize()
    sync_state(g, o)
    g.optimize()
    assert np.array_equal(g.palette_map, o.palette_map)
    assert rel(g.error(), o.error()) < REL_ERR
    assert np.array_equal(g.palette_u16, o.palette_u16) and g.as_json() == o.as_json()
    g.close()


# ---- optimizer steps ----------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", [{}, {"dither": True}, {"perceptual": True}])
def test_step_trajectory_matches_oracle(S, O, flags):
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0002, 256, 64, 1 if flags else 0)
    g, o = pair(S, O, img, 2, 3, **flags)
    g.initialize_tiles()
    o.initialize_tiles()
    g.recalculate_palettes()
    o.recalculate_palettes()
    sched = S.schedule(2, 3, 30)
    for i in [0, 1, 2, 3, 4, 5, 24, 25, 26, 27]:  # random calls, then the first channel calls
        method, p, idx, ch, _ = sched[i]
        eg, bg = g.step(method, p, idx, ch, 1, i, 16 if method == 0 else 0)
        eo, bo = o.step(method, p, idx, ch, 1, i, 16 if method == 0 else 0)
        assert np.array_equal(bg, bo), i
        assert rel(eg, eo) < REL_ERR
        assert np.array_equal(g.palette, o.palette) and np.array_equal(g.palette_map, o.palette_map)
    assert g.as_json() == o.as_json()
    g.close()


def test_deferred_map_is_observably_eager(S, O):
    """Without dither the optimize() that closes an optimizer call (lib.rs:237) is deferred until palette_map is needed;
    every observation must still see it as done at the call: palette edits after the call, error() of the resulting
    mixed state, tile edits, several calls in a row."""
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0008, 256, 64, 1)
    g, o = pair(S, O, img, 2, 3)
    for z in (g, o):
        z.initialize_tiles()
        z.recalculate_palettes()
    for i, (p, idx) in enumerate([(0, 0), (1, 1), (0, 2)]):  # three calls, nothing read in between
        g.step_async(S.METHOD_RANDOM, p, idx, 0, 2, i, 12)
        o.step(0, p, idx, 0, 2, i, 12)
    pal = o.palette.copy(); pal[1] = (pal[1] + 7) % 32
    g.palette = pal; o.palette = pal                       # the owed map belongs to the palette before this edit
    assert np.array_equal(g.palette_map, o.palette_map)
    assert rel(g.error(), o.error()) < REL_ERR              # stored map, edited palette
    g.step_async(S.METHOD_RANDOM, 1, 0, 0, 2, 10, 12)
    o.step(0, 1, 0, 0, 2, 10, 12)
    tp = o.tile_palettes.copy(); tp[:5] = 1 - tp[:5]
    g.tile_palettes = tp; o.tile_palettes = tp
    assert np.array_equal(g.palette_map, o.palette_map) and g.as_json() == o.as_json()
    g.step_async(S.METHOD_CHANNEL, 0, 1, 2, 2, 11, 0)
    o.step(1, 0, 1, 2, 2, 11, 0)
    assert np.array_equal(g.as_rgba(), o.as_rgba()) and rel(g.error(), o.error()) < REL_ERR
    g.close()


def test_nes_steps(S, O):
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0003, 256, 64)
    g, o = pair(S, O, img, 2, 3, nes=True)
    g.initialize_tiles()
    o.initialize_tiles()
    g.recalculate_palettes()
    o.recalculate_palettes()
    for i, (p, idx) in enumerate([(0, 0), (1, 2), (0, 1)]):
        eg, bg = g.step(S.METHOD_NES, p, idx, 0, 1, i)
        eo, bo = o.step(2, p, idx, 0, 1, i)
        assert np.array_equal(bg, bo) and rel(eg, eo) < REL_ERR
    assert np.array_equal(g.palette, o.palette)
    g.close()


def test_split_phase_step_equals_plain_step(S, O, img256):
    """step_begin / (min-reduce) / step_commit with 2 shards on one GPU == step()."""
    from hipmem import DeviceArray
    ref = S.OptimizedImage(img256, 8, 15)
    ref.initialize_tiles()
    ref.recalculate_palettes()
    shards = []
    for _ in range(2):
        s = S.OptimizedImage(img256, 8, 15)
        s.tile_palettes = ref.tile_palettes
        s.palette = ref.palette
        s.optimize()
        shards.append(s)
    for i, (p, idx) in enumerate([(0, 0), (3, 7), (0, 0)]):
        e_ref, b_ref = ref.step(S.METHOD_RANDOM, p, idx, 0, 4, i, 40)
        bufs = [DeviceArray(40, np.float64, fill=0) for _ in range(2)]
        for r, s in enumerate(shards):
            s.step_begin(S.METHOD_RANDOM, p, idx, 0, 4, i, 40, r, 2, bufs[r].ptr)
            s.sync()
        h0, h1 = bufs[0].numpy(), bufs[1].numpy()
        assert np.isinf(h0[1::2]).all() and np.isinf(h1[0::2]).all()
        red = DeviceArray.from_numpy(np.minimum(h0, h1))  # what the RCCL min-all-reduce produces
        for s in shards:
            s.step_commit(red.ptr)
            e, b, _ = s.last_step()
            assert e == e_ref and np.array_equal(b, b_ref)
            assert np.array_equal(s.palette, ref.palette) and np.array_equal(s.palette_map, ref.palette_map)
    for s in shards:
        s.close()
    ref.close()


def test_dither_commit_takes_winner_map(S, O):
    """With dither the commit adopts the winning candidate's map (kept per launch lane) instead of dithering again;
    more chunks than lanes, a ragged last chunk, and a winner scored on another shard (fallback: dither again)."""
    from hipmem import DeviceArray
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0007, 256, 32, 1)
    g, o = pair(S, O, img, 2, 15, dither=True)
    o.initialize_tiles()
    o.recalculate_palettes()
    sync_state(g, o)
    g.optimize()
    o.optimize()
    g.set_chunk(4)
    for i, (p, idx) in enumerate([(0, 0), (1, 7), (0, 0), (1, 14)]):
        eg, bg = g.step(S.METHOD_RANDOM, p, idx, 0, 9, i, 18)
        eo, bo = o.step(0, p, idx, 0, 9, i, 18)
        assert np.array_equal(bg, bo) and rel(eg, eo) < REL_ERR
        assert np.array_equal(g.palette_map, o.palette_map), i
    # a palette set behind the optimizer's back: the stored map is stale, so an unchanged step must dither again
    pal = o.palette.copy(); pal[3] = (pal[3] + 5) % 32
    g.palette = pal; o.palette = pal
    eg, bg = g.step(S.METHOD_CHANNEL, 0, 5, 1, 9, 100, 0)
    eo, bo = o.step(1, 0, 5, 1, 9, 100, 0)
    assert np.array_equal(bg, bo) and np.array_equal(g.palette_map, o.palette_map)
    # two shards of one step: each commits the same winner, only one of them scored it
    shards = []
    for _ in range(2):
        s = S.OptimizedImage(img, 2, 15, dither=True)
        s.tile_palettes = g.tile_palettes
        s.palette = g.palette
        s.optimize()
        shards.append(s)
    for i, (p, idx) in enumerate([(1, 3), (0, 9)]):
        e_ref, b_ref = g.step(S.METHOD_RANDOM, p, idx, 0, 11, 200 + i, 10)
        bufs = [DeviceArray(10, np.float64, fill=0) for _ in range(2)]
        for r, s in enumerate(shards):
            s.step_begin(S.METHOD_RANDOM, p, idx, 0, 11, 200 + i, 10, r, 2, bufs[r].ptr)
            s.sync()
        red = DeviceArray.from_numpy(np.minimum(bufs[0].numpy(), bufs[1].numpy()))
        for s in shards:
            s.step_commit(red.ptr)
            e, b, _ = s.last_step()
            assert e == e_ref and np.array_equal(b, b_ref)
            assert np.array_equal(s.palette_map, g.palette_map)
    for s in shards:
        s.close()
    g.close()


# ---- size-independent properties at BASELINE's full size ---------------------------------------------
@pytest.mark.parametrize("flags", [{}, {"dither": True}])
def test_failed_workspace_grow_leaves_context_usable(S, img256, flags):
    """a hipMalloc that fails while a grow-on-demand workspace is being enlarged returns SNES_ERR_HIP and must not
    leave the old capacity behind (the next, smaller call would then launch on freed/null buffers).  Every allocation of a
    grow is failed in turn; after each failure a small call has to give the same errors as before."""
    from snesimage_amd import _ffi
    L = _ffi.load()
    g = S.OptimizedImage(img256, 8, 15, **flags)
    g.initialize_tiles()
    g.recalculate_palettes()
    small = S.random_candidates(5, 1, 96)
    want = g.score_candidates(2, 4, small)
    big_n = 700
    failures = 0
    for nth in range(0, 40):
        big = S.random_candidates(5, 100 + nth, big_n)
        L.snesimage_debug_fail_alloc(nth)
        try:
            g.score_candidates(2, 4, big)
            grew = True
        except S.SnesImageError as e:
            assert e.code == -2
            grew = False
            failures += 1
        finally:
            L.snesimage_debug_fail_alloc(-1)
        got = g.score_candidates(2, 4, small)  # shrinks the request after the failed grow
        assert np.array_equal(got, want)
        if grew:
            break
        big_n += 64  # the retry above re-allocated at the small size: ask for more again
    assert failures >= 3  # several distinct allocations of the grow were exercised
    g.close()


def test_exact_reconstruction_scores_zero(S):
    rng = np.random.default_rng(21)
    pal5 = rng.integers(0, 32, size=(8, 15, 3)).astype(np.uint8)
    tp = rng.integers(0, 8, size=1024).astype(np.uint8)
    idx = rng.integers(0, 15, size=(256, 256))
    img = np.zeros((256, 256, 4), np.uint8)
    sub = tp.reshape(32, 32).repeat(8, 0).repeat(8, 1)
    c5 = pal5[sub, idx].astype(np.uint16)
    img[..., :3] = (c5 * 8 + c5 // 4).astype(np.uint8)
    img[..., 3] = 255
    g = S.OptimizedImage(img, 8, 15)
    g.tile_palettes = tp
    g.palette = pal5.reshape(-1, 3)
    g.optimize()
    assert g.error() == 0.0                       # 100 - 100 exactly
    assert np.array_equal(g.as_rgba(), img)
    e = g.score_candidates(0, 0, [[(int(pal5[0, 0, 0]) + 9) % 32, 3, 3]])
    assert e[0] >= 0.0
    g.close()


def test_large_batch_is_consistent_and_monotone_under_commit(S, img256):
    g = S.OptimizedImage(img256, 8, 15)
    g.initialize_tiles()
    g.recalculate_palettes()
    e0 = g.error()
    cand = S.random_candidates(77, 1, 2048)
    errs = g.score_candidates(6, 2, cand)
    assert np.isfinite(errs).all() and (errs >= 0).all()
    # duplicates in the candidate list score identically (checksum-of-duplicates property)
    _, first, inv = np.unique(cand, axis=0, return_index=True, return_inverse=True)
    assert np.array_equal(errs, errs[first][inv.reshape(-1)])
    e1, best = g.step(S.METHOD_RANDOM, 6, 2, 0, 77, 1, 2048)
    assert e1 == min(e0, errs.min()) and (e1 == e0 or np.array_equal(best, cand[int(np.argmin(errs))]))
    assert g.error() == e1
    g.close()


# ---- golden fixtures ---------------------------------------------------------------------------------
def _golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _golden(), ids=lambda c: c["name"])
def test_gpu_matches_golden(S, case):
    from snesimage_amd.synth import synth_image
    from hipmem import DeviceArray
    g_ = case
    img = synth_image(g_["seed"], 256, g_["h"], g_["variant"])
    g = S.OptimizedImage(img, g_["count"], g_["size"], dither=g_["dither"], perceptual=g_["perceptual"], nes=g_["nes"])
    g.initialize_tiles()
    assert g.palette.reshape(-1).tolist() == g_["init_palette"]
    assert hashlib.sha256(g.tile_palettes.tobytes()).hexdigest() == g_["init_tile_palettes_sha"]
    assert hashlib.sha256(g.palette_map.tobytes()).hexdigest() == g_["init_map_sha"]
    g.recalculate_palettes()
    assert g.palette.reshape(-1).tolist() == g_["palette"]
    assert hashlib.sha256(g.palette_map.tobytes()).hexdigest() == g_["map_sha"]
    assert rel(g.error(), float.fromhex(g_["error_hex"])) < REL_ERR
    cand = S.random_candidates(1, 42, g_["ncand"])
    n = g_["ncand"]
    d_c = DeviceArray.from_numpy(cand)
    d_e = DeviceArray(n, np.float64, fill=0)
    d_m = DeviceArray((n, g_["h"], 256), np.uint8, fill=0)
    g.score_candidates_device(g_["slot"][0], g_["slot"][1], d_c.ptr, n, d_e.ptr, d_m.ptr)
    g.sync()
    assert hashlib.sha256(d_m.numpy().tobytes()).hexdigest() == g_["cand_maps_sha"]
    assert rel(d_e.numpy(), [float.fromhex(h) for h in g_["cand_errors_hex"]]) < REL_ERR
    err, best = g.step(2 if g_["nes"] else 0, g_["slot"][0], g_["slot"][1], 0, 1, 7)
    assert best.tolist() == g_["step_best"] and rel(err, float.fromhex(g_["step_error_hex"])) < REL_ERR
    assert hashlib.sha256(g.as_json().encode()).hexdigest() == g_["json_sha"]
    g.close()


def test_short_lists_wait_for_the_base_image(S, img256_alpha, monkeypatch):
    """A 64-candidate list reaches its H pass before B's H sweep (on the side stream) has left the block checkpoints the
    candidates resume from: the launch must wait for them.  Many slots, one short list each, against the dense path."""
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img256_alpha, 8, 15)
    monkeypatch.setenv("SNES_SPARSE", "1")
    sparse = S.OptimizedImage(img256_alpha, 8, 15)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes = dense.tile_palettes
    sparse.palette = dense.palette
    sparse.optimize()
    for i in range(40):
        sp, si = i % 8, (7 * i) % 15
        cand = S.random_candidates(100 + i, sp * 15 + si, 64)
        assert np.array_equal(dense.score_candidates(sp, si, cand), sparse.score_candidates(sp, si, cand)), (sp, si)
    dense.close()
    sparse.close()


# ---- row-sparse path == dense path, bit for bit --------------------------------------------------------
@pytest.mark.parametrize("seed,variant,perceptual", [(0x5EED0000, 0, False), (0x5EED0001, 1, False), (0x5EED0007, 0, False),
                                                     (0x5EED0000, 0, True), (0x5EED0001, 1, True)])
def test_sparse_path_equals_dense_path(S, O, seed, variant, perceptual, monkeypatch):
    from snesimage_amd.synth import synth_image
    img = synth_image(seed, 256, 256, variant)
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img, 8, 15, perceptual=perceptual)
    monkeypatch.setenv("SNES_SPARSE", "1")
    monkeypatch.setenv("SNES_SPARSE_MIN", "1")
    sparse = S.OptimizedImage(img, 8, 15, perceptual=perceptual)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes = dense.tile_palettes
    sparse.palette = dense.palette
    sparse.optimize()
    pal = dense.palette
    for slot in [(2, 3), (0, 0), (7, 14)]:
        cand = S.random_candidates(seed, slot[0] * 15 + slot[1], 600 if not perceptual else 200)
        cand[0] = pal[slot[0] * 15 + slot[1]]             # the incumbent colour: wins every pixel that used the slot
        cand[1] = pal[slot[0] * 15 + (slot[1] + 1) % 15]  # duplicate of a neighbour
        cand[2] = [0, 0, 0]
        cand[3] = [31, 31, 31]
        for chunk in (512, 100):
            dense.set_chunk(chunk)
            sparse.set_chunk(chunk)
            ed = dense.score_candidates(slot[0], slot[1], cand)
            es = sparse.score_candidates(slot[0], slot[1], cand)
            assert np.array_equal(ed, es), (slot, chunk, float(np.max(np.abs(ed - es))))
        eo = O_score = None
    # and against the oracle for a handful
    o = O.OracleImage(img, 8, 15, perceptual=perceptual)
    o.tile_palettes = dense.tile_palettes
    o.palette = dense.palette
    o.optimize()
    assert rel(es[:4], o.score_candidates(7, 14, cand[:4])) < REL_ERR
    # a full optimizer call through the sparse path
    e_d, b_d = dense.step(S.METHOD_RANDOM, 3, 3, 0, 9, 1, 400)
    e_s, b_s = sparse.step(S.METHOD_RANDOM, 3, 3, 0, 9, 1, 400)
    assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map)
    dense.close()
    sparse.close()


def test_lane_storage_grows_with_the_lanes_in_use(S, monkeypatch):
    """Candidate storage of the group-sparse path is allocated for the launch lanes a list is dealt to (sparse_alloc): a list that
    fits one launch group leaves the second lane without planes; the first longer list grows the storage — same capacity per lane,
    B's planes behind the last lane — and scores as the dense path does, as do the lists after it."""
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0003, 256, 256, 0)
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img, 8, 15)
    monkeypatch.setenv("SNES_SPARSE", "1")
    monkeypatch.setenv("SNES_SPARSE_MIN", "1")
    sparse = S.OptimizedImage(img, 8, 15)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes = dense.tile_palettes
    sparse.palette = dense.palette
    sparse.optimize()
    dense.set_chunk(100)
    sparse.set_chunk(100)
    cand = S.random_candidates(0x5EED0003, 4 * 15 + 2, 250)
    cand[0] = dense.palette[4 * 15 + 2]
    for n in (100, 250, 60, 250, 100):  # one lane; two lanes (grows); one lane again on the grown storage; two; one
        ed = dense.score_candidates(4, 2, cand[:n])
        es = sparse.score_candidates(4, 2, cand[:n])
        assert np.array_equal(ed, es), (n, float(np.max(np.abs(ed - es))))
    e_d, b_d = dense.step(S.METHOD_RANDOM, 4, 2, 0, 11, 1, 250)
    e_s, b_s = sparse.step(S.METHOD_RANDOM, 4, 2, 0, 11, 1, 250)
    assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map)
    dense.close()
    sparse.close()


@pytest.mark.parametrize("flags", [{"dither": True}, {"perceptual": True}, {}], ids=["dither", "perceptual", "rgb"])
@pytest.mark.parametrize("knobs", [{}, {"SNES_H2Q_MAX": "0", "SNES_DITHER4_MAX": "0"}, {"SNES_H2Q_MAX": "100000", "SNES_DITHER4_MAX": "100000"},
                                   {"SNES_VSPLIT": "0", "SNES_SCAN4_MAX": "0", "SNES_DOWN_TILES": "0", "SNES_H0_MIN": "0"}, {"SNES_SCAN4_MAX": "100000"}],
                         ids=["default", "one-lane", "quad", "round3-base-sweep-scan-and-downscale", "scan4-everywhere"])
def test_long_lists_pin_both_kernel_families(S, img256_alpha, flags, knobs, monkeypatch):
    """Launch groups of more than 512 candidates take the one-lane H pass (k_sparse_h2) and the one-lane resumed
    Floyd-Steinberg (k_dither MODE 2), shorter ones the quad-per-row kernels (k_sparse_h2q, k_dither4): a 1,100-candidate list
    (two lanes of 550 with --dither / --perceptual-palettes, one group otherwise) and a 300-candidate list, at the default
    thresholds and with each family forced everywhere, against the dense path (SNES_SPARSE=0), bit for bit.  Round 4's
    variants ride along: B's V sweep on one wave instead of two (SNES_VSPLIT=0), the scan with one wave per candidate or
    four everywhere (SNES_SCAN4_MAX), the candidates' scales 2.. a block per candidate instead of a block per changed group of
    scale 3 (SNES_DOWN_TILES=0), the H pass of scale 0 behind the downscale instead of beside it on a stream of its own
    (SNES_H0_MIN=0; the default takes the side-by-side path for the 1,100-candidate list)."""
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img256_alpha, 8, 15, **flags)
    monkeypatch.delenv("SNES_SPARSE")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    sparse = S.OptimizedImage(img256_alpha, 8, 15, **flags)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes, sparse.palette = dense.tile_palettes, dense.palette
    sparse.optimize()
    pal = dense.palette
    for (sp, si), n in (((5, 2), 1100), ((1, 14), 300)):
        cand = S.random_candidates(31, sp * 15 + si, n)
        cand[0] = pal[sp * 15 + si]
        cand[1] = pal[sp * 15 + (si + 1) % 15]
        ed = dense.score_candidates(sp, si, cand)
        es = sparse.score_candidates(sp, si, cand)
        assert np.array_equal(ed, es), ((sp, si), n, float(np.max(np.abs(ed - es))))
    e_d, b_d = dense.step(S.METHOD_RANDOM, 4, 4, 0, 9, 1, 1100)  # the commit (with --dither: the winner's resumed map) from a long list
    e_s, b_s = sparse.step(S.METHOD_RANDOM, 4, 4, 0, 9, 1, 1100)
    assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map)
    dense.close()
    sparse.close()


@pytest.mark.parametrize("count,size", [(8, 15), (4, 7)])
def test_quad_dither_kernel_equals_the_one_lane_kernel(S, img256_alpha, count, size, monkeypatch):
    """k_dither4 (a quad of lanes per row: channel per lane, the entry search split four ways) against k_dither (one lane per
    row) in all three roles — the whole image (optimize), the base image B (records and checkpoints) and the resumed runs —
    bit for bit: palette_map after optimize(), the candidates' errors, and the map the commit adopts."""
    res = []
    for quad in ("0", "1"):
        monkeypatch.setenv("SNES_DITHER4", quad)
        monkeypatch.setenv("SNES_DITHER4_MAX", "100000")
        g = S.OptimizedImage(img256_alpha, count, size, dither=True)
        if not res:
            g.initialize_tiles()
            g.recalculate_palettes()
            tiles, pal = g.tile_palettes, g.palette
        else:
            g.tile_palettes, g.palette = tiles, pal
        g.optimize()
        m0 = g.palette_map.copy()
        cand = S.random_candidates(77, 1 * size + 2, 300)
        errs = g.score_candidates(1, 2, cand)
        e, best = g.step(S.METHOD_RANDOM, 1, 2, 0, 77, 5, 300)
        res.append((m0, errs, e, best, g.palette_map.copy()))
        g.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,variant", [(0x5EED0000, 0), (0x5EED0001, 1)])
def test_resumed_dither_equals_full_dither(S, O, seed, variant, monkeypatch):
    """--dither through causality: B dithered once per slot with the slot's entry out of play, every candidate resumed
    from B's checkpoint at the 4-row group of the first pixel it takes, the rows from there on scored by the group-sparse
    kernels — against the round-1 path (every candidate dithered from row 0, dense scoring; SNES_SPARSE=0) bit for bit,
    and against the oracle for a handful.  Candidates include the incumbent colour, duplicates of neighbours, colours
    that win nothing and colours that win the very first pixel."""
    from snesimage_amd.synth import synth_image
    img = synth_image(seed, variant=variant)
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img, 8, 15, dither=True)
    monkeypatch.delenv("SNES_SPARSE")
    sparse = S.OptimizedImage(img, 8, 15, dither=True)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes = dense.tile_palettes
    sparse.palette = dense.palette
    sparse.optimize()
    pal = dense.palette
    o = O.OracleImage(img, 8, 15, dither=True)
    o.tile_palettes = dense.tile_palettes
    o.palette = dense.palette
    o.optimize()
    for slot in [(2, 3), (0, 0), (7, 14)]:
        cand = S.random_candidates(seed, slot[0] * 15 + slot[1], 160)
        cand[0] = pal[slot[0] * 15 + slot[1]]
        cand[1] = pal[slot[0] * 15 + (slot[1] + 1) % 15]
        cand[2] = pal[slot[0] * 15 + (slot[1] + 14) % 15]
        cand[3] = [0, 0, 0]
        cand[4] = [31, 31, 31]
        ed = dense.score_candidates(slot[0], slot[1], cand)
        es = sparse.score_candidates(slot[0], slot[1], cand)
        assert np.array_equal(ed, es), (slot, float(np.max(np.abs(ed - es))))
        assert rel(es[:6], o.score_candidates(slot[0], slot[1], cand[:6])) < REL_ERR
    # full optimizer calls: the commit adopts the winner's resumed map
    for i, (p, idx) in enumerate([(3, 3), (3, 4), (0, 0)]):
        e_d, b_d = dense.step(S.METHOD_RANDOM, p, idx, 0, 9, i, 128)
        e_s, b_s = sparse.step(S.METHOD_RANDOM, p, idx, 0, 9, i, 128)
        assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map)
    e_o, b_o = None, None
    dense.close()
    sparse.close()


def test_dither_with_perceptual_palettes_on_the_sparse_path(S, O, monkeypatch):
    """--dither --perceptual-palettes through causality as well (round 3): B dithered with CIEDE2000 and the slot's entry out
    of play, its record in distance bits (ties to the lower index as bits + 1) and the targets' Lab beside it; the first pixel
    a candidate takes found by the very call the resumed run will make; resumed runs; group-sparse scoring of what differs.
    Against the dense path (every candidate dithered from row 0, SNES_SPARSE=0) bit for bit — scores, committed colour,
    committed map over consecutive slots (B a call ahead) — and against the oracle for a handful."""
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0003, variant=1)
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img, 8, 15, dither=True, perceptual=True)
    monkeypatch.delenv("SNES_SPARSE")
    sparse = S.OptimizedImage(img, 8, 15, dither=True, perceptual=True)
    monkeypatch.setenv("SNES_DITHER4_MAX", "0")  # every list on the one-lane-per-row kernels (what lists of more than 512 take)
    lanes = S.OptimizedImage(img, 8, 15, dither=True, perceptual=True)
    monkeypatch.delenv("SNES_DITHER4_MAX")
    dense.initialize_tiles()
    dense.recalculate_palettes()
    for im in (sparse, lanes):
        im.tile_palettes = dense.tile_palettes
        im.palette = dense.palette
        im.optimize()
    pal = dense.palette
    o = O.OracleImage(img, 8, 15, dither=True, perceptual=True)
    o.tile_palettes = dense.tile_palettes
    o.palette = dense.palette
    o.optimize()
    assert np.array_equal(sparse.palette_map, o.palette_map)
    for slot in [(2, 3), (0, 0), (7, 14)]:
        cand = S.random_candidates(21, slot[0] * 15 + slot[1], 96)
        cand[0] = pal[slot[0] * 15 + slot[1]]
        cand[1] = pal[slot[0] * 15 + (slot[1] + 1) % 15]   # the stand-in's colour: every tie with it goes by index
        cand[2] = pal[slot[0] * 15 + (slot[1] + 14) % 15]
        cand[3] = [0, 0, 0]
        cand[4] = [31, 31, 31]
        ed = dense.score_candidates(slot[0], slot[1], cand)
        es = sparse.score_candidates(slot[0], slot[1], cand)
        assert np.array_equal(ed, es), (slot, int(np.argmax(ed != es)), float(np.max(np.abs(ed - es))))
        assert np.array_equal(es, lanes.score_candidates(slot[0], slot[1], cand))
        assert rel(es[:3], o.score_candidates(slot[0], slot[1], cand[:3])) < REL_ERR
    for i, (p, idx) in enumerate([(3, 3), (3, 4), (3, 5), (3, 6), (0, 0)]):  # consecutive slots: B comes a call ahead
        e_d, b_d = dense.step(S.METHOD_RANDOM, p, idx, 0, 9, i, 80)
        e_s, b_s = sparse.step(S.METHOD_RANDOM, p, idx, 0, 9, i, 80)
        assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map), i
    dense.close()
    sparse.close()
    lanes.close()


@pytest.mark.parametrize("sub_count,sub_size", [(8, 15), (5, 7)])
def test_long_dither_lists_one_wave_per_run(S, img256_alpha, sub_count, sub_size, monkeypatch):
    """Lists of more than 512 candidates resume their Floyd-Steinberg runs one wave per run (k_ditherw: the row above over
    DPP, a shared entry table with the slot's stand-in, the run's own colour from registers) — against the round-1 path
    (every candidate dithered from row 0 by k_dither, dense scoring) bit for bit, with the unrolled 15-entry search and
    the generic one, on an image with transparent pixels; and k_ditherw against the two-wave k_dither (SNES_DITHERW=0)."""
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img256_alpha, sub_count, sub_size, dither=True)
    monkeypatch.delenv("SNES_SPARSE")
    wave = S.OptimizedImage(img256_alpha, sub_count, sub_size, dither=True)
    monkeypatch.setenv("SNES_DITHERW", "0")
    ring = S.OptimizedImage(img256_alpha, sub_count, sub_size, dither=True)
    monkeypatch.delenv("SNES_DITHERW")
    dense.initialize_tiles()
    dense.recalculate_palettes()
    for im in (wave, ring):
        im.tile_palettes = dense.tile_palettes
        im.palette = dense.palette
        im.optimize()
    pal = dense.palette
    for slot in [(0, 0), (sub_count - 1, sub_size - 1), (2, 3)]:
        ci = slot[0] * sub_size + slot[1]
        cand = S.random_candidates(77, ci, 700)
        cand[0] = pal[ci]
        cand[1] = pal[slot[0] * sub_size + (slot[1] + 1) % sub_size]
        cand[2] = [0, 0, 0]
        cand[3] = [31, 31, 31]
        ed = dense.score_candidates(slot[0], slot[1], cand)
        ew = wave.score_candidates(slot[0], slot[1], cand)
        er = ring.score_candidates(slot[0], slot[1], cand)
        assert np.array_equal(ed, ew), (slot, int(np.argmax(ed != ew)), float(np.max(np.abs(ed - ew))))
        assert np.array_equal(ew, er)
    e_d, b_d = dense.step(S.METHOD_RANDOM, 1, 2, 0, 9, 0, 640)
    e_w, b_w = wave.step(S.METHOD_RANDOM, 1, 2, 0, 9, 0, 640)
    assert e_d == e_w and np.array_equal(b_d, b_w) and np.array_equal(dense.palette_map, wave.palette_map)
    for im in (dense, wave, ring):
        im.close()


def test_dither_base_image_a_call_ahead(S, monkeypatch):
    """--dither, call by call: B of the scheduler's next slot is dithered on the side stream during a call and taken by the
    next call iff the commit in between kept the palette (a device flag).  A run over consecutive slots — accepted and
    rejected calls mixed, a call on an unexpected slot, a palette written from outside — equals the run without it."""
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0004)
    ahead = S.OptimizedImage(img, 8, 15, dither=True)
    monkeypatch.setenv("SNES_DITHER_AHEAD", "0")
    plain = S.OptimizedImage(img, 8, 15, dither=True)
    monkeypatch.delenv("SNES_DITHER_AHEAD")
    plain.initialize_tiles()
    plain.recalculate_palettes()
    ahead.tile_palettes = plain.tile_palettes
    ahead.palette = plain.palette
    ahead.optimize()
    plain.optimize()
    calls = S.schedule(8, 15, 40)
    order = [(p, i) for (_, p, i, _, _) in calls[:30]] + [(5, 5), (5, 6), (5, 7), (0, 0), (0, 1)]
    for k, (p, i) in enumerate(order):
        if k == 20:  # the palette changes behind the optimizer's back: the B made ahead is void
            pal = plain.palette.copy()
            pal[17] = [3, 30, 9]
            plain.palette = pal
            ahead.palette = pal
        e_p, b_p = plain.step(S.METHOD_RANDOM, p, i, 0, 5, k, 96)
        e_a, b_a = ahead.step(S.METHOD_RANDOM, p, i, 0, 5, k, 96)
        assert e_p == e_a and np.array_equal(b_p, b_a), k
    assert np.array_equal(plain.palette, ahead.palette) and np.array_equal(plain.palette_map, ahead.palette_map)
    assert plain.error() == ahead.error()
    for slot in [(2, 3), (2, 4), (2, 5), (2, 5), (6, 0)]:  # a sweep without commits: the B made ahead stands as it is
        cand = S.random_candidates(3, slot[0] * 15 + slot[1], 80)
        assert np.array_equal(plain.score_candidates(slot[0], slot[1], cand), ahead.score_candidates(slot[0], slot[1], cand)), slot
    plain.close()
    ahead.close()


@pytest.mark.parametrize("flags", [{}, {"perceptual": True}, {"dither": True}])
def test_reassign_tiles_matches_oracle(S, O, img256_alpha, flags):
    """snesimage_reassign_tiles (not in the reference: TODO.md:36-37): tiles moved, tile_palettes, palette_map and error
    against the oracle, straight after clustering and again after optimizer calls have changed the palette."""
    g, o = pair(S, O, img256_alpha, 4, 7, **flags)
    g.initialize_tiles(); o.initialize_tiles()
    g.recalculate_palettes(); o.recalculate_palettes()
    for rnd in range(2):
        mg, mo = g.reassign_tiles(), o.reassign_tiles()
        assert mg == mo and (rnd > 0 or mo > 0)
        assert np.array_equal(g.tile_palettes, o.tile_palettes) and np.array_equal(g.palette_map, o.palette_map)
        assert rel(g.error(), o.error()) < REL_ERR
        assert g.reassign_tiles() == 0
        for i, (p, idx) in enumerate([(0, 0), (3, 6), (1, 2)]):
            eg, bg = g.step(S.METHOD_RANDOM, p, idx, 0, 3, 10 * rnd + i, 24)
            eo, bo = o.step(0, p, idx, 0, 3, 10 * rnd + i, 24)
            assert np.array_equal(bg, bo) and abs(eg - eo) <= REL_ERR * abs(eo)
    assert g.as_json() == o.as_json()
    g.close()


@pytest.mark.parametrize("count,size", [(4, 1), (2, 2), (1, 15)])
def test_resumed_dither_small_subpalettes(S, O, img256_alpha, count, size, monkeypatch):
    """The resumed --dither path where B has no or a single other entry to fall back on (a one-entry subpalette: the candidate
    takes every opaque pixel of the subpalette, from the first one), and with one subpalette for the whole image."""
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img256_alpha, count, size, dither=True)
    monkeypatch.delenv("SNES_SPARSE")
    sparse = S.OptimizedImage(img256_alpha, count, size, dither=True)
    rng = np.random.default_rng(5)
    pal = rng.integers(0, 32, size=(count * size, 3)).astype(np.uint8)
    tp = rng.integers(0, count, size=1024).astype(np.uint8)
    for g in (dense, sparse):
        g.tile_palettes = tp
        g.palette = pal
        g.optimize()
    o = O.OracleImage(img256_alpha, count, size, dither=True)
    o.tile_palettes = tp
    o.palette = pal
    o.optimize()
    slot = (count - 1, size - 1)
    cand = S.random_candidates(3, 1, 80)
    cand[0] = pal[slot[0] * size + slot[1]]
    ed, es = dense.score_candidates(slot[0], slot[1], cand), sparse.score_candidates(slot[0], slot[1], cand)
    assert np.array_equal(ed, es)
    assert rel(es[:4], o.score_candidates(slot[0], slot[1], cand[:4])) < REL_ERR
    e_d, b_d = dense.step(S.METHOD_RANDOM, slot[0], slot[1], 0, 4, 0, 96)
    e_s, b_s = sparse.step(S.METHOD_RANDOM, slot[0], slot[1], 0, 4, 0, 96)
    assert e_d == e_s and np.array_equal(b_d, b_s) and np.array_equal(dense.palette_map, sparse.palette_map)
    dense.close()
    sparse.close()


# ---- one process, several devices: RCCL inside the library ---------------------------------------------
def test_group_step_over_rccl_equals_plain_step(S, img256):
    """snesimage_group_* with the devices this box has (one): step_begin -> grouped ncclAllReduce(min) -> step_commit must
    reproduce snesimage_step; two contexts on one device are refused (a group takes one context per device)."""
    import ctypes as C
    from snesimage_amd import _ffi
    L = _ffi.load()
    ref = S.OptimizedImage(img256, 8, 15)
    ref.initialize_tiles()
    ref.recalculate_palettes()
    mem = S.OptimizedImage(img256, 8, 15)
    mem.tile_palettes = ref.tile_palettes
    mem.palette = ref.palette
    mem.optimize()
    arr = (C.c_void_p * 1)(mem._c)
    grp = C.c_void_p()
    assert L.snesimage_group_create(arr, 1, C.byref(grp)) == 0, L.snesimage_last_error()
    for i, (method, p, idx, ch, n) in enumerate([(0, 0, 0, 0, 40), (0, 5, 9, 0, 200), (1, 2, 3, 1, 0), (0, 0, 0, 0, 40)]):
        e_ref, b_ref = ref.step(method, p, idx, ch, 6, i, n)
        err, best = C.c_double(0), np.zeros(3, np.uint8)
        assert L.snesimage_group_step(grp, method, p, idx, ch, 6, i, n, C.byref(err), best.ctypes.data_as(_ffi._u8p)) == 0, L.snesimage_last_error()
        assert err.value == e_ref and np.array_equal(best, b_ref)
        assert np.array_equal(mem.palette, ref.palette) and np.array_equal(mem.palette_map, ref.palette_map)
    L.snesimage_group_destroy(grp)
    twin = S.OptimizedImage(img256, 8, 15)
    arr2 = (C.c_void_p * 2)(mem._c, twin._c)
    assert L.snesimage_group_create(arr2, 2, C.byref(grp)) == -1 and b"one context per device" in L.snesimage_last_error()
    # lifetime rules (include/snesimage_hip.h): one group per context; a member destroyed first retires the group
    assert L.snesimage_group_create(arr, 1, C.byref(grp)) == 0
    grp2 = C.c_void_p()
    assert L.snesimage_group_create(arr, 1, C.byref(grp2)) == -3 and b"already belongs to a group" in L.snesimage_last_error()
    mem.close()
    assert L.snesimage_group_step(grp, 0, 0, 0, 0, 6, 9, 40, None, None) == -3 and b"destroyed" in L.snesimage_last_error()
    L.snesimage_group_destroy(grp)
    for z in (ref, twin):
        z.close()


def test_edge_map_product_form_stays_within_bounds_on_adversarial_inputs(S, O):
    """The product evaluates the edge-difference map as (|img2 - mu2| - a1) * r1 with r1 = 1 / (1 + a1) precomputed per image,
    the oracle (like the crate) as (1 + |img2 - mu2|) / (1 + |img1 - mu1|) - 1: equal in exact arithmetic, ~1e-16 apart per
    pixel in binary64.  Worst case for that difference: an image of hard edges and noise (large a1 everywhere) and candidates
    that are near-duplicates of each other (errors a few ulps apart).  Bound the deviation, and require the ORDER of the
    candidates' errors — what the optimizer acts on — to be the oracle's wherever the oracle's errors differ by more than it."""
    rng = np.random.default_rng(5)
    img = np.zeros((256, 256, 4), np.uint8)
    img[..., :3] = rng.integers(0, 2, (256, 256, 1), dtype=np.uint8) * 255  # black / white salt and pepper
    img[64:192, 64:192, :3] = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)  # full-range colour noise
    img[..., 3] = 255
    g, o = pair(S, O, img, 4, 7)
    o.tile_palettes = (np.arange(1024) % 4).astype(np.uint8)  # (the k-means initialisers have nothing to cluster in pure noise)
    o.palette = rng.integers(0, 32, (28, 3), dtype=np.uint8)
    o.optimize()
    sync_state(g, o)
    base = np.array([16, 16, 16], np.uint8)
    cand = np.stack([np.clip(base.astype(int) + d, 0, 31).astype(np.uint8) for d in
                     ([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 1, 0], [1, 0, 1], [0, 1, 1],
                      [-1, -1, 0], [1, -1, 0], [2, 0, 0], [0, 2, 0], [0, 0, 2], [1, 1, 1])] +
                    [[0, 0, 0], [31, 31, 31], [31, 0, 0], [0, 31, 0]])
    eg = g.score_candidates(1, 3, cand)
    eo = o.score_candidates(1, 3, cand)
    dev = np.abs(eg - eo) / np.abs(eo)
    assert dev.max() < 1e-13, dev.max()
    gap = 4 * np.abs(eg - eo).max()
    for a in range(len(cand)):
        for b in range(len(cand)):
            if eo[a] + gap < eo[b]:
                assert eg[a] < eg[b], (a, b)
    g.close()


@pytest.mark.parametrize("h", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("flags", [{}, {"dither": True}, {"perceptual": True}], ids=["rgb", "dither", "perceptual"])
def test_small_heights_sparse_path_equals_dense_path(S, O, h, flags, monkeypatch):
    """Heights below 256 on the group-sparse path, down to the library's smallest (8 rows: two 4-row groups, two scales;
    round 4 — until then B's downscale, which walks 32 x 32 blocks of pixels, kept 8 and 16 rows on the dense path): the
    narrowest scales are a single 4-row group there.  Against the dense path bit for bit, a handful against the oracle, and
    the slot windows against call-by-call stepping."""
    from snesimage_amd.synth import synth_image
    img = synth_image(0x5EED0020 + h, 256, h, 1 if h == 64 else 0)
    monkeypatch.setenv("SNES_SPARSE", "0")
    dense = S.OptimizedImage(img, 4, 7, **flags)
    monkeypatch.setenv("SNES_SPARSE", "1")
    monkeypatch.setenv("SNES_SPARSE_MIN", "1")
    sparse = S.OptimizedImage(img, 4, 7, **flags)
    dense.initialize_tiles()
    dense.recalculate_palettes()
    sparse.tile_palettes, sparse.palette = dense.tile_palettes, dense.palette
    sparse.optimize()
    pal = dense.palette
    for (sp, si), n in (((1, 2), 150), ((3, 6), 40)):
        cand = S.random_candidates(13, sp * 7 + si + h, n)
        cand[0] = pal[sp * 7 + si]
        cand[1] = pal[sp * 7 + (si + 1) % 7]
        ed, es = dense.score_candidates(sp, si, cand), sparse.score_candidates(sp, si, cand)
        assert np.array_equal(ed, es), (h, sp, si, float(np.max(np.abs(ed - es))))
    o = O.OracleImage(img, 4, 7, **flags)
    o.tile_palettes, o.palette = dense.tile_palettes, dense.palette
    o.optimize()
    assert rel(es[:5], o.score_candidates(3, 6, cand[:5])) < REL_ERR
    sched = S.schedule(4, 7, 60)
    state = sched[20][1:]
    log, _, stats = sparse.run_slots(40, seed=3, first_step_id=20, state=state)
    for j in range(40):
        m, p, i, ch, _ = sched[20 + j]
        e, b = dense.step(m, p, i, ch, 3, 20 + j, 0)
        assert (e, b.tolist()) == (log[j][0], log[j][2].tolist()), (h, j)
    assert np.array_equal(dense.palette, sparse.palette) and np.array_equal(dense.palette_map, sparse.palette_map)
    assert stats["windows"] < 40
    dense.close()
    sparse.close()


@pytest.mark.gpu
def test_launch_timing_levels_leave_the_results_alone(S, img256):
    """snesimage_timing_enable: 1 = every bracket (group, H pass, V pass), 2 = the V pass's only (what bench.py keeps on over its
    timed region: two event records per launch group instead of six).  Same errors either way and with timing off."""
    g = S.OptimizedImage(img256, 4, 7)
    g.initialize_tiles()
    g.recalculate_palettes()
    g.optimize()
    cand = S.random_candidates(5, 11, 200)
    want = g.score_candidates(2, 3, cand)
    for level in (1, 2):
        g.timing_enable(level)
        got = g.score_candidates(2, 3, cand)
        t = g.timing_read()
        g.timing_enable(False)
        assert np.array_equal(got, want)
        assert t["launches"] >= 1 and t["candidates"] == 200 and t["vpass0_ms"] > 0.0
        if level == 1:
            assert t["group_ms"] >= t["vpass0_ms"] and t["hpass0_ms"] > 0.0
        else:
            assert t["group_ms"] == 0.0 and t["hpass0_ms"] == 0.0
    g.close()
