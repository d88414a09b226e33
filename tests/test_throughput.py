"""Throughput mode: image sharding on the CPU, concurrent contexts on the GPU."""
import numpy as np
import pytest

from snesimage_amd.throughput import shard_images


@pytest.mark.parametrize("n,world", [(1024, 8), (1024, 1), (10, 4), (3, 8), (0, 2), (129, 2)])
def test_shard_images_is_a_partition(n, world):
    blocks = [shard_images(n, r, world) for r in range(world)]
    flat = [i for b in blocks for i in b]
    assert flat == list(range(n))                      # disjoint, complete, contiguous and in rank order
    sizes = [len(b) for b in blocks]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_images(n, world, world)


@pytest.mark.gpu
@pytest.mark.parametrize("batched,flags", [(False, {}), (True, {}), (True, {"perceptual": True}), (True, {"dither": True}), (True, {"dither": True, "perceptual": True})])
def test_concurrent_images_equal_one_at_a_time(batched, flags):
    """Optimizer calls of several images — enqueued side by side from several host threads, or issued as one launch per
    stage for all of them — give, for every image, exactly the state that stepping that image alone gives."""
    import snesimage_amd as S
    from snesimage_amd.synth import synth_image
    from snesimage_amd.throughput import IMAGE_SEED0, ImageBatch

    ids = [5, 6, 7, 900]
    batch = ImageBatch.synthetic(ids, 4, 7, candidates=24, host_threads=3, batched=batched, **flags)
    batch.initialize()
    batch.run(3)
    batch.run(4)  # the schedule continues where the first run stopped
    sched = S.schedule(4, 7, 7)
    for pos, gid in enumerate(ids):
        solo = S.OptimizedImage(synth_image(IMAGE_SEED0 + gid), 4, 7, **flags)
        solo.initialize_tiles()
        solo.recalculate_palettes()
        for j, (method, p, idx, ch, _) in enumerate(sched):
            e, _ = solo.step(method, p, idx, ch, 1 + gid, j, 24 if method == S.METHOD_RANDOM else 0)
        img = batch.images[pos]
        assert np.array_equal(img.palette, solo.palette) and np.array_equal(img.palette_map, solo.palette_map)
        assert np.array_equal(img.tile_palettes, solo.tile_palettes)
        assert batch.errors()[pos] == e
        solo.close()
    batch.close()


@pytest.mark.gpu
def test_batch_rejects_what_it_does_not_cover():
    """The batched launches cover the group-sparse path.  Round 4: that is every height (16-row images, refused until then, now
    step like their solo twins); what is left — --dither with one-entry subpalettes — is refused, not mis-stepped."""
    import ctypes as C
    import numpy as np
    import snesimage_amd as S
    from snesimage_amd import _ffi
    from snesimage_amd.synth import synth_image
    from snesimage_amd.throughput import IMAGE_SEED0, ImageBatch
    imgs = [(i, synth_image(IMAGE_SEED0 + i, 256, 16)) for i in (1, 2)]
    b = ImageBatch(iter(imgs), 2, 3, candidates=8, batched=True)
    b.initialize()
    solo = [S.OptimizedImage(im, 2, 3) for _, im in imgs]
    for z, m in zip(solo, b.images):
        z.tile_palettes, z.palette = m.tile_palettes, m.palette
        z.optimize()
    b.run(5)
    sched = S.schedule(2, 3, 5)
    for (gid, _), z, m in zip(imgs, solo, b.images):
        for j, (method, p, idx, ch, _) in enumerate(sched):
            z.step(method, p, idx, ch, 1 + gid, j, 8)  # (ImageBatch.run: candidate streams keyed (1 + image index, call number))
        assert np.array_equal(z.palette, m.palette) and np.array_equal(z.palette_map, m.palette_map) and z.error() == m.error()
    b.close()
    for z in solo:
        z.close()
    L = _ffi.load()
    ctxs = []
    for _, im in imgs:
        g = S.OptimizedImage(synth_image(IMAGE_SEED0, 256, 64), 4, 1, dither=True)
        g.tile_palettes = (np.arange(1024) % 4).astype(np.uint8)
        g.palette = np.array([[3, 5, 7], [20, 11, 2], [9, 30, 14], [28, 27, 25]], np.uint8)
        g.optimize()
        ctxs.append(g)
    arr = (C.c_void_p * 2)(*[g._c for g in ctxs])
    out = C.c_void_p()
    assert L.snesimage_batch_create(arr, 2, C.byref(out)) == -5, L.snesimage_last_error()
    for g in ctxs:
        g.close()


@pytest.mark.gpu
def test_batched_channel_calls_and_implicit_sync():
    """The batched form through random AND channel calls of the reference's schedule (2 x 3 entries: calls 24.. are the
    channel sweeps), read back through a member context without an explicit sync."""
    import snesimage_amd as S
    from snesimage_amd import _ffi
    from snesimage_amd.synth import synth_image
    from snesimage_amd.throughput import IMAGE_SEED0, ImageBatch

    ids = [11, 12, 13]
    batch = ImageBatch.synthetic(ids, 2, 3, candidates=10, batched=True, groups=2)
    batch.initialize()
    L = _ffi.load()
    sched = S.schedule(2, 3, 30)
    for j, (method, p, idx, ch, _) in enumerate(sched):  # enqueue only: no sync anywhere
        for h, seeds in batch._batches:
            assert L.snesimage_batch_step_async(h, method, p, idx, ch, seeds, j, 10 if method == S.METHOD_RANDOM else 0) == 0
    got = [(img.palette.copy(), img.palette_map.copy(), img.last_step()[0]) for img in batch.images]  # waits for the batch's stream
    for pos, gid in enumerate(ids):
        solo = S.OptimizedImage(synth_image(IMAGE_SEED0 + gid), 2, 3)
        solo.initialize_tiles()
        solo.recalculate_palettes()
        for j, (method, p, idx, ch, _) in enumerate(sched):
            e, _ = solo.step(method, p, idx, ch, 1 + gid, j, 10 if method == S.METHOD_RANDOM else 0)
        assert np.array_equal(got[pos][0], solo.palette) and np.array_equal(got[pos][1], solo.palette_map) and got[pos][2] == e
        solo.close()
    batch.close()


@pytest.mark.gpu
def test_destroying_a_member_context_retires_the_batch():
    import snesimage_amd as S
    from snesimage_amd import _ffi
    from snesimage_amd.throughput import ImageBatch
    batch = ImageBatch.synthetic([21, 22], 2, 3, candidates=8, batched=True, groups=1)
    batch.initialize()
    batch.run(2)
    batch.images[1].close()  # while lent to the batch
    L = _ffi.load()
    h, seeds = batch._batches[0]
    assert L.snesimage_batch_step_async(h, 0, 0, 0, 0, seeds, 2, 8) == -3
    assert b"destroyed" in L.snesimage_last_error()
    batch.images[0].step(S.METHOD_RANDOM, 0, 0, 0, 1, 2, 8)  # the surviving context is its own again
    batch.close()


@pytest.mark.gpu
def test_member_work_on_its_own_stream_is_ordered_before_the_next_batched_call():
    """a call on a member context between two batched calls (here an un-synchronised snesimage_step_async on
    the member's own stream) must be finished before the batch touches that member's pack, palette and sparse storage
    again; snesimage_set_chunk is refused while the context is lent."""
    import snesimage_amd as S
    from snesimage_amd import _ffi
    from snesimage_amd.synth import synth_image
    from snesimage_amd.throughput import IMAGE_SEED0, ImageBatch

    ids = [31, 32, 33]
    batch = ImageBatch.synthetic(ids, 2, 3, candidates=64, batched=True, groups=1)
    batch.initialize()
    L = _ffi.load()
    h, seeds = batch._batches[0]
    with pytest.raises(S.SnesImageError) as e:
        batch.images[0].set_chunk(128)
    assert e.value.code == -3
    sched = S.schedule(2, 3, 6)
    for j, (method, p, idx, ch, _) in enumerate(sched[:3]):
        assert L.snesimage_batch_step_async(h, method, p, idx, ch, seeds, j, 64) == 0
    m, p, idx, ch, _ = sched[3]
    batch.images[1].step_async(m, p, idx, ch, 777, 3, 64)   # member 1 alone, on its own stream, no sync
    for j, (method, p, idx, ch, _) in enumerate(sched[4:], start=4):
        assert L.snesimage_batch_step_async(h, method, p, idx, ch, seeds, j, 64) == 0
    for pos, gid in enumerate(ids):
        solo = S.OptimizedImage(synth_image(IMAGE_SEED0 + gid), 2, 3)
        solo.initialize_tiles()
        solo.recalculate_palettes()
        for j, (method, p, idx, ch, _) in enumerate(sched):
            if j == 3 and pos != 1:
                continue
            e_solo, _ = solo.step(method, p, idx, ch, 777 if j == 3 else 1 + gid, j, 64)
        img = batch.images[pos]
        assert np.array_equal(img.palette, solo.palette) and np.array_equal(img.palette_map, solo.palette_map)
        assert img.last_step()[0] == e_solo
        solo.close()
    batch.close()


# ---- row T: throughput mode at BASELINE config 5's geometry (8 x 15) against the ORACLE ----------------------------
def _throughput_golden(name):
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")) as f:
        return next(c for c in json.load(f)["throughput"] if c["name"] == name)


def _run_batched(S, perceptual, calls, images, groups, dither=False):
    """The call list through snesimage_batch_step_async (one launch per stage for all images of a group)."""
    from snesimage_amd import _ffi
    from snesimage_amd.synth import synth_image
    from snesimage_amd.throughput import ImageBatch
    batch = ImageBatch(((gid, synth_image(seed, 256, 256, variant)) for gid, seed, variant in images), 8, 15, candidates=64,
                       batched=True, groups=groups, perceptual=perceptual, dither=dither)
    batch.initialize()
    assert batch.dropped == []
    L = _ffi.load()
    errs = [[] for _ in images]
    for j, (method, p, idx, ch) in enumerate(calls):
        for h, seeds in batch._batches:
            assert L.snesimage_batch_step_async(h, method, p, idx, ch, seeds, j, 64 if method == 0 else 0) == 0, L.snesimage_last_error()
        for pos, e in enumerate(batch.errors()):  # reading a member waits for its batch
            errs[pos].append(e)
    return batch, errs


@pytest.mark.gpu
@pytest.mark.parametrize("perceptual", [False, True])
def test_batched_throughput_mode_matches_oracle_at_8x15(O, perceptual):
    """BASELINE config 5 (the reference: `run()` once per file, lib.rs:833-853 + 888-933): eight 256x256 images incl. the
    transparent-square variant, 8 subpalettes x 15, stepped as two library batches through random calls with the
    reference's 64 candidates (lib.rs:205), a slot revisited after its commit and channel sweeps (lib.rs:296).  Every image's
    palette, palette_map, tile_palettes, per-call error and JSON must be what the ORACLE gives stepping that image alone:
    all eight live for the RGB distance; for --perceptual-palettes (CIEDE2000 costs the oracle minutes per image) images
    0-1 against the committed oracle fixture and the rest against a solo HIP context."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    import snesimage_amd as S
    import golden.make_golden as M
    from snesimage_amd.synth import synth_image

    gold = _throughput_golden("cfg5_8x15_perceptual_batch" if perceptual else "cfg5_8x15_rgb_batch")
    calls = [tuple(c) for c in gold["calls"]]
    assert calls == M.THROUGHPUT_CALLS and sum(1 for c in calls if c[0] == 1) >= 1 and len(calls) >= 6
    images = M.THROUGHPUT_IMAGES
    assert len(images) >= 8 and any(v == 1 for _, _, v in images)
    batch, errs = _run_batched(S, perceptual, calls, images, groups=2)

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    # the committed oracle fixture
    for rec in gold["images"]:
        pos = next(i for i, (gid, _, _) in enumerate(images) if gid == rec["gid"])
        img = batch.images[pos]
        assert img.palette.reshape(-1).tolist() == rec["palette"]
        assert sha(img.palette_map) == rec["map_sha"] and sha(img.tile_palettes) == rec["tile_palettes_sha"]
        want = np.array([float.fromhex(h) for h in rec["errors_hex"]])
        assert np.max(np.abs(np.array(errs[pos]) - want) / want) < 1e-11
        assert hashlib.sha256(img.as_json().encode()).hexdigest() == rec["json_sha"]
    if not perceptual:  # every image against the live oracle (ctypes drops the GIL: one thread per image)
        with ThreadPoolExecutor(len(images)) as ex:
            oracle = list(ex.map(lambda a: M.throughput_image(a[0], a[1], a[2], False), images))
        for pos, (o, oerrs) in enumerate(oracle):
            img = batch.images[pos]
            assert np.array_equal(img.palette, o.palette) and np.array_equal(img.palette_map, o.palette_map)
            assert np.array_equal(img.tile_palettes, o.tile_palettes) and img.as_json() == o.as_json()
            assert np.max(np.abs(np.array(errs[pos]) - np.array(oerrs)) / np.array(oerrs)) < 1e-11
            o.close()
    else:
        for pos, (gid, seed, variant) in enumerate(images):
            solo = S.OptimizedImage(synth_image(seed, 256, 256, variant), 8, 15, perceptual=True)
            solo.initialize_tiles()
            solo.recalculate_palettes()
            es = [solo.step(m, p, idx, ch, 1 + gid, j, 64 if m == 0 else 0)[0] for j, (m, p, idx, ch) in enumerate(calls)]
            img = batch.images[pos]
            assert np.array_equal(img.palette, solo.palette) and np.array_equal(img.palette_map, solo.palette_map) and es == errs[pos]
            solo.close()
    batch.close()


@pytest.mark.gpu
def test_batched_throughput_mode_with_dither_matches_oracle(O):
    """Throughput mode with --dither (round 3: the batched Floyd-Steinberg kernels): three 256x256 images incl. the
    transparent-square variant, 8 x 15, the reference's 64 / 32 candidates per call — random calls, a slot revisited after
    its commit, a channel sweep — against the ORACLE stepping every image alone: palette, palette_map (the winner's resumed
    run adopted as the committed map), per-call error, JSON."""
    from concurrent.futures import ThreadPoolExecutor

    import snesimage_amd as S
    import golden.make_golden as M
    from snesimage_amd.synth import synth_image

    calls = [(0, 0, 0, 0), (0, 3, 7, 0), (1, 3, 7, 1), (0, 3, 7, 0)]
    images = M.THROUGHPUT_IMAGES[:3]
    assert any(v == 1 for _, _, v in images)
    batch, errs = _run_batched(S, False, calls, images, groups=1, dither=True)

    def alone(a):
        gid, seed, variant = a
        o = O.OracleImage(synth_image(seed, 256, 256, variant), 8, 15, dither=True)
        o.initialize_tiles()
        o.recalculate_palettes()
        return o, [o.step(m, p, i, ch, 1 + gid, j, 64 if m == 0 else 0)[0] for j, (m, p, i, ch) in enumerate(calls)]

    with ThreadPoolExecutor(len(images)) as ex:
        oracle = list(ex.map(alone, images))
    for pos, (o, oerrs) in enumerate(oracle):
        img = batch.images[pos]
        assert np.array_equal(img.palette, o.palette) and np.array_equal(img.palette_map, o.palette_map), pos
        assert img.as_json() == o.as_json()
        assert np.max(np.abs(np.array(errs[pos]) - np.array(oerrs)) / np.array(oerrs)) < 1e-11
    batch.close()
