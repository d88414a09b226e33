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
@pytest.mark.parametrize("batched,flags", [(False, {}), (True, {}), (True, {"perceptual": True})])
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
    import snesimage_amd as S
    from snesimage_amd.throughput import ImageBatch
    for flags in ({"dither": True}, {"dither": True, "perceptual": True}):
        b = ImageBatch.synthetic([1, 2], 2, 3, candidates=8, batched=True, **flags)
        with pytest.raises(S.SnesImageError) as e:
            b.initialize()
        assert e.value.code == -5
        b.close()


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
