"""Throughput mode: a batch of independent images on one GPU (SURVEY §8d config 5, §8e "replicas only").

Every image is its own `OptimizedImage` context with its own HIP streams; the optimizer calls of
different images share nothing, so the only scheduling question is keeping the GPU's queues full.
A step with the reference's 64 candidates (lib.rs:205) is far too small to fill 256 CUs on its own;
here the calls of many images are enqueued side by side (`snesimage_step_async` never blocks on the
device) by a few host threads — ctypes drops the GIL for the duration of each call — and the device
overlaps them.  Across GPUs the images are split into contiguous blocks (`shard_images`); there is
no collective in this mode.
"""
import ctypes as C
import threading

from . import _ffi, api
from .synth import synth_image

IMAGE_SEED0 = 0x5EED0000  # image i of a batch is synth_image(IMAGE_SEED0 + i)


def shard_images(n_images, rank, world):
    """Contiguous block of image indices owned by `rank`: sizes differ by at most one, earlier ranks get the extra."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


class ImageBatch:
    """`images`: iterable of (global_index, rgba) pairs, all optimised with the same palette geometry and flags."""

    def __init__(self, images, sub_count, sub_size, device=0, candidates=64, host_threads=8, dither=False,
                 perceptual=False, nes=False, batched=False, groups=4):
        self.sub_count, self.sub_size, self.candidates, self.nes = int(sub_count), int(sub_size), int(candidates), bool(nes)
        self.batched, self._batches, self.groups = bool(batched), [], max(1, int(groups))
        self.ids = []
        self.images = []
        for gid, rgba in images:
            self.ids.append(int(gid))
            img = api.OptimizedImage(rgba, sub_count, sub_size, dither=dither, perceptual=perceptual, nes=nes, device=device)
            img.set_chunk(max(self.candidates, 64))  # workspace for one call's candidates, not the library's 1,024-candidate default
            self.images.append(img)
        self.host_threads = max(1, min(int(host_threads), len(self.images)))
        self.calls_done = 0
        self.dropped = []

    @classmethod
    def synthetic(cls, indices, sub_count, sub_size, **kw):
        return cls(((i, synth_image(IMAGE_SEED0 + i)) for i in indices), sub_count, sub_size, **kw)

    def __len__(self):
        return len(self.images)

    def _parallel(self, fn):
        """fn(positions) on every host thread, image positions dealt round-robin."""
        errs = []

        def work(t):
            try:
                fn(range(t, len(self.images), self.host_threads))
            except BaseException as e:  # surface the first failure in the caller's thread
                errs.append(e)

        ts = [threading.Thread(target=work, args=(t,)) for t in range(self.host_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    def initialize(self, drop_failed=False):
        """The reference's TileAssignment and Clustering phases for every image (lib.rs:982-1000).  `drop_failed`: an image
        on which cogset's k-means precondition fails (the reference panics there) leaves the batch instead of
        ending it; `self.dropped` lists the indices."""
        failed = []

        def init(mine):
            for i in mine:
                try:
                    self.images[i].initialize_tiles()
                    self.images[i].recalculate_palettes()
                except api.SnesImageError as e:
                    if not (drop_failed and e.code == -4):
                        raise
                    failed.append(i)
        self._parallel(init)
        self.dropped = [self.ids[i] for i in sorted(failed)]
        for i in sorted(failed, reverse=True):
            self.images[i].close()
            del self.images[i], self.ids[i]
        if self.batched:
            self._open_batch()

    def _open_batch(self):
        """`groups` library batches over interleaved halves (thirds, ...) of the images: each has its own stream, so the
        latency-bound stages of one group's call run beside the VALU-bound ones of another's."""
        L = _ffi.load()
        g = min(self.groups, len(self.images))
        for k in range(g):
            pos = list(range(k, len(self.images), g))
            arr = (C.c_void_p * len(pos))(*[self.images[i]._c for i in pos])
            h = C.c_void_p()
            rc = L.snesimage_batch_create(arr, len(pos), C.byref(h))
            if rc != 0:
                raise api.SnesImageError(rc, L.snesimage_last_error().decode())
            self._batches.append((h, (C.c_uint64 * len(pos))(*[1 + self.ids[i] for i in pos])))

    def run(self, n_calls):
        """Enqueue the next `n_calls` optimizer calls of the reference's slot schedule (lib.rs:881-933) for every image,
        then wait for the device.  Candidate streams are keyed (1 + image index, call number)."""
        sched = api.schedule(self.sub_count, self.sub_size, self.calls_done + n_calls, nes=self.nes)[self.calls_done:]
        first = self.calls_done
        if self._batches:  # one call = one launch per stage for all images of a group
            L = _ffi.load()
            for j, (method, p, idx, ch, _) in enumerate(sched):
                for h, seeds in self._batches:
                    rc = L.snesimage_batch_step_async(h, method, p, idx, ch, seeds, first + j,
                                                      self.candidates if method == api.METHOD_RANDOM else 0)
                    if rc != 0:
                        raise api.SnesImageError(rc, L.snesimage_last_error().decode())
            self.sync()
            self.calls_done += n_calls
            return

        def go(mine):  # breadth first: call j of every image before call j+1 of any, so all streams stay populated
            for j, (method, p, idx, ch, _) in enumerate(sched):
                n = self.candidates if method == api.METHOD_RANDOM else 0
                for i in mine:
                    self.images[i].step_async(method, p, idx, ch, 1 + self.ids[i], first + j, n)
        self._parallel(go)
        self.sync()
        self.calls_done += n_calls

    def sync(self):
        for h, _ in self._batches:
            rc = _ffi.load().snesimage_batch_sync(h)
            if rc != 0:
                raise api.SnesImageError(rc, _ffi.load().snesimage_last_error().decode())
        for img in self.images:
            img.sync()

    def errors(self):
        return [img.last_step()[0] for img in self.images]

    def close(self):
        for h, _ in self._batches:
            _ffi.load().snesimage_batch_destroy(h)
        self._batches = []
        for img in self.images:
            img.close()
        self.images = []
