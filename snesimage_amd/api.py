"""Host-side mirror of the reference's `OptimizedImage` (src/lib.rs:33-626) over the HIP library.

Method names, argument meaning and error behaviour follow the reference: every method that the
reference declares `-> anyhow::Result<..>` raises `SnesImageError` carrying the library's message.
All arithmetic runs in libsnesimage_hip.so on the GPU; this module only marshals buffers.
"""
import ctypes as C

import numpy as np

from . import _ffi

DITHER, PERCEPTUAL, NES = 1, 2, 4
METHOD_RANDOM, METHOD_CHANNEL, METHOD_NES = 0, 1, 2


class SnesImageError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("snesimage_hip error %d: %s" % (code, message))
        self.code = code


def _p(a, t):
    return a.ctypes.data_as(t)


def random_candidates(seed, step_id, n):
    """Counter-RNG candidate list of (seed, step_id): n x 3 raw 5-bit (r,g,b) — lib.rs:206-208."""
    out = np.zeros((n, 3), np.uint8)
    _ffi.load().snesimage_random_candidates(seed, step_id, n, _p(out, _ffi._u8p))
    return out


def schedule(sub_count, sub_size, n_calls, nes=False):
    """Replay the slot scheduler of lib.rs:881-933: [(method, palette, index, channel, step)]."""
    L = _ffi.load()
    p, i, ch, st, m = (C.c_uint32(0) for _ in range(5))
    out = []
    for _ in range(n_calls):
        cur = (p.value, i.value, ch.value, st.value)
        L.snesimage_schedule_next(sub_count, sub_size, int(nes), C.byref(p), C.byref(i), C.byref(ch), C.byref(st),
                                  C.byref(m))
        out.append((m.value,) + cur)
    return out


class OptimizedImage:
    """`OptimizedImage::new(source, palette_count, palette_size, dither, perceptual_palettes, nes)`."""

    def __init__(self, rgba, sub_count, sub_size, dither=False, perceptual=False, nes=False, device=0):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("rgba must be an (H, W, 4) uint8 array")
        self.h, self.w = int(rgba.shape[0]), int(rgba.shape[1])
        self.sub_count, self.sub_size = int(sub_count), int(sub_size)
        self.flags = (DITHER if dither else 0) | (PERCEPTUAL if perceptual else 0) | (NES if nes else 0)
        self._L = _ffi.load()
        ctx = C.c_void_p()
        rc = self._L.snesimage_create(_p(rgba, _ffi._u8p), self.w, self.h, self.sub_count, self.sub_size, self.flags,
                                      int(device), C.byref(ctx))
        if rc != 0:
            raise SnesImageError(rc, self._L.snesimage_last_error().decode())
        self._c = ctx

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_c", None):
            self._L.snesimage_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise SnesImageError(rc, self._L.snesimage_last_error().decode())

    # -- plumbing -------------------------------------------------------------------------------
    def set_stream(self, hip_stream_handle):
        self._chk(self._L.snesimage_set_stream(self._c, C.c_void_p(hip_stream_handle or 0)))

    def sync(self):
        self._chk(self._L.snesimage_sync(self._c))

    def set_chunk(self, chunk):
        self._chk(self._L.snesimage_set_chunk(self._c, int(chunk)))

    def timing_enable(self, on=True):
        """True / 1: every bracket (group, H pass, V pass); 2: the V pass's only (two event records per launch group instead of six)."""
        self._chk(self._L.snesimage_timing_enable(self._c, int(on)))

    def timing_read(self):
        ms = (C.c_double * 3)()
        n, k = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.snesimage_timing_read(self._c, ms, C.byref(n), C.byref(k)))
        return {"group_ms": ms[0], "hpass0_ms": ms[1], "vpass0_ms": ms[2], "launches": n.value, "candidates": k.value}

    # -- the reference's methods ------------------------------------------------------------------
    def initialize_tiles(self):  # lib.rs:79
        self._chk(self._L.snesimage_initialize_tiles(self._c))

    def recalculate_palettes(self):  # lib.rs:407
        self._chk(self._L.snesimage_recalculate_palettes(self._c))

    def optimize(self):  # lib.rs:425
        self._chk(self._L.snesimage_optimize(self._c))

    def error(self):  # lib.rs:503
        out = C.c_double(0)
        self._chk(self._L.snesimage_error(self._c, C.byref(out)))
        return out.value

    def reassign_tiles(self):
        """Move every tile to the subpalette that reproduces it best (not in the reference: TODO.md:36-37); tiles moved."""
        moved = C.c_uint32(0)
        self._chk(self._L.snesimage_reassign_tiles(self._c, C.byref(moved)))
        return moved.value

    def score_candidates(self, palette, index, rgb5):
        """Loop body of lib.rs:205-220 for an explicit candidate list -> errors (float64)."""
        cand = np.ascontiguousarray(rgb5, np.uint8).reshape(-1, 3)
        errs = np.zeros(cand.shape[0], np.float64)
        self._chk(self._L.snesimage_score_candidates(self._c, palette, index, _p(cand, _ffi._u8p), cand.shape[0],
                                                     _p(errs, _ffi._f64p)))
        return errs

    def score_candidates_device(self, palette, index, d_rgb5_ptr, n, d_errors_ptr, d_maps_ptr=0):
        self._chk(self._L.snesimage_score_candidates_device(self._c, palette, index, C.c_void_p(d_rgb5_ptr), n,
                                                            C.c_void_p(d_errors_ptr), C.c_void_p(d_maps_ptr or 0)))

    def remap_candidates_device(self, palette, index, d_rgb5_ptr, n, d_maps_ptr):
        """optimize() of every candidate without error(): n palette_maps into device memory (asynchronous)."""
        self._chk(self._L.snesimage_remap_candidates_device(self._c, palette, index, C.c_void_p(d_rgb5_ptr), n,
                                                            C.c_void_p(d_maps_ptr)))

    def step(self, method, palette, index, channel=0, seed=1, step_id=0, n_random=0):
        """optimize_palette_entry_{random,channel,nes} + lib.rs:906-910 -> (error, best rgb5)."""
        err = C.c_double(0)
        best = np.zeros(3, np.uint8)
        self._chk(self._L.snesimage_step(self._c, method, palette, index, channel, seed, step_id, n_random,
                                         C.byref(err), _p(best, _ffi._u8p)))
        return err.value, best

    def step_async(self, method, palette, index, channel=0, seed=1, step_id=0, n_random=0):
        self._chk(self._L.snesimage_step_async(self._c, method, palette, index, channel, seed, step_id, n_random))

    def last_step(self):
        err, k = C.c_double(0), C.c_int32(0)
        best = np.zeros(3, np.uint8)
        self._chk(self._L.snesimage_last_step(self._c, C.byref(err), _p(best, _ffi._u8p), C.byref(k)))
        return err.value, best, k.value

    def run_slots(self, n_calls, seed=1, first_step_id=0, state=(0, 0, 0, 0), window=0, want_log=True, n_random=0):
        """The reference's loop (lib.rs:888-933) for n_calls calls from scheduler state (palette, index, channel, step),
        speculatively several calls per launch (bit-identical to `step` per scheduled call).  Returns (log, state, stats):
        log[j] = (error, best_k, rgb5, changed) after call j."""
        st = [C.c_uint32(int(v)) for v in state]
        log = (_ffi.CallResult * n_calls)() if want_log else None
        stats = _ffi.RunStats()
        self._chk(self._L.snesimage_run_slots(self._c, n_calls, seed, first_step_id, C.byref(st[0]), C.byref(st[1]), C.byref(st[2]),
                                              C.byref(st[3]), int(n_random), int(window), log, C.byref(stats)))
        out = [(r.error, r.best_k, np.array(r.rgb5[:], np.uint8), int(r.changed)) for r in log] if want_log else None
        return out, tuple(v.value for v in st), {k: getattr(stats, k) for k in ("calls", "accepted", "windows", "voided", "scored", "useful")}

    def slots_reserve(self, n_slots):
        """Allocate the storage of windows of up to n_slots calls now instead of on first use."""
        self._chk(self._L.snesimage_slots_reserve(self._c, int(n_slots)))

    def slots_begin(self, n_slots, seed, first_step_id, state, n_random=0, shard_rank=0, shard_count=1, d_errors_ptr=0):
        """Phase 1 of a slot window -> (calls taken, candidates per call)."""
        taken, stride = C.c_uint32(0), C.c_uint32(0)
        self._chk(self._L.snesimage_slots_begin(self._c, n_slots, seed, first_step_id, state[0], state[1], state[2], state[3], n_random,
                                                shard_rank, shard_count, C.c_void_p(d_errors_ptr or 0), C.byref(taken), C.byref(stride)))
        return taken.value, stride.value

    def slots_commit(self, d_errors_ptr=0, n_taken=0):
        """Phase 2 -> (calls consumed, accepted flag, log of the consumed calls)."""
        used, acc = C.c_uint32(0), C.c_uint32(0)
        log = (_ffi.CallResult * max(1, n_taken))() if n_taken else None
        self._chk(self._L.snesimage_slots_commit(self._c, C.c_void_p(d_errors_ptr or 0), C.byref(used), C.byref(acc), log))
        out = [(r.error, r.best_k, np.array(r.rgb5[:], np.uint8), int(r.changed)) for r in log[:used.value]] if log else None
        return used.value, acc.value, out

    def step_begin(self, method, palette, index, channel, seed, step_id, n_total, shard_rank, shard_count,
                   d_errors_ptr):
        self._chk(self._L.snesimage_step_begin(self._c, method, palette, index, channel, seed, step_id, n_total,
                                               shard_rank, shard_count, C.c_void_p(d_errors_ptr)))

    def step_commit(self, d_errors_ptr):
        self._chk(self._L.snesimage_step_commit(self._c, C.c_void_p(d_errors_ptr)))

    # -- state ------------------------------------------------------------------------------------
    @property
    def tile_palettes(self):
        out = np.zeros(1024, np.uint8)
        self._chk(self._L.snesimage_get_tile_palettes(self._c, _p(out, _ffi._u8p)))
        return out

    @tile_palettes.setter
    def tile_palettes(self, v):
        v = np.ascontiguousarray(v, np.uint8).reshape(1024)
        self._chk(self._L.snesimage_set_tile_palettes(self._c, _p(v, _ffi._u8p)))

    @property
    def palette(self):
        out = np.zeros((self.sub_count * self.sub_size, 3), np.uint8)
        self._chk(self._L.snesimage_get_palette_rgb5(self._c, _p(out, _ffi._u8p)))
        return out

    @palette.setter
    def palette(self, v):
        v = np.ascontiguousarray(v, np.uint8).reshape(self.sub_count * self.sub_size, 3)
        self._chk(self._L.snesimage_set_palette_rgb5(self._c, _p(v, _ffi._u8p)))

    @property
    def palette_u16(self):
        out = np.zeros(self.sub_count * self.sub_size, np.uint16)
        self._chk(self._L.snesimage_get_palette_u16(self._c, _p(out, _ffi._u16p)))
        return out

    @property
    def palette_map(self):
        out = np.zeros((self.h, self.w), np.uint8)
        self._chk(self._L.snesimage_get_palette_map(self._c, _p(out, _ffi._u8p)))
        return out

    @palette_map.setter
    def palette_map(self, v):
        v = np.ascontiguousarray(v, np.uint8).reshape(self.h, self.w)
        self._chk(self._L.snesimage_set_palette_map(self._c, _p(v, _ffi._u8p)))

    def as_rgba(self):  # lib.rs:550
        out = np.zeros((self.h, self.w, 4), np.uint8)
        self._chk(self._L.snesimage_as_rgba(self._c, _p(out, _ffi._u8p)))
        return out

    def as_json(self):  # lib.rs:579 + .to_string() (:1002)
        need = self._L.snesimage_as_json(self._c, None, 0)
        if need < 0:
            raise SnesImageError(int(need), self._L.snesimage_last_error().decode())
        buf = C.create_string_buffer(int(need))
        self._L.snesimage_as_json(self._c, buf, need)
        return buf.value.decode()


def debug_math(op, x, y=None, device=0):
    """Evaluate the kernels' deterministic math on the GPU (bit-parity tests)."""
    L = _ffi.load()
    x = np.ascontiguousarray(x, np.float32)
    yy = x if y is None else np.ascontiguousarray(y, np.float32)
    per_in = 3 if op in (5, 6) else 1
    n = x.size // per_in
    out = np.zeros(n * (3 if op == 6 else 1), np.float32)
    rc = L.snesimage_debug_math(device, op, _p(x, _ffi._f32p), _p(yy, _ffi._f32p), n, _p(out, _ffi._f32p))
    if rc != 0:
        raise SnesImageError(rc, L.snesimage_last_error().decode())
    return out
