"""ctypes binding of the CPU oracle (oracle/libsnes_oracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The shipped package (snesimage_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsnes_oracle.so")

DITHER, PERCEPTUAL, NES = 1, 2, 4

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle if the shared object is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("snes_oracle.cpp", "snes_oracle.h", "det_math.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_cache_source.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_blur_mode.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_variant.argtypes = [C.c_void_p, C.c_int]
        for name in ("oracle_initialize_tiles", "oracle_recalculate_palettes", "oracle_optimize"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.oracle_error.argtypes = [C.c_void_p, _f64p]
        L.oracle_reassign_tiles.argtypes = [C.c_void_p, _u32p]
        L.oracle_score_candidates.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, _f64p, _u8p]
        L.oracle_step.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.c_uint64, C.c_uint32, _f64p, _u8p]
        L.oracle_random_candidates.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, _u8p]
        L.oracle_schedule_next.argtypes = [C.c_uint32, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, C.c_int]
        for name in ("oracle_get_tile_palettes", "oracle_set_tile_palettes", "oracle_get_palette_rgb5",
                     "oracle_set_palette_rgb5", "oracle_get_palette_map", "oracle_set_palette_map",
                     "oracle_as_rgba"):
            getattr(L, name).argtypes = [C.c_void_p, _u8p]
        L.oracle_get_palette_u16.argtypes = [C.c_void_p, _u16p]
        L.oracle_as_json.restype = C.c_int64
        L.oracle_as_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.oracle_distance_red_mean.restype = C.c_double
        L.oracle_distance_red_mean.argtypes = [_u8p, _u8p]
        L.oracle_distance_cielab.restype = C.c_double
        L.oracle_distance_cielab.argtypes = [_u8p, _u8p]
        L.oracle_red_mean_key.restype = C.c_uint32
        L.oracle_red_mean_key.argtypes = [_u8p, _u8p]
        L.oracle_srgb8_to_lab.argtypes = [_u8p, _f32p]
        L.oracle_ciede2000.restype = C.c_float
        L.oracle_ciede2000.argtypes = [_f32p, _f32p]
        L.oracle_lab_to_srgb8.argtypes = [_f64p, _u8p]
        L.oracle_snes_as_rgba.argtypes = [_u8p, _u8p]
        L.oracle_snes_as_u16.restype = C.c_uint16
        L.oracle_snes_as_u16.argtypes = [_u8p]
        L.oracle_nes_color.argtypes = [C.c_uint32, _u8p]
        L.oracle_new_nes_only.argtypes = [_u8p, C.c_int, _u8p]
        L.oracle_closest_color_index.restype = C.c_uint32
        L.oracle_closest_color_index.argtypes = [_u8p, C.c_uint32, _f64p, C.c_int]
        L.oracle_kmeans.argtypes = [_f64p, C.c_uint32, C.c_uint32, _f64p, _u32p, _u32p]
        L.oracle_ssimulacra2_rgba.argtypes = [_u8p, _u8p, C.c_uint32, C.c_uint32, C.c_int, _f64p]
        L.oracle_blur_plane.argtypes = [_f32p, _f32p, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_blur_constants.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_det_math.argtypes = [C.c_int, _f32p, _f32p, C.c_uint32, _f32p]
        L.oracle_synth_image.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _u8p]
        L.oracle_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def synth_image(seed, w=256, h=256, variant=0):
    out = np.zeros((h, w, 4), np.uint8)
    lib().oracle_synth_image(seed, w, h, variant, _p(out, _u8p))
    return out


def random_candidates(seed, step_id, n):
    out = np.zeros((n, 3), np.uint8)
    lib().oracle_random_candidates(seed, step_id, n, _p(out, _u8p))
    return out


def distance_red_mean(c1, c2):
    a, b = _u8(c1), _u8(c2)
    return lib().oracle_distance_red_mean(_p(a, _u8p), _p(b, _u8p))


def red_mean_key(c1, c2):
    a, b = _u8(c1), _u8(c2)
    return lib().oracle_red_mean_key(_p(a, _u8p), _p(b, _u8p))


def distance_cielab(c1, c2):
    a, b = _u8(c1), _u8(c2)
    return lib().oracle_distance_cielab(_p(a, _u8p), _p(b, _u8p))


def srgb8_to_lab(rgb):
    a = _u8(rgb)
    out = np.zeros(3, np.float32)
    lib().oracle_srgb8_to_lab(_p(a, _u8p), _p(out, _f32p))
    return out


def ciede2000(lab1, lab2):
    a = np.ascontiguousarray(lab1, np.float32)
    b = np.ascontiguousarray(lab2, np.float32)
    return float(lib().oracle_ciede2000(_p(a, _f32p), _p(b, _f32p)))


def lab_to_srgb8(lab):
    a = np.ascontiguousarray(lab, np.float64)
    out = np.zeros(3, np.uint8)
    lib().oracle_lab_to_srgb8(_p(a, _f64p), _p(out, _u8p))
    return out


def snes_as_rgba(rgb5):
    a = _u8(rgb5)
    out = np.zeros(4, np.uint8)
    lib().oracle_snes_as_rgba(_p(a, _u8p), _p(out, _u8p))
    return out


def snes_as_u16(rgb5):
    a = _u8(rgb5)
    return int(lib().oracle_snes_as_u16(_p(a, _u8p)))


def nes_color(index):
    out = np.zeros(3, np.uint8)
    lib().oracle_nes_color(index, _p(out, _u8p))
    return out


def new_nes_only(rgb5, cielab=False):
    a = _u8(rgb5)
    out = np.zeros(3, np.uint8)
    lib().oracle_new_nes_only(_p(a, _u8p), int(cielab), _p(out, _u8p))
    return out


def closest_color_index(entries_rgb5, target, cielab=False):
    e = _u8(entries_rgb5).reshape(-1, 3)
    t = np.ascontiguousarray(target, np.float64)
    return int(lib().oracle_closest_color_index(_p(e, _u8p), e.shape[0], _p(t, _f64p), int(cielab)))


def kmeans(points, k):
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    centres = np.zeros((k, 3), np.float64)
    assign = np.zeros(pts.shape[0], np.uint32)
    iters = C.c_uint32(0)
    rc = lib().oracle_kmeans(_p(pts, _f64p), pts.shape[0], k, _p(centres, _f64p), _p(assign, _u32p), C.byref(iters))
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return centres, assign, iters.value


def ssimulacra2_rgba(src, dst, blur_mode=0):
    a, b = _u8(src), _u8(dst)
    h, w = a.shape[0], a.shape[1]
    out = C.c_double(0)
    rc = lib().oracle_ssimulacra2_rgba(_p(a, _u8p), _p(b, _u8p), w, h, blur_mode, C.byref(out))
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())
    return out.value


def blur_plane(plane, mode=0):
    a = np.ascontiguousarray(plane, np.float32)
    out = np.zeros_like(a)
    lib().oracle_blur_plane(_p(a, _f32p), _p(out, _f32p), a.shape[1], a.shape[0], mode)
    return out


def blur_constants():
    n2, d1, fir = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().oracle_blur_constants(_p(n2, _f32p), _p(d1, _f32p), _p(fir, _f32p))
    return n2, d1, fir


def det_math(op, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.zeros_like(x)
    lib().oracle_det_math(op, _p(x, _f32p), _p(y, _f32p), x.size, _p(out, _f32p))
    return out


def schedule(sub_count, sub_size, n_calls, nes=False):
    """Replay lib.rs:881-933: list of (method, palette, index, channel, step) for n_calls calls."""
    p, i, ch, st, m = (C.c_uint32(0) for _ in range(5))
    out = []
    for _ in range(n_calls):
        cur = (p.value, i.value, ch.value, st.value)
        lib().oracle_schedule_next(sub_count, sub_size, C.byref(p), C.byref(i), C.byref(ch), C.byref(st),
                                   C.byref(m), int(nes))
        out.append((m.value,) + cur)
    return out


class OracleImage:
    """Mirror of the reference's `OptimizedImage` (lib.rs:33-626) over the CPU oracle."""

    def __init__(self, rgba, sub_count, sub_size, dither=False, perceptual=False, nes=False, cache_source=True):
        rgba = _u8(rgba)
        self.h, self.w = rgba.shape[0], rgba.shape[1]
        self.sub_count, self.sub_size = sub_count, sub_size
        flags = (DITHER if dither else 0) | (PERCEPTUAL if perceptual else 0) | (NES if nes else 0)
        self._L = lib()
        self._c = self._L.oracle_create(_p(rgba, _u8p), self.w, self.h, sub_count, sub_size, flags)
        if not self._c:
            raise ValueError(self._L.oracle_last_error().decode())
        self._L.oracle_set_cache_source(self._c, int(cache_source))

    def close(self):
        if self._c:
            self._L.oracle_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self._L.oracle_last_error().decode())

    def set_blur_mode(self, mode):
        self._L.oracle_set_blur_mode(self._c, mode)

    def set_variant(self, bits):
        """What-if variants of the unpinned third-party arithmetic (snes_oracle.h: oracle_set_variant); 0 = the restatement."""
        self._L.oracle_set_variant(self._c, bits)

    def set_cache_source(self, on):
        self._L.oracle_set_cache_source(self._c, int(on))

    def initialize_tiles(self):
        self._chk(self._L.oracle_initialize_tiles(self._c))

    def recalculate_palettes(self):
        self._chk(self._L.oracle_recalculate_palettes(self._c))

    def optimize(self):
        self._chk(self._L.oracle_optimize(self._c))

    def error(self):
        out = C.c_double(0)
        self._chk(self._L.oracle_error(self._c, C.byref(out)))
        return out.value

    def reassign_tiles(self):
        """Move every tile to the subpalette that reproduces it best (not in the reference: TODO.md:36-37); tiles moved."""
        moved = C.c_uint32(0)
        self._chk(self._L.oracle_reassign_tiles(self._c, C.byref(moved)))
        return moved.value

    def score_candidates(self, palette, index, rgb5, want_maps=False):
        cand = _u8(rgb5).reshape(-1, 3)
        n = cand.shape[0]
        errs = np.zeros(n, np.float64)
        maps = np.zeros((n, self.h, self.w), np.uint8) if want_maps else None
        self._chk(self._L.oracle_score_candidates(self._c, palette, index, _p(cand, _u8p), n, _p(errs, _f64p),
                                                  _p(maps, _u8p) if want_maps else None))
        return (errs, maps) if want_maps else errs

    def step(self, method, palette, index, channel=0, seed=1, step_id=0, n_random=0):
        err = C.c_double(0)
        best = np.zeros(3, np.uint8)
        self._chk(self._L.oracle_step(self._c, method, palette, index, channel, seed, step_id, n_random,
                                      C.byref(err), _p(best, _u8p)))
        return err.value, best

    @property
    def tile_palettes(self):
        out = np.zeros(1024, np.uint8)
        self._L.oracle_get_tile_palettes(self._c, _p(out, _u8p))
        return out

    @tile_palettes.setter
    def tile_palettes(self, v):
        v = _u8(v).reshape(1024)
        self._L.oracle_set_tile_palettes(self._c, _p(v, _u8p))

    @property
    def palette(self):
        out = np.zeros((self.sub_count * self.sub_size, 3), np.uint8)
        self._L.oracle_get_palette_rgb5(self._c, _p(out, _u8p))
        return out

    @palette.setter
    def palette(self, v):
        v = _u8(v).reshape(self.sub_count * self.sub_size, 3)
        self._L.oracle_set_palette_rgb5(self._c, _p(v, _u8p))

    @property
    def palette_u16(self):
        out = np.zeros(self.sub_count * self.sub_size, np.uint16)
        self._L.oracle_get_palette_u16(self._c, _p(out, _u16p))
        return out

    @property
    def palette_map(self):
        out = np.zeros((self.h, self.w), np.uint8)
        self._L.oracle_get_palette_map(self._c, _p(out, _u8p))
        return out

    @palette_map.setter
    def palette_map(self, v):
        v = _u8(v).reshape(self.h, self.w)
        self._L.oracle_set_palette_map(self._c, _p(v, _u8p))

    def as_rgba(self):
        out = np.zeros((self.h, self.w, 4), np.uint8)
        self._L.oracle_as_rgba(self._c, _p(out, _u8p))
        return out

    def as_json(self):
        need = self._L.oracle_as_json(self._c, None, 0)
        buf = C.create_string_buffer(need)
        self._L.oracle_as_json(self._c, buf, need)
        return buf.value.decode()
