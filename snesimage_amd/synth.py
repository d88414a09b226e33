"""Synthetic 256xH RGBA8 test images: smooth gradient + bounded noise from splitmix64.

pixel (x, y), raster order, z = next splitmix64 value:
    r = (x + (z & 63)) mod 256, g = (y + ((z >> 8) & 63)) mod 256, b = ((x + y) // 2 + ((z >> 16) & 63)) mod 256
variant 1 clears alpha in the square [96,160) x [96,160) (exercises the transparent-pixel rules).
Image i of a batch uses seed 0x5EED0000 + i.
"""
import numpy as np

BASE_SEED = 0x5EED0000
_M = (1 << 64) - 1


def _splitmix64_stream(seed, n):
    k = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_image(seed=BASE_SEED, w=256, h=256, variant=0):
    z = _splitmix64_stream(seed, w * h).reshape(h, w)
    x = np.arange(w, dtype=np.uint64)[None, :]
    y = np.arange(h, dtype=np.uint64)[:, None]
    out = np.zeros((h, w, 4), np.uint8)
    out[..., 0] = ((x + (z & np.uint64(63))) & np.uint64(255)).astype(np.uint8)
    out[..., 1] = ((y + ((z >> np.uint64(8)) & np.uint64(63))) & np.uint64(255)).astype(np.uint8)
    out[..., 2] = (((x + y) // np.uint64(2) + ((z >> np.uint64(16)) & np.uint64(63))) & np.uint64(255)).astype(np.uint8)
    out[..., 3] = 255
    if variant == 1:
        out[96:min(160, h), 96:160, 3] = 0
    return out
