"""snesimage_amd — MI355X (gfx950) implementation of snesimage's palette-optimizer hot path.

The product is the C-ABI shared library `libsnesimage_hip.so` (include/snesimage_hip.h); this
package is the thin Python host side used by the tests and the benchmark.  Importing `api`
loads the library and fails loudly when it has not been built — there is no CPU fallback.
"""
from .api import (DITHER, METHOD_CHANNEL, METHOD_NES, METHOD_RANDOM, NES, PERCEPTUAL, OptimizedImage,  # noqa: F401
                  SnesImageError, debug_math, random_candidates, schedule)
