"""The C-ABI library loads on a machine without a GPU, exports exactly what include/snesimage_hip.h
declares, and fails loudly (no silent CPU fallback) when no HIP device is usable."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "snesimage_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snesimage_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from snesimage_amd import _ffi
    bound = sorted(n for n, _, _ in _ffi.SIGNATURES)
    assert bound == _declared()


def test_library_exports_every_declared_symbol():
    from snesimage_amd import _ffi
    lib = _ffi.load()
    for name in _declared():
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.snesimage_version()


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "snesimage_hip.h")).read()
    for fn in ("snesimage_initialize_tiles", "snesimage_recalculate_palettes", "snesimage_optimize", "snesimage_error"):
        line = next(l for l in text.splitlines() if fn + "(" in l)
        assert "lib.rs:" in line, fn


def test_host_helpers_match_oracle(O):
    import snesimage_amd as S
    for seed, step, n in [(1, 0, 64), (7, 123, 5), (2 ** 63 + 5, 2 ** 40, 33)]:
        assert np.array_equal(S.random_candidates(seed, step, n), O.random_candidates(seed, step, n))
    c = S.random_candidates(3, 9, 4096)
    assert c.max() == 31 and c.min() == 0
    for (cnt, size, nes) in [(2, 3, False), (8, 15, False), (1, 7, False), (2, 3, True)]:
        assert S.schedule(cnt, size, 500, nes) == O.schedule(cnt, size, 500, nes)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_gpu(img256):
    import snesimage_amd as S
    with pytest.raises(S.SnesImageError) as e:
        S.OptimizedImage(img256, 8, 15)
    assert e.value.code == -2 and "hip" in str(e.value).lower()


def test_argument_validation(img256):
    from snesimage_amd import _ffi
    lib = _ffi.load()
    ctx = C.c_void_p()
    p = img256.ctypes.data_as(_ffi._u8p)
    assert lib.snesimage_create(p, 128, 256, 8, 15, 0, 0, C.byref(ctx)) == -1 and b"256" in lib.snesimage_last_error()
    assert lib.snesimage_create(p, 256, 100, 8, 15, 0, 0, C.byref(ctx)) == -1
    assert lib.snesimage_create(p, 256, 256, 20, 15, 0, 0, C.byref(ctx)) == -1
    assert lib.snesimage_create(p, 256, 256, 8, 15, 0, -1, C.byref(ctx)) == -1 and b"no CPU path" in lib.snesimage_last_error()
    assert lib.snesimage_create(None, 256, 256, 8, 15, 0, 0, C.byref(ctx)) == -1
    assert lib.snesimage_optimize(None) == -1


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or open anything under oracle/."""
    pkg = os.path.join(ROOT, "snesimage_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in text and "libsnes_oracle" not in text and "snes_oracle.h" not in text, f
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libsnesimage_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_library_in_the_tree_is_built_from_the_sources_in_the_tree():
    """snesimage_version() carries the hash of the sources the library was built from (csrc/Makefile: SRC_HASH); bench.py
    withholds counter figures stamped with another build's hash.  A library left behind by an earlier build — a header
    comment edited, `make` not run — would void every committed counter summary at the next build: keep them in step."""
    import glob
    import hashlib
    from snesimage_amd import _ffi
    lib = _ffi.load()
    v = lib.snesimage_version().decode()
    assert "src:" in v
    csrc = os.path.join(ROOT, "snesimage_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.inc")), key=os.path.basename)
    files += [os.path.join(ROOT, "include", "snesimage_hip.h"), os.path.join(ROOT, "include", "ssimulacra2_constants.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    assert v.split("src:")[1].strip() == h.hexdigest()[:16], "snesimage_amd/libsnesimage_hip.so is older than its sources: run make -C snesimage_amd/csrc"
