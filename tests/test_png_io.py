"""PNG input of the headless driver (snesimage_amd/csrc/png_io.hpp): `image::open(..).into_rgba8()` (src/lib.rs:836)
restated for PNG.  The files are produced here by a small independent encoder (every colour type, bit depth, row filter,
Adam7) and the expected RGBA8 is computed from the samples with numpy; Pillow, when importable, is a second opinion for
the <= 8-bit cases.  `--decode-only` never touches the GPU."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "snesimage_amd", "snesimage_cli")
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def pack_rows(samples, depth):
    """samples (h, w*ch) ints -> list of packed byte rows"""
    h, n = samples.shape
    rows = []
    for y in range(h):
        if depth == 8:
            rows.append(bytes(samples[y].astype(np.uint8)))
        elif depth == 16:
            rows.append(samples[y].astype(">u2").tobytes())
        else:
            per = 8 // depth
            pad = (-n) % per
            v = np.concatenate([samples[y], np.zeros(pad, samples.dtype)]).reshape(-1, per)
            out = np.zeros(len(v), np.uint32)
            for k in range(per):
                out |= v[:, k].astype(np.uint32) << (8 - depth * (k + 1))
            rows.append(bytes(out.astype(np.uint8)))
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, rng):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for row in rows:
        ft = int(rng.integers(0, 5))
        f = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            f[i] = (v - pred) & 0xFF
        out.append(ft)
        out += f
        prev = row
    return bytes(out)


def encode_png(samples, ctype, depth, interlace=False, plte=None, trns=None, seed=0, idat_split=1):
    """samples: (h, w, channels) int array of raw `depth`-bit samples (palette indices for ctype 3)."""
    rng = np.random.default_rng(seed)
    h, w, ch = samples.shape
    assert ch == CHANNELS[ctype]
    bpp = max(1, ch * depth // 8)
    raw = b""
    for (x0, y0, dx, dy) in (ADAM7 if interlace else [(0, 0, 1, 1)]):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    z = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))  # ancillary chunks are skipped
    if plte is not None:
        out += chunk(b"PLTE", bytes(np.asarray(plte, np.uint8).reshape(-1)))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    step = max(1, len(z) // idat_split)
    for i in range(0, len(z), step):
        out += chunk(b"IDAT", z[i:i + step])
    return out + chunk(b"IEND", b"")


def to8(v, depth):
    v = v.astype(np.uint32)
    return {1: v * 255, 2: v * 85, 4: v * 17, 8: v, 16: (v + 128) // 257}[depth].astype(np.uint8)


def expected_rgba(samples, ctype, depth, plte=None, trns=None):
    h, w, _ = samples.shape
    out = np.zeros((h, w, 4), np.uint8)
    out[..., 3] = 255
    if ctype == 0:
        out[..., :3] = to8(samples[..., 0], depth)[..., None]
        if trns is not None:
            key = struct.unpack(">H", bytes(trns))[0] & ((1 << depth) - 1)
            out[..., 3] = np.where(samples[..., 0] == key, 0, 255)
    elif ctype == 2:
        out[..., :3] = to8(samples, depth)
        if trns is not None:
            key = np.array(struct.unpack(">HHH", bytes(trns))) & ((1 << depth) - 1)
            out[..., 3] = np.where((samples == key).all(-1), 0, 255)
    elif ctype == 3:
        pal = np.asarray(plte, np.uint8)
        out[..., :3] = pal[samples[..., 0]]
        if trns is not None:
            a = np.full(256, 255, np.uint8)
            a[:len(trns)] = np.frombuffer(bytes(trns), np.uint8)
            out[..., 3] = a[samples[..., 0]]
    elif ctype == 4:
        out[..., :3] = to8(samples[..., 0], depth)[..., None]
        out[..., 3] = to8(samples[..., 1], depth)
    else:
        out[...] = to8(samples, depth)
    return out


def decode_with_cli(tmp_path, png_bytes, name="t"):
    src = tmp_path / (name + ".png")
    dst = tmp_path / (name + ".rgba")
    src.write_bytes(png_bytes)
    r = subprocess.run([CLI, str(src), str(dst), "--decode-only"], capture_output=True, text=True, timeout=60)
    return r, dst


CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("ctype,depth", CASES)
@pytest.mark.parametrize("interlace", [False, True])
def test_every_colour_type_and_depth(tmp_path, ctype, depth, interlace):
    assert os.path.exists(CLI), "build with make -C snesimage_amd/csrc"
    rng = np.random.default_rng(ctype * 100 + depth + (50 if interlace else 0))
    w, h = 256, 13  # odd height: ragged Adam7 passes; width what the driver accepts
    samples = rng.integers(0, 1 << depth, size=(h, w, CHANNELS[ctype]), dtype=np.int64)
    plte = rng.integers(0, 256, size=(1 << depth, 3)) if ctype == 3 else None
    trns = None
    if ctype == 3:
        trns = rng.integers(0, 256, size=max(1, (1 << depth) // 2)).astype(np.uint8).tobytes()  # shorter than the palette: rest opaque
    elif ctype == 0:
        trns = struct.pack(">H", int(samples[0, 0, 0]))
    elif ctype == 2:
        trns = struct.pack(">HHH", *[int(v) for v in samples[1, 2]])
    png = encode_png(samples, ctype, depth, interlace, plte, trns, seed=depth, idat_split=3)
    r, dst = decode_with_cli(tmp_path, png)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split()[-2:] == ["256", str(h)]
    got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(h, w, 4)
    assert np.array_equal(got, expected_rgba(samples, ctype, depth, plte, trns))


def test_sixteen_bit_reduction_is_round_to_nearest(tmp_path):
    """image 0.25: u16 -> u8 is (c + 128) / 257 == round(c * 255 / 65535), for all 65,536 values."""
    v = np.arange(65536, dtype=np.int64).reshape(256, 256, 1)
    r, dst = decode_with_cli(tmp_path, encode_png(v.transpose(1, 0, 2).copy(), 0, 16))
    assert r.returncode == 0
    got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(256, 256, 4)[..., 0].T.reshape(-1)
    want = np.floor(np.arange(65536) * 255.0 / 65535.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got, want)


def test_pillow_agrees_when_available(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    rgba = rng.integers(0, 256, size=(16, 256, 4), dtype=np.uint8)
    for mode in ("RGBA", "RGB", "L", "LA", "P", "1"):
        im = Image.fromarray(rgba, "RGBA").convert(mode)
        p = tmp_path / ("pil_%s.png" % mode)
        im.save(p)
        want = np.asarray(Image.open(p).convert("RGBA"))
        r, dst = decode_with_cli(tmp_path, p.read_bytes(), "pil_" + mode)
        assert r.returncode == 0, mode + r.stdout
        assert np.array_equal(np.frombuffer(dst.read_bytes(), np.uint8).reshape(16, 256, 4), want), mode


def test_errors_follow_the_reference_convention(tmp_path):
    rng = np.random.default_rng(9)
    good = encode_png(rng.integers(0, 256, size=(8, 256, 3)), 2, 8)
    bad_crc = bytearray(good)
    bad_crc[40] ^= 1
    r, dst = decode_with_cli(tmp_path, bytes(bad_crc), "crc")
    assert r.returncode == 1 and "Error running application:" in r.stdout and not dst.exists()
    r, _ = decode_with_cli(tmp_path, good[:len(good) // 2], "trunc")
    assert r.returncode == 1 and "Error running application:" in r.stdout
    r, _ = decode_with_cli(tmp_path, encode_png(rng.integers(0, 256, size=(8, 64, 3)), 2, 8), "narrow")
    assert r.returncode == 1 and "Image size must be 256x256" in r.stdout  # src/lib.rs:838-840
    # a file without the PNG signature is read as raw RGBA8 rows of 256 pixels
    r, dst = decode_with_cli(tmp_path, bytes(256 * 4 * 8), "raw")
    assert r.returncode == 0 and dst.read_bytes() == bytes(256 * 4 * 8)
