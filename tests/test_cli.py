"""Headless driver (snesimage_amd/csrc/cli.cpp): same command-line surface as the reference's clap
struct (src/config.rs:3-31), same error convention (src/main.rs:16-19), same JSON (src/lib.rs:579-625)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "snesimage_amd", "snesimage_cli")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300)


def test_cli_usage_errors():
    assert os.path.exists(CLI), "build with make -C snesimage_amd/csrc"
    r = run("--help")
    assert r.returncode == 0
    for flag in ("--subpalette-count", "--subpalette-size", "--dither", "--perceptual-palettes", "--nes", "-c", "-s", "-d"):
        assert flag in r.stderr
    r = run("only_one_positional")
    assert r.returncode == 2 and "required arguments" in r.stderr
    r = run("a", "b", "--bogus")
    assert r.returncode == 2 and "unexpected argument" in r.stderr
    r = run("-V")
    assert r.returncode == 0 and "snesimage 0.1.1" in r.stdout


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_cli_fails_loudly_without_gpu(tmp_path):
    r = run("synth:1", str(tmp_path / "o.json"), "-c", "8", "-s", "15")
    assert r.returncode == 1
    assert "Using source image: synth:1" in r.stdout and "Error running application:" in r.stdout
    assert not (tmp_path / "o.json").exists()


def test_cli_rejects_missing_file(tmp_path):
    r = run(str(tmp_path / "nope.rgba"), str(tmp_path / "o.json"))
    assert r.returncode == 1 and "Error running application:" in r.stdout


def test_cli_resume_rejects_subpalettes_the_json_cannot_hold(tmp_path):
    """as_json keeps 15 colours per subpalette (lib.rs:583-593): --resume with a larger -s would read past them."""
    (tmp_path / "prev.json").write_text(json.dumps({"palette": [0] * 16, "tile_palettes": [0] * 1024, "tiles": []}))
    r = run("synth:1", str(tmp_path / "o.json"), "-c", "1", "-s", "16", "--resume", str(tmp_path / "prev.json"))
    assert r.returncode == 1 and "--subpalette-size <= 15" in r.stdout
    assert not (tmp_path / "o.json").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("extra,count,size,flags", [([], 8, 15, {}), (["-d"], 2, 3, {"dither": True}), (["--nes"], 2, 3, {"nes": True})])
def test_cli_end_to_end_matches_oracle(tmp_path, O, extra, count, size, flags):
    """init -> cluster -> N optimizer calls in the reference's slot order -> JSON, byte for byte."""
    from snesimage_amd.synth import synth_image
    seed_img, calls, ncand = 0x5EED0002, 7, 12
    img = synth_image(seed_img)
    src = tmp_path / "in.rgba"
    src.write_bytes(img.tobytes())
    out = tmp_path / "out.json"
    r = run(str(src), str(out), "-c", str(count), "-s", str(size), "--calls", str(calls), "--candidates", str(ncand), "--seed", "5", *extra)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Finished assigning initial tiles" in r.stdout and "Writing output to" in r.stdout and "Current Error:" in r.stdout
    o = O.OracleImage(img, count, size, **flags)
    o.initialize_tiles()
    o.recalculate_palettes()
    last = None
    for i, (method, p, idx, ch, _) in enumerate(O.schedule(count, size, calls, nes=bool(flags.get("nes")))):
        last, _ = o.step(method, p, idx, ch, 5, i, ncand if method == 0 else 0)
    text = out.read_text()
    assert text == o.as_json()
    doc = json.loads(text)
    assert len(doc["tiles"]) == 1024 and len(doc["palette"]) == 16 * count
    # the logged error is Rust's `{}` of the f64: shortest round-trip decimal
    logged = [l.split("Current Error: ")[1] for l in r.stdout.splitlines() if "Current Error: " in l]
    assert abs(float(logged[-1]) - last) <= 1e-9 * abs(last)
    # synth: source gives the same image as the raw file
    out2 = tmp_path / "out2.json"
    r2 = run("synth:%d" % seed_img, str(out2), "-c", str(count), "-s", str(size), "--calls", str(calls), "--candidates", str(ncand), "--seed", "5", *extra)
    assert r2.returncode == 0 and out2.read_text() == text


@pytest.mark.gpu
def test_cli_reassign_tiles_matches_oracle(tmp_path, O):
    """--reassign-tiles 1 (not in the reference: TODO.md:36-37): after every sweep over the slots (2 x 3 entries: calls 6, 12,
    ...) each tile moves to the subpalette that reproduces it best; the JSON is the oracle's doing the same, and without
    the flag the output is untouched by the feature."""
    from snesimage_amd.synth import synth_image
    seed_img, calls, ncand, count, size = 0x5EED0004, 14, 12, 2, 3
    img = synth_image(seed_img)
    out = tmp_path / "out.json"
    r = run("synth:%d" % seed_img, str(out), "-c", str(count), "-s", str(size), "--calls", str(calls), "--candidates", str(ncand), "--seed", "5",
            "--reassign-tiles", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    o = O.OracleImage(img, count, size)
    o.initialize_tiles()
    o.recalculate_palettes()
    moved = []
    sched = O.schedule(count, size, calls + 1)
    for i, (method, p, idx, ch, step) in enumerate(sched[:calls]):
        o.step(method, p, idx, ch, 5, i, ncand if method == 0 else 0)
        if sched[i + 1][4] != step:  # the sweep counter advanced
            moved.append(o.reassign_tiles())
    assert len(moved) == 2 and moved[0] > 0
    assert [int(l.split("Reassigned ")[1].split()[0]) for l in r.stdout.splitlines() if "Reassigned " in l] == moved
    assert out.read_text() == o.as_json()
    plain = tmp_path / "plain.json"
    r2 = run("synth:%d" % seed_img, str(plain), "-c", str(count), "-s", str(size), "--calls", str(calls), "--candidates", str(ncand), "--seed", "5")
    assert r2.returncode == 0 and "Reassigned" not in r2.stdout and plain.read_text() != out.read_text()


def read_plain_png(data):
    """Decoder for what the driver writes: RGBA8, filter 0 on every row."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) & 0xFFFFFFFF == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if kind == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 6)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
def test_cli_png_source_and_preview(tmp_path):
    """A PNG source gives the JSON of the same pixels as a raw file; --preview shows source | as_rgba() of the result."""
    from snesimage_amd.synth import synth_image
    from test_png_io import encode_png
    img = synth_image(0x5EED0004, 256, 64, 1)  # with a transparent square
    raw, png = tmp_path / "in.rgba", tmp_path / "in.png"
    raw.write_bytes(img.tobytes())
    png.write_bytes(encode_png(img.astype(np.int64), 6, 8, interlace=True, seed=4))
    args = ["-c", "4", "-s", "7", "--calls", "5", "--candidates", "10", "--seed", "3"]
    r1 = run(str(raw), str(tmp_path / "a.json"), *args)
    r2 = run(str(png), str(tmp_path / "b.json"), *args, "--preview", str(tmp_path / "p.png"))
    assert r1.returncode == 0 and r2.returncode == 0, r2.stdout + r2.stderr
    assert (tmp_path / "a.json").read_text() == (tmp_path / "b.json").read_text()
    both = read_plain_png((tmp_path / "p.png").read_bytes())
    assert both.shape == (64, 512, 4) and np.array_equal(both[:, :256], img)
    # the right half is the reconstruction of the JSON: tiles -> palette -> 8-bit expansion (lib.rs:550-577, 662-669)
    doc = json.loads((tmp_path / "b.json").read_text())
    pal = np.array(doc["palette"], np.uint32).reshape(-1, 16)
    tiles = np.array(doc["tiles"]).reshape(-1, 8, 8)
    tp = np.array(doc["tile_palettes"])
    want = np.zeros((64, 256, 4), np.uint8)
    for t in range(8 * 32):
        ty, tx = divmod(t, 32)
        v = tiles[t]
        c = pal[tp[t]][v]
        r5, g5, b5 = c & 31, (c >> 5) & 31, (c >> 10) & 31
        blk = np.stack([r5 * 8 + r5 // 4, g5 * 8 + g5 // 4, b5 * 8 + b5 // 4, np.full_like(r5, 255)], -1).astype(np.uint8)
        blk[v == 0] = 0
        want[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = blk
    assert np.array_equal(both[:, 256:], want)


@pytest.mark.gpu
def test_cli_resume_continues_from_a_previous_output(tmp_path):
    """--resume: palette and tile palettes come from an earlier JSON, the tiles from optimize() on them — so resuming with
    no further calls reproduces the file, and a bad file is an error in the reference's convention."""
    a, b, c = tmp_path / "a.json", tmp_path / "b.json", tmp_path / "c.json"
    base = ["synth:1592590339", "-c", "4", "-s", "7"]
    assert run(*base[:1], str(a), *base[1:], "--calls", "6", "--candidates", "12").returncode == 0
    r = run(*base[:1], str(b), *base[1:], "--resume", str(a))
    assert r.returncode == 0 and "Resumed from" in r.stdout and "Finished assigning initial tiles" not in r.stdout
    assert b.read_text() == a.read_text()
    r = run(*base[:1], str(c), *base[1:], "--resume", str(a), "--calls", "4", "--candidates", "12")
    assert r.returncode == 0 and json.loads(c.read_text())["tile_palettes"] == json.loads(a.read_text())["tile_palettes"]
    r = run(*base[:1], str(c), "-c", "2", "-s", "7", "--resume", str(a))  # 4 subpalettes in the file, 2 asked for
    assert r.returncode == 1 and "Error running application:" in r.stdout
    (tmp_path / "junk.json").write_text('{"palette":[[1,2]],"tile_palettes":[]}')
    r = run(*base[:1], str(c), *base[1:], "--resume", str(tmp_path / "junk.json"))
    assert r.returncode == 1 and "Error running application:" in r.stdout


@pytest.mark.gpu
def test_cli_devices_flag_gives_the_single_device_result(tmp_path):
    """--devices shards every call over the listed GPUs through the library's RCCL group; with the one device of this box
    the output must equal the plain run byte for byte."""
    a, b = tmp_path / "a.json", tmp_path / "b.json"
    args = ["synth:1592590340", "-c", "4", "-s", "7", "--calls", "8", "--candidates", "20", "--seed", "9"]
    r1 = run(args[0], str(a), *args[1:])
    r2 = run(args[0], str(b), *args[1:], "--devices", "0")
    assert r1.returncode == 0 and r2.returncode == 0, r2.stdout + r2.stderr
    assert "Sharding candidates over 1 device(s)" in r2.stdout and a.read_text() == b.read_text()
    # the reference's 64 candidates per call: slot windows, sharded by calls through snesimage_group_run_slots
    args = ["synth:1592590340", "-c", "4", "-s", "7", "--calls", "40", "--seed", "9"]
    r1 = run(args[0], str(a), *args[1:])
    r2 = run(args[0], str(b), *args[1:], "--devices", "0")
    assert r1.returncode == 0 and r2.returncode == 0, r2.stdout + r2.stderr
    assert "launch sets" in r2.stdout and a.read_text() == b.read_text()
    # the group starts from the state recalculate_palettes leaves (a palette_map that is not optimize() of the palette): only
    # the first call is stepped alone, the rest go through windows — fewer launch sets than calls, as in the plain run
    import re
    m1, m2 = re.search(r"Ran (\d+) calls in (\d+) launch sets", r1.stdout), re.search(r"Ran (\d+) calls in (\d+) launch sets", r2.stdout)
    assert m1 and m2 and int(m2.group(1)) == 40
    assert int(m2.group(2)) < 40, "--devices fell back to one launch set per call: %s" % m2.group(0)
    assert int(m1.group(2)) < 40


def _log_lines(stdout):
    """The driver's log without its timestamps and without the launch-set summary (which differs between window sizes)."""
    return [l.split("]", 1)[1] for l in stdout.splitlines() if "] Ran " not in l and "] Writing output" not in l and "]" in l]


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("traj_rgb_2x3", []), ("traj_dither_2x3", ["-d"])])
def test_cli_default_loop_is_speculative_and_matches_the_oracle_trajectory(tmp_path, name, extra):
    """With the reference's 64 candidates per call the driver runs snesimage_run_slots (several calls per launch set): the
    JSON of 216 calls equals the oracle's call-by-call trajectory (tests/golden/trajectories.json) byte for byte, and the
    log — every colour change, every error — equals the log of --window 1."""
    import hashlib
    case = next(c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "trajectories.json")))["cases"] if c["name"] == name)
    assert case["first_call"] == 0
    a, b = tmp_path / "a.json", tmp_path / "b.json"
    args = ["synth:%d" % case["seed"], "-c", str(case["count"]), "-s", str(case["size"]), "--calls", str(len(case["calls"])), "--seed", str(case["cand_seed"]), *extra]
    r1 = run(args[0], str(a), *args[1:])
    r2 = run(args[0], str(b), *args[1:], "--window", "1")
    assert r1.returncode == 0 and r2.returncode == 0, r1.stdout + r1.stderr + r2.stdout + r2.stderr
    assert hashlib.sha256(a.read_bytes()).hexdigest() == case["json_sha"]
    assert a.read_bytes() == b.read_bytes()
    assert "launch sets" in r1.stdout and _log_lines(r1.stdout) == _log_lines(r2.stdout)
    changes = [l for l in r1.stdout.splitlines() if "Setting color" in l]
    assert len(changes) == case["accepted"]
