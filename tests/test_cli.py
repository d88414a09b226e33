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
