"""The file readers of the C++ mirror (PNG, JPEG: what rt_image_load and both mirrors' Texture(path) run; OBJ, COLLADA, FBX)
on damaged files, under AddressSanitizer + UBSan on the host (tools/fuzz_decoders.cpp): truncations, byte flips, runs of 0xFF,
deleted spans.  A reader may refuse such a file or decode garbage; it must not touch memory it does not own, overflow a signed
integer, size a buffer by a header the data cannot back, or hang.  CPU only."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_RES = "/root/reference/res"


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("fuzz") / "fuzz_decoders")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "raytracer_amd", "host"), os.path.join(ROOT, "tools", "fuzz_decoders.cpp"), "-lz", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert r.returncode == 0, r.stderr
    return exe


def run(exe, iters, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, str(iters)] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    decoded, refused = (int(x) for x in r.stdout.split()[1::2])
    assert decoded + refused == iters * len(files)
    return decoded, refused


def test_damaged_images_and_obj(fuzzer, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    im = np.clip(np.cumsum(rng.normal(0, 6, (40, 57, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    Image.fromarray(im).save(str(tmp_path / "a.jpg"), quality=80, subsampling=2)
    Image.fromarray(im).save(str(tmp_path / "b.jpg"), quality=60, subsampling=1, restart_marker_blocks=2)
    Image.fromarray(im[..., 0]).save(str(tmp_path / "g.jpg"), quality=80)
    Image.fromarray(im).save(str(tmp_path / "a.png"))
    Image.fromarray(im[..., 0]).save(str(tmp_path / "g.png"))
    Image.fromarray(im).convert("P").save(str(tmp_path / "p.png"))
    Image.fromarray(np.dstack([im, im[..., :1]])).save(str(tmp_path / "rgba.png"))
    (tmp_path / "q.obj").write_text("mtllib q.mtl\nv -1 0 -1\nv 1 0 -1\nv 1 0.5 1\nv -1 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 1 0\n"
                                    "usemtl m\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf 1//1 2//1 3//1\nf 1 2 3\nf -1 -2 -3\n")
    (tmp_path / "q.mtl").write_text("newmtl m\nKd 0.5 0.4 0.3\nmap_Kd a.png\n")
    names = ["a.jpg", "b.jpg", "g.jpg", "a.png", "g.png", "p.png", "rgba.png", "q.obj"]
    decoded, refused = run(fuzzer, 400, [str(tmp_path / n) for n in names])
    assert decoded > 0 and refused > 0


def test_damaged_pngs_of_every_kind(fuzzer, tmp_path):
    """Seeds for the paths of the PNG reader that the files above do not reach: packed 1 / 2 / 4-bit samples, 16 bits, palettes
    with a tRNS, colour keys, gray + alpha, Adam7 (odd sizes, so that some passes are empty)."""
    from test_texture_decode import write_png
    rng = np.random.default_rng(1)
    h, w = 13, 21
    names = []

    def add(name, *a, **k):
        write_png(str(tmp_path / name), *a, **k)
        names.append(name)
    pal = rng.integers(0, 256, (16, 3))
    for il in (False, True):
        t = "i" if il else "n"
        for d in (1, 2, 4):
            add(f"g{d}{t}.png", rng.integers(0, 1 << d, (h, w)), d, interlace=il)
            add(f"p{d}{t}.png", rng.integers(0, min(16, 1 << d), (h, w)), d, palette=pal[:min(16, 1 << d)], trns=bytes([0, 128, 255][:min(3, 1 << d)]), interlace=il)
        add(f"g16{t}.png", rng.integers(0, 65536, (h, w)), 16, trns=(1234).to_bytes(2, "big"), interlace=il)
        add(f"ga8{t}.png", rng.integers(0, 256, (h, w, 2)), 8, interlace=il)
        add(f"rgb8k{t}.png", rng.integers(0, 4, (h, w, 3)) * 85, 8, trns=b"\x00\x55\x00\xaa\x00\xff", interlace=il)
        add(f"rgba16{t}.png", rng.integers(0, 65536, (h, w, 4)), 16, interlace=il)
    decoded, refused = run(fuzzer, 150, [str(tmp_path / n) for n in names])
    assert decoded > 0 and refused > 0


@pytest.mark.skipif(not os.path.isdir(REF_RES), reason="needs the reference checkout (its model files are the seeds)")
def test_damaged_reference_models(fuzzer, tmp_path):
    files = []
    for rel in ("lowpolypbrtrash/source/model.dae", "pbrsword/source/murasama.fbx"):
        dst = str(tmp_path / os.path.basename(rel))
        shutil.copy(os.path.join(REF_RES, rel), dst)     # the fuzzer writes its cases beside the seed file: never into the reference
        files.append(dst)
    decoded, refused = run(fuzzer, 60, files)
    assert decoded > 0 and refused > 0
