"""The C++ host mirror (raytracer_amd/host/rt_host.hpp): its GLM restatements must give the same
bits as the Python mirror (CPU), and a scene built through its class API must render to the
committed golden frame (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import hostmath

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "host", "rt_headless")


def _bits(line):
    return np.array([int(x, 16) for x in line.split()[1:]], np.uint32)


def test_cpp_host_math_matches_python_bit_for_bit():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE, "math"], capture_output=True, text=True, check=True).stdout.splitlines()
    m = hostmath.model_matrix((1.5, -2.0, 0.25), (0.0, 0.38268343, 0.0, 0.92387953), (2.0, 0.5, 1.25))
    assert np.array_equal(_bits(out[0]), m.reshape(16).view(np.uint32))
    assert np.array_equal(_bits(out[1]), hostmath.inverse(m).reshape(16).view(np.uint32))
    assert np.array_equal(_bits(out[2]), hostmath.camera_rays(7, 5, 45.0).reshape(-1).view(np.uint32))


@pytest.mark.gpu
def test_cpp_host_renders_the_golden_frame(tmp_path):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_golden.npz"))
    png, raw = str(tmp_path / "o.png"), str(tmp_path / "o.f32")
    p = subprocess.run([EXE, "render", png, raw, "64", "64", "1", "4"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    fb = np.fromfile(raw, np.float32).reshape(64, 64, 4)
    assert np.array_equal(fb, gold["c1_s1_fb"])
    from PIL import Image
    im = np.asarray(Image.open(png))
    assert im.shape == (64, 64, 4) and tuple(im[0, 0, :3]) == (104, 104, 139)  # sky, top-left after the flip
