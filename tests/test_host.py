"""CPU tests of the host layer: GLM restatement, primitives, flattening, the C-ABI library's
exported symbols (no compute calls without a GPU), and the multi-rank sharding over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import capi, hostmath, multigpu, scenes
from raytracer_amd.scene import Cube, Material, Scene, Sphere, Texture, cube_mesh, expand_to_rgba8, sphere_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_matrix_and_inverse():
    m = hostmath.model_matrix((1.5, -2.0, 0.25), (0.0, 0.38268343, 0.0, 0.92387953), (2.0, 0.5, 1.25))
    inv = hostmath.inverse(m)
    np.testing.assert_allclose(hostmath.mat4_mul(m, inv), np.eye(4), atol=2e-6)
    # zero quaternion (glm::quat()) behaves as identity
    assert np.array_equal(hostmath.mat4_cast((0, 0, 0, 0)), np.eye(4, dtype=np.float32))
    t = hostmath.model_matrix((1, 2, 3), (0, 0, 0, 0), (1, 1, 1))
    assert np.array_equal(t.reshape(16)[12:15], np.array([1, 2, 3], np.float32))  # column-major, linal.hpp:5-9


def test_camera_rays():
    r = hostmath.camera_rays(8, 6, 45.0)
    assert r.shape == (6, 8, 4) and r.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(r[..., :3], axis=-1), 1.0, atol=1e-6)
    assert np.array_equal(r[3, 4], np.array([0, 0, -1, 0], np.float32))  # i = W/2, j = H/2 is the axis
    assert r[0, 0, 0] < 0 and r[0, 0, 1] < 0                             # row 0 = bottom of the picture
    vh = 2 * np.tan(np.radians(22.5))
    np.testing.assert_allclose(r[3, 5, 0] / -r[3, 5, 2], vh * 8 / 6 / 8, rtol=1e-6)


def test_primitives_match_reference_tables():
    c, s = cube_mesh(), sphere_mesh()
    assert c.triangleCount == 12 and c.vertices.size == 108 and c.normals is None
    assert s.triangleCount == 352 and s.vertices.size == 219 * 3
    assert np.array_equal(c.getMinBound(), [-.5, -.5, -.5]) and np.array_equal(c.getMaxBound(), [.5, .5, .5])
    ref = "/root/reference/src/Cube.hpp"
    if os.path.exists(ref):
        txt = open(ref).read()
        body = txt[txt.index("vertices = {"):txt.index("};")]
        vals = np.array([float(x) for x in re.findall(r"-?\d+\.\d+(?=f)", body)], np.float32)
        assert np.array_equal(vals, c.vertices)
        uv = txt[txt.index("uvCoords = {"):]
        uv = uv[:uv.index("};")]
        assert np.array_equal(np.array([float(x) for x in re.findall(r"\d+\.\d+(?=f)", uv)], np.float32), c.uvCoords)


def test_cube_ignores_its_transform_arguments():
    c = Cube(position=(5, 5, 5), scale=(3, 3, 3))
    assert np.array_equal(c.getModelMatrix(), np.eye(4, dtype=np.float32))  # Cube.hpp:8
    s = Sphere(position=(1.5, 0, 0))
    assert s.getModelMatrix()[3][0] == np.float32(1.5)


def test_scene_api_and_flatten():
    sc = Scene()
    m = Material("Default", (0.5, 0.5, 0.5, 1.0), 0.5, 0.0)
    a, b = Cube(), Sphere(position=(1.5, 0, 0))
    a.setMaterial(m); b.setMaterial(m)
    assert sc.addToScene("cube", a) and sc.addToScene("sphere", b) and not sc.addToScene("cube", b)
    f = sc.flatten()
    assert [o["meshIds"] for o in f.objects] == [[0], [1]] and len(f.materials) == 1
    assert f.triangle_count() == 364
    assert sc.removeFromSceen("cube") and not sc.removeFromSceen("cube")
    assert len(sc.flatten().objects) == 1


def test_texture_expansion_quirk_and_material_setters():
    gray = np.arange(6, dtype=np.uint8).reshape(2, 3)
    e = expand_to_rgba8(gray)
    assert e.shape == (2, 3, 4)
    assert list(e[0, 0]) == [0, 1, 2, 255] and list(e[1, 2]) == [5, 0, 0, 255]  # G,B = next pixels (Material.cu:187-193)
    rgb = np.zeros((2, 2, 3), np.uint8); rgb[..., 1] = 7
    assert list(expand_to_rgba8(rgb)[1, 1]) == [0, 7, 0, 255]
    m = Material("x")
    assert not m.setAlbedoTexture("/nonexistent/file.png") and m.albedoTexture is None
    assert m.setAlbedoTexture(Texture(rgb)) and m.albedoTexture is not None
    assert m.setAlbedoTexture("") and m.albedoTexture is None


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"rt_ctx"}
    lib = capi.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(capi.EXPORTS) == declared
    assert b"gfx950" in lib.rt_version()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RtError):
        capi.Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "rt_oracle.h" not in txt, f


def test_launch_constant_divisions(tmp_path):
    """The wavefront kernels divide by their launch constants (samples per pixel, tiles per row, rows per band) with a
    multiply and two shifts (csrc/rt_device.h, make_fast_div / fast_div): the constants the host code makes, with the device
    formula restated on host integers, against n / d for every divisor to 200 000, the powers of two and their neighbours up
    to 2^31 and two million random pairs (tests/cpp/fast_div_check.cpp, built with hipcc here: the header is HIP's)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "fast_div_check")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "fast_div_check.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and " bad 0" in p.stdout, p.stdout


def test_band_and_frame_partitions():
    for world in (1, 2, 3, 8):
        rows = np.zeros(1080, int)
        for r in range(world):
            for (x0, y0, x1, y1) in multigpu.band_rects(1920, 1080, r, world):
                assert (x0, x1) == (0, 1920)
                rows[y0:y1] += 1
        assert (rows == 1).all()
        fr = sorted(sum((multigpu.frame_indices(5, 21, r, world) for r in range(world)), []))
        assert fr == list(range(5, 26))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from oracle import oracle_py as orc
from raytracer_amd import multigpu, scenes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
scene, cam, st = scenes.config1(64, 48, 2, 3); cam.setPosition((0.6, 0.3, 4.0))
flat = scene.flatten()
full, _, _ = orc.render(flat, cam, st, threads=1)
# tiles: each rank renders its bands (the oracle stands in for the renderer on CPU)
fb = np.zeros((48, 64, 4), np.float32)
for r in multigpu.band_rects(64, 48, rank, world):
    part, _, _ = orc.render(flat, cam, st, rect=r, threads=1)
    fb[r[1]:r[3]] = part[r[1]:r[3]]
comp = multigpu.composite_tiles(dist, torch.from_numpy(fb), 64, 48, rank, world).numpy()
assert np.array_equal(comp, full), "tile composite differs from the single-rank frame"
# ragged case: 8-row bands over 44 rows = 6 bands, the last one short, packets padded
fb2 = np.zeros((48, 64, 4), np.float32)
for r in multigpu.band_rects(64, 44, rank, world, band_rows=8):
    fb2[r[1]:r[3]] = full[r[1]:r[3]]
comp2 = multigpu.composite_tiles(dist, torch.from_numpy(fb2[:44].copy()), 64, 44, rank, world, band_rows=8).numpy()
assert np.array_equal(comp2, full[:44]), "ragged tile composite differs"
# the compositor bench.py uses: the frame lives in a padded tensor, one strided copy each way, no per-band loop
tc = multigpu.TileCompositor(torch, dist, 64, 44, rank, world, band_rows=8)
assert tc.padded[0].shape[0] == 48 and tc.frame.shape[0] == 44 and tc.per_rank == 3
for b in range(rank, tc.nb, world):   # what rt_render_bands(firstBand = rank, bandStride = world) fills
    tc.frame[b * 8:min(44, (b + 1) * 8)] = torch.from_numpy(full[b * 8:min(44, (b + 1) * 8)])
assert np.array_equal(tc.composite().numpy(), full[:44]), "TileCompositor differs"
# ... and its pipelined form: two buffers, frame i composited while frame i + 1 is rendered (on CPU: in call order)
tc2 = multigpu.TileCompositor(torch, dist, 64, 44, rank, world, band_rows=8, buffers=2)
frames_ref = []
for i in range(3):
    st.frameCount = 10 + i
    ref_i, _, _ = orc.render(flat, cam, st, threads=1)
    frames_ref.append(ref_i)
    b = tc2.acquire(i)
    tc2.frames[b].zero_()
    for band in range(rank, tc2.nb, world):
        tc2.frames[b][band * 8:min(44, (band + 1) * 8)] = torch.from_numpy(ref_i[band * 8:min(44, (band + 1) * 8)])
    tc2.submit(b)
    if i >= 1:   # the previous frame's buffer is complete by now
        assert np.array_equal(tc2.frames[(i - 1) % 2].numpy(), frames_ref[i - 1][:44])
tc2.drain()
assert np.array_equal(tc2.frames[0].numpy(), frames_ref[2][:44])
st.frameCount = 0
# frames: progressive split + reduce == serial sum within 1e-6
frames = list(range(0, 6))
local = torch.zeros((48, 64, 4))
for f in multigpu.frame_indices(0, 6, rank, world):
    st.frameCount = f
    img, _, _ = orc.render(flat, cam, st, threads=1)
    local += torch.from_numpy(img)
multigpu.reduce_frames(dist, local, world)
if rank == 0:
    serial = np.zeros((48, 64, 4), np.float64)
    for f in frames:
        st.frameCount = f
        serial += orc.render(flat, cam, st, threads=1)[0]
    assert np.abs(local.numpy() / 6 - serial / 6).max() < 1e-6
# the same split through the double-buffered accumulator bench.py uses (reduce of frame i behind the render of i+1)
bufs = [torch.zeros((48, 64, 4)), torch.zeros((48, 64, 4))]
acc = torch.zeros((48, 64, 4))
fa = multigpu.FrameAccumulator(dist, world, rank, bufs, acc)
for step in range(3):
    b = fa.acquire(step)
    st.frameCount = step * world + rank
    bufs[b].copy_(torch.from_numpy(orc.render(flat, cam, st, threads=1)[0]))
    fa.submit(b)
fa.drain()
if rank == 0:
    serial = np.zeros((48, 64, 4), np.float64)
    for f in range(3 * world):
        st.frameCount = f
        serial += orc.render(flat, cam, st, threads=1)[0]
    assert np.abs(acc.numpy() / (3 * world) - serial / (3 * world)).max() < 1e-6
else:
    assert float(acc.abs().max()) == 0.0
dist.barrier(); dist.destroy_process_group()
sys.stdout.write("rank %d ok\n" % rank); sys.stdout.flush()  # one write: the two ranks finish together
'''


def test_two_rank_sharding_over_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:  # a free rendezvous port (a fixed one can linger in TIME_WAIT after an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(w), ROOT],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "rank 0 ok" in p.stdout and "rank 1 ok" in p.stdout
