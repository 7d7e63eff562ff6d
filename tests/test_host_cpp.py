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


@pytest.mark.gpu
def test_cpp_frame_counters_and_sharing(tmp_path):
    """The C++ GPURaytracer advances frameCount / progressiveFrameCount as GPURaytracer.cu:69-73 does (seven frames,
    progressive on-on-on-off-off-on-on, each compared with the oracle's frame for the expected counters), and
    Textures / CustomModels of one path share one upload (Material.cu:165-178, CustomModel.cpp:10-24)."""
    from oracle import oracle_py as orc
    from raytracer_amd import scenes
    import test_assets as ta
    W, H = 72, 48
    raw, png, model = str(tmp_path / "f.f32"), str(tmp_path / "t.png"), str(tmp_path / "m.fbx")
    ta.write_fbx(model)
    p = subprocess.run([EXE, "frames", raw, str(W), str(H), png, model], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.fromfile(raw, np.float32).reshape(7, H, W, 4)
    scene, cam, st = scenes.config1(W, H, 2, 3)
    cam.setPosition((0.6, 0.3, 4.0))
    flat = scene.flatten()
    acc = None
    for k, (fc, pfc) in enumerate([(0, 0), (1, 1), (2, 2), (3, 3), (4, 0), (5, 0), (6, 1)]):
        st.frameCount, st.progressiveFrameCount = fc, pfc
        acc, _, _ = orc.render(flat, cam, st, prev=acc)
        assert np.array_equal(got[k], acc), (k, fc, pfc)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3])
def test_cpp_one_thread_drives_n_contexts(tmp_path, n):
    """MultiGPURaytracer (rt_host.hpp): one host thread, n contexts, each rendering its interleaved bands of one frame
    with one rt_render_bands call; the host composite equals the oracle's frame (8-row bands over 50 rows: ragged)."""
    from oracle import oracle_py as orc
    from raytracer_amd import scenes
    W, H, spp = 70, 50, 3
    raw = str(tmp_path / "m.f32")
    p = subprocess.run([EXE, "multi", raw, str(W), str(H), str(spp), str(n)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    scene, cam, st = scenes.config1(W, H, spp, 4)
    cam.setPosition((0.6, 0.3, 4.0))
    st.frameCount = 3
    ref, _, _ = orc.render(scene.flatten(), cam, st)
    assert np.array_equal(np.fromfile(raw, np.float32).reshape(H, W, 4), ref)


def _cpp_asset(path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    p = subprocess.run([EXE, "asset", path], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout.splitlines()


def _compare_with_python(path):
    """rt_assets.hpp against raytracer_amd/assets.py: sizes, material slots and a CRC of every array."""
    import zlib
    from raytracer_amd import assets
    a = assets.load_asset(path)
    out = _cpp_asset(path)
    head = out[0].split()
    assert (int(head[1]), int(head[3]), int(head[5])) == (len(a.meshes), len(a.materials), a.triangle_count)
    assert head[7] == a.texture_directory()
    for line, m in zip([l for l in out if l.startswith("mesh ")], a.meshes):
        f = line.split()
        crc = lambda x: zlib.crc32(np.ascontiguousarray(x).tobytes()) & 0xFFFFFFFF
        nrm = m.normals if m.normals is not None else np.zeros(0, np.float32)
        assert [int(v) for v in f[1:]] == [m.positions.shape[0], m.indices.shape[0], m.material, crc(m.positions), crc(m.indices),
                                         crc(m.uvs), crc(nrm)], (path, line)
    for line, m in zip([l for l in out if l.startswith("material ")], a.materials):
        f = line.split(" ", 2)
        assert f[1] == m.name.split(" ")[0] or line[len("material "):].startswith(m.name)
        for role, tex in m.textures.items():
            assert f"{role}={tex}" in line


def test_cpp_asset_flatteners_match_python(tmp_path):
    import test_assets as ta
    ta.write_fbx(str(tmp_path / "m.fbx"))
    _compare_with_python(str(tmp_path / "m.fbx"))
    # OBJ with an MTL (texture names with blanks and options) and relative indices
    (tmp_path / "m.mtl").write_text("newmtl skin\nKd 0.2 0.4 0.6\nmap_Kd tex/a b.png\nmap_Bump -bm 1.0 tex/n.png\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0.5\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 0.25\nvn 0 0 1\n"
                                    "usemtl skin\nf 1/1/1 2/2/1 3/3/1 4/4/1\nusemtl bare\nf -4 -3 -2\n")
    _compare_with_python(str(tmp_path / "m.obj"))
    assert _cpp_asset(str(tmp_path / "m.obj"))[-2].startswith("material skin albedo=tex/a b.png normal=tex/n.png")
    bad = tmp_path / "bad.dae"
    bad.write_text("<COLLADA><library_geometries>")
    p = subprocess.run([EXE, "asset", str(tmp_path / "missing.fbx")], capture_output=True, text=True)
    assert p.returncode == 3


@pytest.mark.skipif(not os.path.exists("/root/reference/res"), reason="reference checkout absent (it is on the GPU box)")
def test_cpp_asset_flatteners_on_reference_files():
    _compare_with_python("/root/reference/res/lowpolypbrtrash/source/model.dae")
    _compare_with_python("/root/reference/res/pbrsword/source/murasama.fbx")


@pytest.mark.gpu
def test_cpp_legacy_cpu_raytracer(tmp_path):
    """rth::CPURaytracer, the reference's legacy renderer behind its "Use GPU" toggle (processor/CPU/CPURaytracer.cpp),
    against the independent numpy restatement oracle/legacy_cpu.py on a textured, rotated, emissive scene: same frame bit
    for bit (both call the C library's logf / cosf / expf).  Needs a context only because the mirror's Mesh / Material
    constructors upload, as the reference's do when a GPU is present; the frame itself is rendered on the host."""
    from oracle import legacy_cpu
    from raytracer_amd import hostmath
    from raytracer_amd.scene import cube_mesh, sphere_mesh
    W, H = 40, 28
    raw = str(tmp_path / "l.f32")
    p = subprocess.run([EXE, "legacy", raw, str(W), str(H)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.fromfile(raw, np.float32).reshape(H, W, 4)
    i = np.arange(256)
    tex = np.zeros((16, 16, 4), np.uint8)
    tex.reshape(-1, 4)[:, 0] = (i * 7 + 30) & 255; tex.reshape(-1, 4)[:, 1] = 255 - i; tex.reshape(-1, 4)[:, 2] = (i * 13) & 255; tex[..., 3] = 255
    mats = {"a": dict(albedo=(0.8, 0.3, 0.2, 1.0), emissionColor=(0, 0, 0), emissionStrength=0.0, tex={"albedo": tex}),
            "b": dict(albedo=(0.2, 0.6, 0.9, 1.0), emissionColor=(0.05, 0.1, 0.02), emissionStrength=1.0, tex={}),
            "e": dict(albedo=(0.9, 0.9, 0.9, 1.0), emissionColor=(1.0, 0.8, 0.5), emissionStrength=2.0, tex={})}

    def obj(mesh, pos, rot, scl, mat):
        m = hostmath.model_matrix(pos, rot, scl)   # [col][row], as glm stores it
        return dict(model=m, vertices=mesh.vertices.reshape(-1, 3), indices=mesh.indices.reshape(-1, 3),
                    uvs=None if mesh.uvCoords is None else mesh.uvCoords.reshape(-1, 2), min=mesh.getMinBound(), max=mesh.getMaxBound(), material=mats[mat])
    objects = [obj(cube_mesh(), (-0.7, 0.0, 0.0), (0.0, 0.1, 0.0, 0.995), (1.3, 1.6, 1.1), "a"),
               obj(sphere_mesh(), (0.9, 0.1, 0.3), (0.1, 0.2, 0.0, 0.97), (1.4, 1.0, 1.2), "b"),
               obj(cube_mesh(), (0.2, 1.6, 0.5), (0, 0, 0, 0), (3.0, 0.2, 3.0), "e")]
    ref = legacy_cpu.render(objects, (0.1, 0.4, 5.0), (-0.05, 0.02, 0.0, 0.998), hostmath.camera_rays(W, H, 45.0), 1.2,
                            bounce_count=3, max_distance=100.0, ao_intensity=0.05, seed=0)
    assert (ref[..., :3] > 0).any(-1).mean() > 0.15 and len(np.unique(ref[..., 0])) > 50     # not a trivial frame (no sky in this renderer)
    bad = np.nonzero((got != ref).any(-1))
    assert bad[0].size == 0, (bad[0].size, got[bad][:4], ref[bad][:4])
