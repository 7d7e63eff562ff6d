"""Texture(path) texel bytes (Material.cu:162-206) and the path-keyed sharing of textures and models
(Material.cu:165-178, 234-252; CustomModel.cpp:10-45) in both host mirrors.  CPU only.

Pinning: oracle/_ref/libstbi_ref.so is the reference's own vendored decoder (src/stb_image.h v2.28) compiled from
where it lies by `make -C oracle ref`.  When the reference checkout and that build are present (this container),
every PNG and JPEG the reference ships is decoded by stb_image, by the library's rt_image_load (what the Python
mirror's Texture(path) calls) and by the C++ mirror, and the bytes must be identical: 16-bit PNGs included (stb reduces
with >> 8, stb_image.h:1198), and JPEG, where the decoder shows in the bytes (PIL differs from stb by up to 3 levels on
these files), through a decoder that makes stb_image's fixed-point choices (raytracer_amd/host/rt_jpeg.hpp).  JPEGs written
here with other sampling factors, qualities, restart intervals and optimised tables are compared the same way."""
import ctypes as C
import glob
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from raytracer_amd import scene as scn
from raytracer_amd.scene import CustomModel, Material, Scene, Texture, expand_to_rgba8

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "host", "rt_headless")
STB = os.path.join(ROOT, "oracle", "_ref", "libstbi_ref.so")
REF_RES = "/root/reference/res"


def cpp_decode(path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE, "decode", path], capture_output=True, text=True, check=True).stdout.split()
    return None if out == ["refused"] else tuple(int(v) for v in out)


def write_png(path, arr, depth, palette=None, trns=None, interlace=False):
    """Minimal PNG writer for the tests: arr is HxW (gray, or palette indices when `palette` is given) or HxWx3 (RGB) or
    HxWx2 / HxWx4 (with alpha); depth 1, 2, 4 (gray / palette), 8 or 16; trns = the body of a tRNS chunk; Adam7 on request."""
    h, w = arr.shape[:2]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctype = 3 if palette is not None else {1: 0, 2: 4, 3: 2, 4: 6}[ch]

    def rows(a):   # filter type 0, samples packed
        hh, ww = a.shape[:2]
        out = b""
        for y in range(hh):
            r = a[y].reshape(-1)
            if depth == 16:
                body = r.astype(">u2").tobytes()
            elif depth == 8:
                body = r.astype(np.uint8).tobytes()
            else:
                bits = np.unpackbits(r.astype(np.uint8)[:, None], axis=1)[:, 8 - depth:].reshape(-1)
                body = np.packbits(bits).tobytes()
            out += b"\x00" + body
        return out

    if interlace:
        raw = b""
        for xo, yo, xs, ys in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = arr[yo::ys, xo::xs]
            if sub.shape[0] and sub.shape[1]:
                raw += rows(sub)
    else:
        raw = rows(arr)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)
    body = chunk(b"IHDR", ihdr)
    if palette is not None:
        body += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        body += chunk(b"tRNS", bytes(trns))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + body + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.skipif(not os.path.exists(STB), reason="needs `make -C oracle ref` (the reference checkout)")
def test_png_kinds_decode_like_stb_image(tmp_path):
    """Low bit depths, palettes, colour keys (tRNS) and Adam7 interlacing, all of which stb_image reads (Material.cu:179) and
    the reference could be handed: same bytes and the same channel count from rt_image_load, the Python mirror and the C++
    mirror as from the reference's own stb_image."""
    rng = np.random.default_rng(11)
    k = 0

    def case(arr, depth, **kw):
        nonlocal k
        f = str(tmp_path / f"k{k}.png"); k += 1
        write_png(f, arr, depth, **kw)
        same_as_stb(f)
        return stb_load(f)

    for w, h in ((37, 23), (8, 8), (1, 1), (3, 9)):
        for depth in (1, 2, 4):
            g = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)
            for il in (False, True):
                assert case(g, depth, interlace=il).shape[2] == 1                                       # gray
                assert case(g, depth, interlace=il, trns=struct.pack(">H", int(g[0, 0]))).shape[2] == 2   # gray + colour key
                pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8)
                assert case(g, depth, palette=pal, interlace=il).shape[2] == 3                          # palette
                assert case(g, depth, palette=pal, trns=rng.integers(0, 256, max(1, (1 << depth) - 1), dtype=np.uint8).tobytes(),
                            interlace=il).shape[2] == 4                                                 # palette + alphas (a short list)
        g8 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rgb8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        g16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        rgb16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
        pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
        for il in (False, True):
            assert case(g8, 8, interlace=il, trns=struct.pack(">H", int(g8[0, 0]))).shape[2] == 2
            assert case(rgb8, 8, interlace=il, trns=struct.pack(">HHH", *[int(v) for v in rgb8[h // 2, w // 2]])).shape[2] == 4
            assert case(g16, 16, interlace=il, trns=struct.pack(">H", int(g16[0, 0]))).shape[2] == 2
            assert case(rgb16, 16, interlace=il, trns=struct.pack(">HHH", *[int(v) for v in rgb16[0, 0]])).shape[2] == 4
            assert case(g8, 8, palette=pal, interlace=il, trns=rng.integers(0, 256, 200, dtype=np.uint8).tobytes()).shape[2] == 4
            case(rng.integers(0, 256, (h, w, 2), dtype=np.uint8), 8, interlace=il)
            case(rng.integers(0, 65536, (h, w, 4), dtype=np.uint16), 16, interlace=il)
    # a colour key that differs from a pixel only in the low byte of a 16-bit sample is not that pixel's key
    g16 = np.full((4, 4), 0x1234, np.uint16); g16[1, 1] = 0x1235
    out = case(g16, 16, trns=struct.pack(">H", 0x1235))
    assert out[1, 1, 1] == 0 and out[0, 0, 1] == 255 and out[1, 1, 0] == 0x12
    # an image with an alpha channel and a tRNS chunk: an error in stb_image, a refusal here
    f = str(tmp_path / "bad.png")
    write_png(f, rng.integers(0, 256, (5, 5, 4), dtype=np.uint8), 8, trns=b"\x00\x00")
    from raytracer_amd import capi
    assert stb_load(f) is None and capi.image_load(f) is None and cpp_decode(f) is None


def test_sixteen_bit_png_is_reduced_with_shift_8(tmp_path):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    g[0, :4] = (0x00FF, 0x0100, 0x80FF, 0xFFFF)            # >> 8 gives 0, 1, 128, 255; rounding would give 1, 1, 129, 255
    rgb = rng.integers(0, 65536, (19, 23, 3), dtype=np.uint16)
    write_png(str(tmp_path / "g16.png"), g, 16)
    write_png(str(tmp_path / "rgb16.png"), rgb, 16)
    for name, arr in (("g16.png", g), ("rgb16.png", rgb)):
        want = (arr >> 8).astype(np.uint8)
        t = Texture(str(tmp_path / name))
        assert t.getLoadSuccess() and np.array_equal(t.rgba8, expand_to_rgba8(want))
        w, h, n, crc = cpp_decode(str(tmp_path / name))
        assert (w, h, n) == (arr.shape[1], arr.shape[0], 1 if arr.ndim == 2 else 3) and crc == zlib.crc32(want.tobytes())
    assert tuple(Texture(str(tmp_path / "g16.png")).rgba8[0, :4, 0]) == (0, 1, 128, 255)


def stb_load(path):
    L = C.CDLL(STB)
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(n), 0)     # Material.cu:179: stbi_load(path, ..., 0)
    if not p:
        return None
    ref = np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).copy()
    L.stbi_image_free(p)
    return ref


def same_as_stb(f):
    from raytracer_amd import capi
    ref = stb_load(f)
    assert ref is not None, f
    got = capi.image_load(f)
    assert got is not None and np.array_equal(got, ref), f                                          # rt_image_load
    t = Texture(f)
    assert t.getLoadSuccess() and np.array_equal(t.rgba8, expand_to_rgba8(ref)), f                  # Python mirror, after the 4-channel expansion
    assert cpp_decode(f) == (ref.shape[1], ref.shape[0], ref.shape[2], zlib.crc32(ref.tobytes())), f  # C++ mirror


@pytest.mark.skipif(not (os.path.isdir(REF_RES) and os.path.exists(STB)), reason="needs the reference checkout and `make -C oracle ref`")
def test_reference_textures_decode_like_the_references_own_stb_image():
    pngs = sorted(glob.glob(REF_RES + "/**/*.png", recursive=True))
    jpgs = sorted(glob.glob(REF_RES + "/**/*.jp*g", recursive=True))
    assert len(pngs) >= 12 and len(jpgs) >= 14
    sixteen = 0
    for f in pngs + jpgs:
        same_as_stb(f)
        sixteen += f.endswith(".png") and struct.unpack(">B", open(f, "rb").read(25)[24:25])[0] == 16
    assert sixteen >= 2   # Base_Color (RGB16) and Displacement (gray16) are among them
    # the fixture the GPU box renders with holds the same bytes
    fx = os.path.join(ROOT, "tests", "golden", "ref_assets.npz")
    if os.path.exists(fx):
        z = np.load(fx)
        for key, rel in (("sword_roughness", "pbrsword/texture/Material.002_Roughness.png"), ("metal_metalness", "metalmaterial/Metal046B_1K-PNG_Metalness.png"),
                         ("airplane_albedo", "basiclowpoly/Airplane Texture.png"), ("sword_base_color", "pbrsword/texture/Material.002_Base_Color.png")):
            assert np.array_equal(expand_to_rgba8(z[key]), Texture(REF_RES + "/" + rel).rgba8)
            assert np.array_equal(z[key].reshape(stb_load(REF_RES + "/" + rel).shape), stb_load(REF_RES + "/" + rel))   # C2's and C3's colour maps = the reference's stbi_load bytes


@pytest.mark.skipif(not os.path.exists(STB), reason="needs `make -C oracle ref` (the reference checkout)")
def test_written_jpegs_decode_like_stb_image(tmp_path):
    """Sampling 4:4:4 / 4:2:2 / 4:2:0, gray, two qualities, restart intervals, optimised Huffman tables, odd extents."""
    from PIL import Image
    rng = np.random.default_rng(1)
    noise = (rng.random((61, 83, 3)) * 255).astype(np.uint8)
    smooth = np.clip(np.cumsum(rng.normal(0, 6, (120, 177, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    k = 0
    for im in (noise, smooth):
        for ss in (0, 1, 2):
            for q in (35, 90):
                f = str(tmp_path / f"t{k}.jpg"); k += 1
                Image.fromarray(im).save(f, quality=q, subsampling=ss)
                same_as_stb(f)
    Image.fromarray(smooth[..., 0]).save(str(tmp_path / "g.jpg"), quality=80)
    Image.fromarray(smooth).save(str(tmp_path / "r.jpg"), quality=80, subsampling=2, restart_marker_blocks=3)
    Image.fromarray(smooth).save(str(tmp_path / "o.jpg"), quality=80, optimize=True, subsampling=1)
    Image.fromarray(smooth[:1, :1]).save(str(tmp_path / "one.jpg"), quality=80)
    for name in ("g.jpg", "r.jpg", "o.jpg", "one.jpg"):
        same_as_stb(str(tmp_path / name))


def test_jpeg_loads_in_both_mirrors_and_unsupported_files_are_refused(tmp_path):
    from PIL import Image
    img = (np.random.default_rng(2).random((32, 48, 3)) * 255).astype(np.uint8)
    Image.fromarray(img).save(str(tmp_path / "a.jpg"), quality=90)
    assert cpp_decode(str(tmp_path / "a.jpg"))[:3] == (48, 32, 3)
    t = Texture(str(tmp_path / "a.jpg"))
    assert t.getLoadSuccess() and t.rgba8.shape == (32, 48, 4) and np.abs(t.rgba8[..., :3].astype(int) - np.asarray(Image.open(str(tmp_path / "a.jpg"))).astype(int)).max() <= 4   # libjpeg is close, not equal
    Image.fromarray(img).save(str(tmp_path / "p.jpg"), quality=90, progressive=True)
    (tmp_path / "junk.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00" * 40)
    for bad in ("junk.png", "missing.png"):   # broken PNG, no file: refused, never a crash
        assert cpp_decode(str(tmp_path / bad)) is None and not Texture(str(tmp_path / bad)).getLoadSuccess()
    # a progressive JPEG (stb_image reads those; rt_image_load does not): the C++ mirror refuses it, the Python mirror decodes
    # it with Pillow -- a stated parity deviation: the bytes are close to stb_image's, not equal
    from raytracer_amd import capi
    assert capi.image_load(str(tmp_path / "p.jpg")) is None and cpp_decode(str(tmp_path / "p.jpg")) is None
    tp = Texture(str(tmp_path / "p.jpg"))
    assert tp.getLoadSuccess() and tp.rgba8.shape == (32, 48, 4)
    assert np.abs(tp.rgba8[..., :3].astype(int) - Texture(str(tmp_path / "a.jpg")).rgba8[..., :3].astype(int)).max() <= 6
    m = Material("m")
    assert m.setAlbedoTexture(str(tmp_path / "a.jpg")) and not m.setAlbedoTexture(str(tmp_path / "missing.png"))
    assert m.albedoTexture is not None   # a failed load keeps the old texture (setTextureRef, Material.cu:113-130)


def test_textures_and_models_are_shared_by_path(tmp_path):
    img = (np.random.default_rng(3).random((8, 8, 3)) * 255).astype(np.uint8)
    write_png(str(tmp_path / "t.png"), img, 8)
    before = scn.TEXTURE_DECODES
    a, b = Material("a"), Material("b")
    assert a.setAlbedoTexture(str(tmp_path / "t.png")) and b.setRoughnessTexture(str(tmp_path / "t.png"))
    assert scn.TEXTURE_DECODES == before + 1 and a.albedoTexture is not b.roughnessTexture
    assert a.albedoTexture.rgba8 is b.roughnessTexture.rgba8
    (tmp_path / "q.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0.5 1\nv -1 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    parses = CustomModel.PARSES
    m1 = CustomModel(str(tmp_path / "q.obj"))
    m2 = CustomModel(str(tmp_path / "q.obj"), position=(2.0, 0.0, 0.0))
    assert CustomModel.PARSES == parses + 1 and m1.getLoadSuccess() and m2.getLoadSuccess()
    assert m1.meshes[0][0] is m2.meshes[0][0]
    m1.setMaterial(a); m2.setMaterial(b)
    sc = Scene(); sc.addToScene("m1", m1); sc.addToScene("m2", m2)
    flat = sc.flatten()
    assert len(flat.meshes) == 1 and len(flat.textures) == 1 and len(flat.objects) == 2   # one upload each
    assert flat.objects[1]["model"][12] == np.float32(2.0)
    del m1, m2, sc, flat
    import gc
    gc.collect()
    CustomModel(str(tmp_path / "q.obj"))
    assert CustomModel.PARSES == parses + 2   # the last owner gone: parsed again (customModels.erase, CustomModel.cpp:26-45)
