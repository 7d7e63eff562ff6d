"""C-ABI behaviour beyond a single frame (include/rt_api.h): hot material updates on a committed
scene (Material.hpp:155-159), handle lifetime against the committed tables, the frame-counter rule of
GPURaytracer::raytrace (GPURaytracer.cu:69-73), and the redo lists of a frame cut into many chunks.
Every frame is compared with the CPU oracle on the same inputs, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as orc
from raytracer_amd import capi, scenes
from raytracer_amd.scene import Camera, Cube, Material, Mesh, MeshObject, RenderSettings, Scene, Sphere, Texture

pytestmark = pytest.mark.gpu


mat_desc = capi.Context.material_desc


def small_scene():
    sc = Scene()
    a = Material("a", (0.8, 0.3, 0.2, 1.0), 0.4, 0.1)
    a.setAlbedoTexture(Texture(scenes.procedural_texture(64, 3, "albedo")))
    b = Material("b", (0.2, 0.5, 0.8, 1.0), 0.7, 0.6)
    s = Sphere(position=(0.9, 0.1, 0.0), scale=(1.4, 1.4, 1.4)); s.setMaterial(a); sc.addToScene("s", s)
    c = Cube(); c.setPosition((-0.8, -0.2, 0.3)); c.setRotation((0.1, 0.3, 0.0, 0.95)); c.setMaterial(b); sc.addToScene("c", c)
    slab = MeshObject([(scenes.merge_meshes([scenes.box_mesh((-4, -1.2, -4), (4, -0.9, 4))]), 0)], [b])
    sc.addToScene("slab", slab)
    cam = Camera((0.2, 0.8, 5.0), (-0.08, 0.0, 0.0, 0.997), 45.0, 1.0)
    cam.setWidth(112); cam.setHeight(72); cam.calcRays()
    return sc, cam


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["bvh-wavefront", "brute", "bvh-persistent"])
def test_material_update_on_committed_scene(ctx, mode):
    """rt_material_update patches the committed tables in place: constants, an unbound texture, and a
    texture created after the commit (the reference's GUI calls set*Texture + sendToGPU on a live material)."""
    sc, cam = small_scene()
    st = RenderSettings(bounceCount=4, sampleCount=2, frameCount=5)
    flat = sc.flatten()
    ctx.set_option(capi.OPT_TRAVERSAL, mode)
    ids = ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    ctx.render(cam, st, fb)
    ref, rid, nq = orc.render(flat, cam, st)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)

    # (1) constants of material b, emission switched on
    flat.materials[1].update(albedo=np.array([0.9, 0.9, 0.2, 1.0], np.float32), roughness=np.float32(0.15),
                             metal=np.float32(1.0), emissionColor=np.array([0.2, 0.9, 0.4], np.float32),
                             emissionStrength=np.float32(1.5))
    d = mat_desc(flat.materials[1], ids["tex"])
    ctx.check(ctx.L.rt_material_update(ctx.h, ids["mat"][1], C.byref(d)))
    ctx.render(cam, st, fb)
    ref, rid, nq = orc.render(flat, cam, st)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref) and ctx.stats().rays == nq

    # (2) a texture that did not exist at commit time bound as a's roughness map; a's albedo map unbound
    rough = scenes.procedural_texture(32, 17, "scalar")
    out = C.c_int32()
    ctx.check(ctx.L.rt_texture_create(ctx.h, rough.ctypes.data, 32, 32, C.byref(out)))
    flat.textures.append(rough)
    handles = ids["tex"] + [out.value]
    flat.materials[0].update(texRoughness=len(flat.textures) - 1, texAlbedo=-1)
    d = mat_desc(flat.materials[0], handles)
    ctx.check(ctx.L.rt_material_update(ctx.h, ids["mat"][0], C.byref(d)))
    ctx.render(cam, st, fb)
    ref, rid, nq = orc.render(flat, cam, st)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)
    assert np.array_equal(ctx.read_ids(fb, cam.width, cam.height), rid)

    # (3) invalid handles are refused and leave the committed scene as it was
    bad = mat_desc(flat.materials[0], handles)
    bad.texMetal = 9999
    assert ctx.L.rt_material_update(ctx.h, ids["mat"][0], C.byref(bad)) == -1
    assert ctx.L.rt_material_update(ctx.h, 4242, C.byref(d)) == -1
    ctx.render(cam, st, fb)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)
    ctx.check(ctx.L.rt_texture_destroy(ctx.h, out.value))   # bound by a committed material: see the next test
    with pytest.raises(capi.RtError, match="before rt_scene_commit"):
        ctx.render(cam, st, fb)
    ctx.framebuffer_destroy(fb)
    ctx.upload_scene(Scene().flatten())


def test_texture_destroy_against_the_committed_tables(ctx):
    """A texture the committed scene samples cannot be freed under it: the context asks for a new commit.
    A texture no committed material binds can go, and the scene stays renderable."""
    sc, cam = small_scene()
    st = RenderSettings(bounceCount=3, sampleCount=1)
    flat = sc.flatten()
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ids = ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    spare = scenes.procedural_texture(32, 1, "albedo")
    out = C.c_int32()
    ctx.check(ctx.L.rt_texture_create(ctx.h, spare.ctypes.data, 32, 32, C.byref(out)))
    ctx.check(ctx.L.rt_texture_destroy(ctx.h, out.value))
    ctx.render(cam, st, fb)
    ref, _, _ = orc.render(flat, cam, st)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)
    ctx.check(ctx.L.rt_texture_destroy(ctx.h, ids["tex"][0]))
    with pytest.raises(capi.RtError, match="before rt_scene_commit"):
        ctx.render(cam, st, fb)
    ids["tex"].clear()                     # already destroyed
    ctx.framebuffer_destroy(fb)
    ctx.upload_scene(Scene().flatten())    # releases the rest and commits an empty scene


def test_frame_counters_advance_like_the_reference(ctx):
    """GPURaytracer::raytrace (GPURaytracer.cu:69-73): frameCount++ every frame; progressiveFrameCount++
    while progressive rendering is on, reset to 0 by the first frame rendered with it off."""
    sc, cam = small_scene()
    flat = sc.flatten()
    rt = capi.GPURaytracer(ctx)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    rt.setBounceCount(3); rt.setSampleCount(2)
    fb = ctx.framebuffer(cam.width, cam.height)
    st = RenderSettings(bounceCount=3, sampleCount=2)
    acc = None
    expect = []
    # progressive on for three frames, off for two, on again for two
    plan = [True, True, True, False, False, True, True]
    fc, pfc = 0, 0
    for prog in plan:
        rt.setProgressiveRendering(prog)
        assert (rt.st.frameCount, rt.st.progressiveFrameCount) == (fc, pfc)
        st.frameCount, st.progressiveFrameCount = fc, pfc
        acc, _, _ = orc.render(flat, cam, st, prev=acc)
        rt.raytrace(sc, cam, fb)
        assert np.array_equal(ctx.read(fb, cam.width, cam.height), acc), (fc, pfc)
        expect.append((fc, pfc))
        fc += 1
        pfc = pfc + 1 if prog else 0
    assert expect == [(0, 0), (1, 1), (2, 2), (3, 3), (4, 0), (5, 0), (6, 1)]
    ctx.framebuffer_destroy(fb)


def test_moving_an_object_of_the_same_scene_is_rendered(ctx):
    """The reference rebuilds its instance table from the live objects every frame (Scene.cpp:3-29): a
    transform edited between two raytrace() calls on the same Scene object must show in the next frame."""
    sc, cam = small_scene()
    rt = capi.GPURaytracer(ctx)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    rt.setBounceCount(3); rt.setSampleCount(1)
    fb = ctx.framebuffer(cam.width, cam.height)
    st = RenderSettings(bounceCount=3, sampleCount=1)
    for k in range(3):
        sc.getObjects()["s"].setPosition((0.9 - 0.4 * k, 0.1 + 0.2 * k, 0.1 * k))
        if k == 2:
            extra = Cube(); extra.setPosition((0.0, 1.3, 0.5)); extra.setMaterial(sc.getObjects()["c"].materials[0])
            sc.addToScene("extra", extra)
        st.frameCount = k
        ref, _, _ = orc.render(sc.flatten(), cam, st)
        rt.raytrace(sc, cam, fb)
        assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref), k
    ctx.framebuffer_destroy(fb)


@pytest.mark.parametrize("lanes,chunk", [(4, 1024), (3, 512), (1, 256)])
def test_every_pixel_flagged_in_every_chunk(ctx, lanes, chunk):
    """A flat quad in front of the scene: the BVH finds it, the reference's object cull rejects it
    (zero-thickness box, strict '<': SURVEY 8a-Q4), so every pixel it covers is flagged and goes through the
    chunk's redo list -- with the frame cut into many interleaved chunks over several lanes, each list
    must hold all the pixels of its chunk."""
    quad = Mesh(np.array([-6, -6, 0, 6, -6, 0, 6, 6, 0, -6, 6, 0], np.float32), np.array([0, 1, 2, 0, 2, 3], np.int32),
                np.array([0, 0, 1, 0, 1, 1, 0, 1], np.float32), None)
    sc, cam = small_scene()
    sc.addToScene("veil", MeshObject([(quad, 0)], [Material("v")], position=(0.0, 0.0, 2.5)))
    cam.setWidth(83); cam.setHeight(37); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=4, frameCount=2)
    flat = sc.flatten()
    ref, rid, nq = orc.render(flat, cam, st)
    assert not (rid[..., 0] == 3).any() and (rid[..., 0] != 0xFFFFFFFF).any()
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.set_option(capi.OPT_WF_LANES, lanes)
    ctx.set_option(capi.OPT_WF_CHUNK, chunk)
    try:
        ctx.upload_scene(flat)
        ctx.set_camera_rays(cam.preTransformRays)
        fb = ctx.framebuffer(cam.width, cam.height, ids=True)
        ctx.render(cam, st, fb)
        assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)
        assert np.array_equal(ctx.read_ids(fb, cam.width, cam.height), rid)
        assert ctx.stats().rays == nq
        ctx.framebuffer_destroy(fb)
    finally:
        ctx.set_option(capi.OPT_WF_LANES, 0)
        ctx.set_option(capi.OPT_WF_CHUNK, 0)


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["bvh-wavefront", "brute", "bvh-persistent"])
def test_render_bands_compose_the_frame(ctx, mode):
    """rt_render_bands: the interleaved row bands of n callers (one pass of the pipeline each) compose the
    rt_render frame bit for bit -- ragged last band, a sub-rectangle, more ranks than bands, progressive blend."""
    sc, cam = small_scene()
    cam.setWidth(150); cam.setHeight(117); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=3, frameCount=4, progressiveFrameCount=2)
    flat = sc.flatten()
    prev = np.random.default_rng(2).random((cam.height, cam.width, 4)).astype(np.float32)
    ctx.set_option(capi.OPT_TRAVERSAL, mode)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    fb2 = ctx.framebuffer(cam.width, cam.height, ids=True)
    for rect in (None, (11, 5, 140, 101)):
        ctx.write(fb, prev); ctx.write(fb2, prev)
        ctx.render(cam, st, fb, rect)
        want, wid, rays = ctx.read(fb, cam.width, cam.height), ctx.read_ids(fb, cam.width, cam.height), ctx.stats().rays
        for band_rows, world in ((16, 3), (8, 5), (64, 4)):
            ctx.write(fb2, prev)
            total = 0
            for r in range(world):
                ctx.render(cam, st, fb2, rect, bands=(band_rows, r, world))
                total += ctx.stats().rays
            assert np.array_equal(ctx.read(fb2, cam.width, cam.height), want), (rect, band_rows, world)
            got_ids = ctx.read_ids(fb2, cam.width, cam.height)
            x0, y0, x1, y1 = rect or (0, 0, cam.width, cam.height)
            assert np.array_equal(got_ids[y0:y1, x0:x1], wid[y0:y1, x0:x1]) and total == rays
    ref, _, _ = orc.render(flat, cam, st, prev=prev)
    ctx.write(fb, prev)
    ctx.render(cam, st, fb)
    assert np.array_equal(ctx.read(fb, cam.width, cam.height), ref)
    with pytest.raises(capi.RtError, match="multiple of 8"):
        ctx.render(cam, st, fb, None, bands=(12, 0, 2))
    ctx.framebuffer_destroy(fb); ctx.framebuffer_destroy(fb2)


def test_framebuffer_write_right_after_create(ctx):
    """rt_framebuffer_create clears the buffer in stream order: a write (or a render) that follows at once must not be
    overtaken by the clear.  (It ran on the null stream once, which the context's non-blocking stream does not wait for:
    2 of 28 000 random frames of tests/soak_parity.py lost a 4 KB page of their previous frame.)"""
    rng = np.random.default_rng(9)
    for w, h in ((123, 16), (82, 51), (640, 360), (31, 7)):
        data = rng.random((h, w, 4)).astype(np.float32)
        for _ in range(1500):
            fb = ctx.framebuffer(w, h, ids=True)
            ctx.write(fb, data)
            got = ctx.read(fb, w, h)
            ctx.framebuffer_destroy(fb)
            assert np.array_equal(got, data)


def test_present_and_screenshot(ctx, tmp_path):
    """rt_present: the RGBA32F texels the reference's window uploads for display (RenderWindow.cpp:172-176, 282-283) and the
    flipped sRGB8 screenshot (RenderWindow.cpp:389-421), converted on the GPU, byte for byte the host restatement --
    on a rendered frame (its sky is the reference's one known answer, sRGB8 (104,104,139)) and on adversarial values."""
    from raytracer_amd import present
    scene, cam, st = scenes.config1(96, 64, 2, 3)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(scene.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(96, 64)
    ctx.render(cam, st, fb)
    lin = ctx.read(fb, 96, 64)
    assert np.array_equal(present.present_rgba32f(ctx, fb, 96, 64), lin)
    assert np.array_equal(present.present_rgba32f(ctx, fb, 96, 64, flip=True), lin[::-1])
    table = present.thresholds(ctx.L)
    assert np.all(np.diff(table) > 0) and np.abs(table.view(np.int32) - present.thresholds_python().view(np.int32)).max() <= 1
    shot = present.screenshot(ctx, fb, 96, 64)
    assert np.array_equal(shot, present.to_srgb8(lin, table, flip=True))
    assert tuple(shot[0, 0]) == (104, 104, 139, 255)                       # top-left: sky (README screenshots of the reference)
    assert np.array_equal(present.screenshot(ctx, fb, 96, 64, flip=False), present.to_srgb8(lin, table, flip=False))
    present.save_png(str(tmp_path / "s.png"), shot)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "s.png"))), shot)
    # every threshold, its neighbours, and values outside [0, 1] / non-finite
    t = np.concatenate([table, np.nextafter(table, np.float32(-1)), np.nextafter(table, np.float32(2)),
                        np.array([0.0, -0.0, 1.0, -1.0, 2.0, np.inf, -np.inf, np.nan, 1e-30, 0.5], np.float32)]).astype(np.float32)
    img = np.zeros((64, 96, 4), np.float32)
    img.reshape(-1, 4)[:t.size, 0] = t
    img.reshape(-1, 4)[:t.size, 1] = t[::-1]
    img.reshape(-1, 4)[:t.size, 3] = np.clip(t, 0, 1)
    ctx.write(fb, img)
    got = present.screenshot(ctx, fb, 96, 64)
    assert np.array_equal(got, present.to_srgb8(img, table, flip=True))
    # against the textbook formula on random values (away from the thresholds the two agree on every byte)
    rnd = np.random.default_rng(9).random((64, 96, 4)).astype(np.float32)
    ctx.write(fb, rnd)
    got = present.screenshot(ctx, fb, 96, 64, flip=False)
    x = rnd[..., :3].astype(np.float64)
    enc = np.where(x <= 0.0031308, 12.92 * x, 1.055 * x ** (1 / 2.4) - 0.055)
    assert np.array_equal(got[..., :3], np.floor(enc * 255.0 + 0.5).astype(np.uint8))
    ctx.framebuffer_destroy(fb)
