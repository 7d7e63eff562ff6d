"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bars: primary hit ids and the all-segment checksum bit-exact; linear
radiance max-abs <= 1e-3 and PSNR >= 60 dB (in practice the two agree bit for bit)."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from raytracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 999.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def render_gpu(ctx, flat, cam, st, mode, rect=None, prev=None):
    ctx.set_option(capi.OPT_TRAVERSAL, mode)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    if prev is not None:
        ctx.write(fb, prev)
    ctx.render(cam, st, fb, rect)
    img = ctx.read(fb, cam.width, cam.height)
    ids = ctx.read_ids(fb, cam.width, cam.height)
    stats = ctx.stats()
    ctx.framebuffer_destroy(fb)
    return img, ids, stats


def check(ref, rid, nq, img, ids, stats):
    assert np.array_equal(ids[..., :3], rid[..., :3]), "primary hit ids differ"
    assert np.array_equal(ids[..., 3], rid[..., 3]), "path checksum differs"
    assert stats.rays == nq
    err = np.abs(img[..., :3] - ref[..., :3]).max()
    assert err <= 1e-3, f"max abs radiance error {err}"
    assert psnr(img[..., :3], ref[..., :3]) >= 60.0


@pytest.mark.parametrize("mode", [1, 0, 2], ids=["brute", "bvh-wavefront", "bvh-persistent"])
@pytest.mark.parametrize("spp,depth", [(1, 4), (2, 3), (3, 4), (4, 4), (6, 2), (9, 4)])
def test_config1(ctx, mode, spp, depth):
    scene, cam, st = scenes.config1(256, 256, spp, depth)
    flat = scene.flatten()
    ref, rid, nq = orc.render(flat, cam, st)
    img, ids, stats = render_gpu(ctx, flat, cam, st, mode)
    check(ref, rid, nq, img, ids, stats)
    assert np.array_equal(img, ref), "radiance is expected to match bit for bit"
    if mode == 0 and depth > 0:
        # the camera rays went through the packet kernel: one node fetch serves up to 64 box tests
        assert 0 < stats.packet_node_tests <= stats.node_visits
        assert stats.packet_node_fetches <= stats.packet_node_tests <= 64 * stats.packet_node_fetches
    elif mode != 0:
        assert stats.packet_node_tests == 0 and stats.packet_node_fetches == 0


import os

from raytracer_amd.scene import Camera, Cube, Material, Mesh, MeshObject, RenderSettings, Scene, Sphere, Texture

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


def compare_modes(ctx, flat, cam, st, exact=True, modes=(1, 0, 2)):
    ref, rid, nq = orc.render(flat, cam, st)
    stats = None
    for mode in modes:
        img, ids, stats = render_gpu(ctx, flat, cam, st, mode)
        check(ref, rid, nq, img, ids, stats)
        if exact:
            assert np.array_equal(img, ref, equal_nan=True)
    return ref, rid, stats


@pytest.mark.parametrize("mode", [1, 0, 2], ids=["brute", "bvh-wavefront", "bvh-persistent"])
def test_committed_goldens(ctx, mode):
    for name, (spp, depth, aa) in {"c1_s1": (1, 4, True), "c1_s5": (5, 3, True), "c1_noaa": (1, 2, False)}.items():
        scene, cam, st = scenes.config1(64, 64, spp, depth)
        cam.setPosition((0.6, 0.3, 4.0))
        st.antiAliasingEnabled = aa
        st.frameCount = 3
        img, ids, stats = render_gpu(ctx, scene.flatten(), cam, st, mode)
        assert np.array_equal(ids, GOLD[name + "_ids"]) and np.array_equal(img, GOLD[name + "_fb"])
        assert stats.rays == int(GOLD[name + "_rays"][0])
    scene, cam, st = scenes.config2(96, 54, 4, 4, 64)
    img, ids, stats = render_gpu(ctx, scene.flatten(), cam, st, mode)
    assert np.array_equal(ids, GOLD["c2_ids"]) and np.array_equal(img, GOLD["c2_fb"])


def pbr_scene(seed=3, n_obj=24, tex=64):
    """Random rotated / non-uniformly scaled instances with all four texture kinds bound."""
    rng = np.random.default_rng(seed)
    mats = []
    for k in range(4):
        m = Material(f"m{k}", tuple(rng.random(3) * 0.8 + 0.1) + (1.0,), float(rng.random()), float(rng.random()),
                     tuple(rng.random(3)), float(rng.random() * (k == 3)))
        if k != 1:
            m.setAlbedoTexture(Texture(scenes.procedural_texture(tex, seed + k, "albedo")))
        if k % 2 == 0:
            m.setNormalTexture(Texture(scenes.procedural_texture(tex, seed + 10 + k, "normal")))
        # (the roughness and metal bindings read a map's first channel only, from a copy of that channel: two of the maps are
        # colour images whose channels differ, so that the channel read matters)
        if k < 2:
            m.setRoughnessTexture(Texture(scenes.procedural_texture(tex // 2, seed + 20 + k, "scalar")[..., 0]))
        elif k == 2:
            m.setRoughnessTexture(Texture(scenes.procedural_texture(tex // 2, seed + 20 + k, "albedo")))
        if k in (1, 2):
            m.setMetalTexture(Texture(scenes.procedural_texture(tex, seed + 30 + k, "scalar")[..., :3]))
        elif k == 3:
            m.setMetalTexture(Texture(scenes.procedural_texture(tex, seed + 30 + k, "albedo")[..., :3]))
        mats.append(m)
    ell = scenes.merge_meshes([scenes.ellipsoid_mesh(0.5, 0.35, 0.45, 12, 8)])
    box = scenes.merge_meshes([scenes.box_mesh((-0.4, -0.3, -0.5), (0.4, 0.3, 0.5))])
    sc = Scene()
    for i in range(n_obj):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pos = rng.uniform(-3, 3, 3) * (1, 0.6, 1)
        scl = rng.uniform(0.5, 1.8, 3)
        kind = i % 4
        if kind == 0:
            o = Sphere(pos, q, scl); o.setMaterial(mats[i % 4])
        elif kind == 1:
            o = Cube(); o.setPosition(pos); o.setRotation(q); o.setScale(scl); o.setMaterial(mats[(i // 4) % 4])
        elif kind == 2:
            o = MeshObject([(ell, 1), (box, 0)], [mats[i % 4], mats[(i + 1) % 4]], pos, q, scl)
        else:
            o = MeshObject([(box, 0)], [mats[(i + 2) % 4]], pos, q, scl)
        sc.addToScene(f"o{i}", o)
    q = np.array([-0.1, 0.05, 0.0, 0.99], np.float32); q /= np.linalg.norm(q)
    cam = Camera((0.3, 1.0, 8.0), q, 50.0, 1.3)
    return sc, cam


@pytest.mark.parametrize("spp,depth,aa", [(1, 4, True), (4, 3, True), (2, 5, False)])
def test_textured_transformed_instances(ctx, spp, depth, aa):
    sc, cam = pbr_scene()
    cam.setWidth(160); cam.setHeight(96); cam.calcRays()
    st = RenderSettings(bounceCount=depth, sampleCount=spp, antiAliasingEnabled=aa, frameCount=11)
    compare_modes(ctx, sc.flatten(), cam, st)


def test_random_scene_soak(ctx):
    """48 seeds of tests/soak_parity.py: random scenes, cameras, frame sizes, depths, maxDistance, progressive blends and
    sub-rectangles, 1 500 ray queries each (random, aimed at vertices and edge midpoints, axis-parallel), all three modes, bit for bit (the tool runs any range of seeds: 20 000 of them in all three modes were
    clean once the framebuffer clear it caught racing with rt_framebuffer_write had been put in stream order)."""
    import soak_parity
    # ... and the seeds whose ray queries first showed hits lying before their object's box entry (rt_trace.h, trace_bvh)
    early = [1365, 1833, 1911, 1941, 2011, 3237, 3862, 3924, 4443, 4599, 4731, 5917]
    # ... the seeds whose collapsed objects (two axes scaled to zero: collinear triangles) showed hits no box can stand for
    # (rt_commit.hip, thin_kernel), and a few of the later styles (axis-aligned grid scenes, collapsed axes)
    thin = [106695, 108439, 108511] + list(range(100000, 100016)) + list(range(200000, 200020))   # + the reference's meshes in random poses, far-out scenes
    bad = {seed: why for seed in list(range(48)) + early + thin for why in [soak_parity.run_seed(ctx, seed, (0, 2, 1))] if why}
    assert not bad, bad


def test_random_scene_soak_large_frames(ctx):
    """24 seeds of the same tool with the frame extents multiplied by six (frames of several hundred pixels a side: many
    pieces per stripe, full shade batches, the noalloc / bands / chunks variants at those sizes), BVH pipelines, bit for bit."""
    import soak_parity
    soak_parity.SCALE = 6
    try:
        bad = {seed: why for seed in list(range(16)) + [100000, 100006, 100007, 200000, 200008, 200009, 200018, 200019]
               for why in [soak_parity.run_seed(ctx, seed, (0, 2))] if why}
    finally:
        soak_parity.SCALE = 1
    assert not bad, bad


def test_live_scene_sequence_soak(ctx):
    """16 seeds of the sequence form of tests/soak_parity.py: a live scene edited between frames (objects moved, removed,
    added, material constants and maps changed, camera moved, progressive rendering toggled) through the GPURaytracer
    mirror, seven frames each, every frame bit for bit (1 500 seeds were clean on the build it came with)."""
    import soak_parity
    bad = {seed: why for seed in range(16) for why in [soak_parity.run_sequence(ctx, seed)] if why}
    assert not bad, bad


@pytest.mark.parametrize("seed", [5, 6])
def test_camera_rays_aimed_at_collinear_triangles(ctx, seed):
    """Primary-hit parity on collinear triangles by construction (tests/thin_scene.py): anti-aliasing off, and for 195 pixels an
    object placed so that the line its collinear triangles lie on passes through the pixel's ray at a vertex -- needles (84
    overlapping triangles on one axis, ordinary rigid instances) and instances of the reference's sword (ten such triangles).
    Per the oracle several of those pixels see a sliver at a distance that lies outside the stretch of the ray the sliver's
    own bounding box holds, which a walk of the BVH has no reason to reach: wf_gen / path_kernel test every camera ray
    against the listed lines and send such samples to the exact walks."""
    import thin_scene
    sc, cam, st, aimed = thin_scene.aimed_scene(seed)
    flat = sc.flatten()
    assert len(thin_scene.beyond_any_box(sc, cam, st, aimed)) >= 5
    compare_modes(ctx, flat, cam, st)
    # ... and with anti-aliasing on (rays that pass the lines by a jitter's width), several samples
    st.antiAliasingEnabled = True; st.sampleCount = 3
    compare_modes(ctx, flat, cam, st)
    # The tree alone (round 3): the box a walk sees for a sliver is grown by what rounding noise can displace the hit the
    # reference reports (rt_commit.hip, tri_box_walk), so with every listed-line test switched off the walks still reach
    # those tests.  (With ungrown boxes and the tests off this fails on both seeds: tools/thin_teeth.sh.)
    ctx.set_option(capi.OPT_THIN_LIST, 0)
    try:
        compare_modes(ctx, flat, cam, st)
        st.antiAliasingEnabled = False; st.sampleCount = 1
        compare_modes(ctx, flat, cam, st)
    finally:
        ctx.set_option(capi.OPT_THIN_LIST, 1)


def random_rays(rng, n, extent=4.0):
    o = rng.uniform(-extent * 2, extent * 2, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = o; r[:, 3] = rng.choice([100.0, 100.0, 7.5, 3.0], n); r[:, 4:7] = d
    return r


def _world_vertices(flat):
    verts = []
    for o in flat.objects:
        m = np.asarray(o["model"], np.float64).reshape(4, 4).T
        for mid in o["meshIds"]:
            v = flat.meshes[mid].vertices.reshape(-1, 3).astype(np.float64)
            verts.append((v @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
    return np.concatenate(verts)


def adversarial_rays(flat, rng, n):
    """Rays aimed exactly at world-space vertices and edge midpoints, and axis-parallel rays."""
    verts = _world_vertices(flat)
    a = verts[rng.integers(0, len(verts), n)]
    b = verts[rng.integers(0, len(verts), n)]
    tgt = np.where(rng.random((n, 1)) < 0.5, a, (0.5 * (a + b)).astype(np.float32))
    o = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
    axis = rng.random(n) < 0.25
    k = rng.integers(0, 3, n)
    o[axis] = tgt[axis]
    o[axis, k[axis]] += rng.choice([-7.0, 7.0], axis.sum()).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = o; r[:, 3] = 100.0; r[:, 4:7] = d.astype(np.float32)
    return r


def test_trace_rays_random_and_adversarial(ctx):
    sc, _ = pbr_scene(seed=9, n_obj=40)
    flat = sc.flatten()
    rng = np.random.default_rng(1)
    rays = np.concatenate([random_rays(rng, 60000), adversarial_rays(flat, rng, 60000)])
    ref = orc.trace_rays(flat, rays)
    try:
        for mode, builder in ((1, 0), (0, 0), (0, 1)):
            ctx.set_option(capi.OPT_TRAVERSAL, mode)
            ctx.set_option(capi.OPT_BUILDER, builder)
            ctx.upload_scene(flat)
            got = ctx.trace_rays(rays)
            bad = np.nonzero((got != ref).any(axis=1))[0]
            assert bad.size == 0, (mode, builder, bad[:5], got[bad[:5]], ref[bad[:5]])
    finally:
        ctx.set_option(capi.OPT_BUILDER, 0)
    assert (ref[:, 0] != 0xFFFFFFFF).mean() > 0.3


def touching_scene():
    """Objects that touch or coincide: coplanar shared faces, exact duplicates (z-fighting) and a
    sphere resting on a slab.  Exercises the in-order exact path of the BVH traversal."""
    sc = Scene()
    m = Material("a", (0.7, 0.3, 0.3, 1), 0.4, 0.2)
    n = Material("b", (0.3, 0.7, 0.3, 1), 0.6, 0.0)
    for i, x in enumerate((-1.0, 0.0, 1.0)):   # three unit cubes sharing faces
        c = Cube(); c.setPosition((x, 0.0, 0.0)); c.setMaterial(m if i % 2 else n); sc.addToScene(f"c{i}", c)
    d = Cube(); d.setPosition((0.0, 0.0, 0.0)); d.setMaterial(m); sc.addToScene("dup", d)          # exact duplicate of c1
    e = Cube(); e.setPosition((0.0, 1.0, 0.0)); e.setMaterial(n); sc.addToScene("top", e)          # stacked, shared face
    s = Sphere(position=(0.0, 2.0, 0.0)); s.setMaterial(m); sc.addToScene("ball", s)                # rests on `top`
    s2 = Sphere(position=(0.0, 2.0, 0.0)); s2.setMaterial(n); sc.addToScene("ball2", s2)            # duplicate sphere
    return sc


def test_touching_and_duplicate_objects_take_the_exact_path(ctx):
    sc = touching_scene()
    cam = Camera((1.2, 2.0, 6.0), (-0.12, 0.08, 0.0, 0.99), 45.0, 1.0)
    cam.setWidth(128); cam.setHeight(128); cam.calcRays()
    st = RenderSettings(bounceCount=4, sampleCount=2)
    flat = sc.flatten()
    _, _, stats = compare_modes(ctx, flat, cam, st)
    assert stats.slow_path > 0
    rng = np.random.default_rng(4)
    rays = np.concatenate([random_rays(rng, 40000, 2.0), adversarial_rays(flat, rng, 40000)])
    ref = orc.trace_rays(flat, rays)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    got = ctx.trace_rays(rays)
    assert np.array_equal(got, ref)


def _check_frames_and_rays(ctx, sc, cams, rng, ray_extent, ray_center, n_rays=30000, spp=2, depth=4):
    flat = sc.flatten()
    for cam in cams:
        st = RenderSettings(bounceCount=depth, sampleCount=spp, maxDistance=1e6)
        compare_modes(ctx, flat, cam, st)
    # random rays through the neighbourhood of ray_center, and rays from around it aimed at vertices / edge midpoints
    center = np.asarray(ray_center, np.float32)
    a = random_rays(rng, n_rays, ray_extent)
    a[:, :3] += center
    b = np.zeros((n_rays, 8), np.float32)
    o = (center + rng.uniform(-2 * ray_extent, 2 * ray_extent, (n_rays, 3))).astype(np.float32)
    verts = _world_vertices(flat)
    tgt = verts[rng.integers(0, len(verts), n_rays)]
    d = tgt - o
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    b[:, :3] = o; b[:, 4:7] = d.astype(np.float32)
    rays = np.concatenate([a, b])
    rays[:, 3] = 1e6
    ref = orc.trace_rays(flat, rays)
    for mode, builder in ((1, 0), (0, 0), (0, 1)):
        ctx.set_option(capi.OPT_TRAVERSAL, mode)
        ctx.set_option(capi.OPT_BUILDER, builder)
        try:
            ctx.upload_scene(flat)
            got = ctx.trace_rays(rays)
        finally:
            ctx.set_option(capi.OPT_BUILDER, 0)
        bad = np.nonzero((got != ref).any(axis=1))[0]
        assert bad.size == 0, (mode, builder, bad[:5], got[bad[:5]], ref[bad[:5]])
    return ref


def _cam(pos, quat, w=112, h=80, fov=45.0):
    cam = Camera(pos, quat, fov, 1.0)
    cam.setWidth(w); cam.setHeight(h); cam.calcRays()
    return cam


def test_extreme_scales_in_one_scene(ctx):
    """Non-uniform scales from 1e-3 to 1e3 inside one scene, rotated: cond(model) up to 1e6, so the object-space
    cull (GPURaytracer.cu:267-315, through the host's float inverse) and the world-space triangle test round very
    differently.  rt_scene_commit widens the tie band with the worst cond (band_rel_of), which sends every
    doubtful ray through the exact in-order emulation; result: same bits as brute force and as the oracle."""
    rng = np.random.default_rng(21)
    m = [Material("a", (0.7, 0.3, 0.3, 1), 0.4, 0.2), Material("b", (0.3, 0.7, 0.3, 1), 0.6, 0.0)]
    sc = Scene()
    def add(name, o, mat):
        o.setMaterial(mat); sc.addToScene(name, o)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    add("slab", Sphere((0.0, -2.0, 0.0), (0, 0, 0, 1), (1e3, 1.0, 1e3)), m[0])               # a 1000-unit pancake
    add("needle", Sphere((0.5, 0.0, 0.0), q, (1e-3, 4.0, 1e-3)), m[1])                         # a 4-unit needle 1 mm thick
    c = Cube(); c.setPosition((-1.0, 0.0, 0.5)); c.setRotation(q); c.setScale((1e-3, 2.0, 1e3))   # a 1000 x 2 sheet 1 mm thick
    add("sheet", c, m[0])
    c = Cube(); c.setPosition((1.5, -0.5, 0.0)); c.setScale((1e-2, 1e-2, 1e-2)); add("grain", c, m[1])
    c = Cube(); c.setPosition((1.5, -0.5 + 0.005 + 0.5, 0.0)); add("on-grain", c, m[0])          # touches the grain's top face
    add("big", Sphere((0.0, 0.0, -600.0), (0, 0, 0, 1), (1e3, 1e3, 1e3)), m[1])
    add("in-big", Sphere((0.0, 0.0, -100.0), q, (3.0, 1e-2, 3.0)), m[0])                        # touches nothing, sits in front of `big`
    cams = [_cam((0.4, 0.6, 6.0), (-0.05, 0.03, 0.0, 0.998)), _cam((1.5, -0.2, 0.8), (-0.2, 0.0, 0.0, 0.98), fov=30.0),
            _cam((40.0, 25.0, 60.0), (-0.17, 0.28, 0.05, 0.94))]
    ref = _check_frames_and_rays(ctx, sc, cams, rng, 6.0, (0, 0, 0))
    assert (ref[:, 0] != 0xFFFFFFFF).mean() > 0.3 and len(np.unique(ref[:, 0])) >= 7


def test_scene_far_from_the_origin(ctx):
    """The touching / duplicate-object scene and a random textured one 1e4 units from the origin on every axis:
    one float ulp is 1e-3 there, the inverse model matrices cancel against 1e4, and shared faces still have
    to resolve to the reference's object order."""
    rng = np.random.default_rng(22)
    off = np.array([1.0e4, -1.0e4, 1.0e4], np.float32)
    sc = touching_scene()
    extra, _ = pbr_scene(seed=4, n_obj=10, tex=32)
    for k, (name, o) in enumerate(extra.getObjects().items()):
        sc.addToScene("x" + name, o)
    for name, o in sc.getObjects().items():
        o.setPosition(np.asarray(o.Position, np.float32) + off)
    cams = [_cam(tuple(off + np.array([1.2, 2.0, 6.0], np.float32)), (-0.12, 0.08, 0.0, 0.99)),
            _cam(tuple(off + np.array([0.3, 1.0, 9.0], np.float32)), (-0.05, 0.0, 0.0, 0.999), fov=50.0)]
    ref = _check_frames_and_rays(ctx, sc, cams, rng, 4.0, off)
    assert (ref[:, 0] != 0xFFFFFFFF).mean() > 0.2


def test_camera_inside_a_huge_box(ctx):
    """The camera inside a 2e4-unit cube (scale 2e4 on the unit cube: every ray ends on it) with ordinary and
    tiny objects around it, some touching the cube's floor."""
    rng = np.random.default_rng(23)
    m = [Material("wall", (0.6, 0.6, 0.7, 1), 0.8, 0.0), Material("o", (0.8, 0.4, 0.2, 1), 0.3, 0.5)]
    sc = Scene()
    box = Cube(); box.setScale((2e4, 2e4, 2e4)); box.setMaterial(m[0]); sc.addToScene("room", box)
    c = Cube(); c.setPosition((0.0, -1e4 + 0.5, -4.0)); c.setMaterial(m[1]); sc.addToScene("on-floor", c)      # rests on the room's floor
    s = Sphere((1.5, -1e4 + 1.0, -5.0), (0, 0, 0, 1), (2.0, 2.0, 2.0)); s.setMaterial(m[1]); sc.addToScene("ball", s)
    t = Cube(); t.setPosition((-1.0, -1e4 + 1.5, -3.0)); t.setScale((1e-2, 1.0, 1e-2)); t.setMaterial(m[0]); sc.addToScene("post", t)
    cams = [_cam((0.0, -1e4 + 1.6, 2.0), (-0.05, 0.0, 0.0, 0.999)), _cam((3.0, -1e4 + 4.0, 1.0), (-0.25, 0.2, 0.05, 0.945))]
    ref = _check_frames_and_rays(ctx, sc, cams, rng, 5.0, (0.0, -1e4 + 2.0, -3.0))
    assert (ref[:, 0] != 0xFFFFFFFF).mean() > 0.9


def test_edge_cases(ctx):
    cam = Camera((0, 0, 5)); cam.setWidth(40); cam.setHeight(24); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=1)
    # empty scene
    compare_modes(ctx, Scene().flatten(), cam, st)
    # a flat quad is invisible (strict slab test); a single triangle inside generous bounds is not
    quad = Mesh(np.array([-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0], np.float32), np.array([0, 1, 2, 0, 2, 3], np.int32),
                np.array([0, 0, 1, 0, 1, 1, 0, 1], np.float32), None)
    m = Material("m")
    sc = Scene(); sc.addToScene("q", MeshObject([(quad, 0)], [m]))
    _, rid, _ = compare_modes(ctx, sc.flatten(), cam, st)
    assert (rid[..., 0] == 0xFFFFFFFF).all()
    tri = Mesh(np.array([-1, -1, 0, 1, -1, 0, 0, 1, 0], np.float32), np.array([0, 1, 2], np.int32),
               np.array([0, 0, 1, 0, 0.5, 1], np.float32), None)
    o = MeshObject([(tri, 0)], [m])
    o.compositeMinBounds = np.array([-1, -1, -0.5], np.float32); o.compositeMaxBounds = np.array([1, 1, 0.5], np.float32)
    sc = Scene(); sc.addToScene("t", o)
    _, rid, _ = compare_modes(ctx, sc.flatten(), cam, st)
    assert (rid[..., 0] == 0).any()
    # degenerate UVs: NaN tangent frame flows into the shading normal exactly as in the reference
    slab = scenes.box_mesh((-1, -1, -0.1), (1, 1, 0.1))
    mesh = Mesh(slab[0].reshape(-1), slab[1], np.zeros(slab[2].size, np.float32), slab[3].reshape(-1))
    sc = Scene(); sc.addToScene("s", MeshObject([(mesh, 0)], [m]))
    compare_modes(ctx, sc.flatten(), cam, st)
    # a mesh without a UV array (the reference would fault; defined as all-zero UVs)
    mesh = Mesh(slab[0].reshape(-1), slab[1], None, slab[3].reshape(-1))
    sc = Scene(); sc.addToScene("s", MeshObject([(mesh, 0)], [m]))
    compare_modes(ctx, sc.flatten(), cam, st)
    # camera inside an object, tiny maxDistance, bounceCount 1 and 0
    sc, cam2, st2 = scenes.config1(48, 48, 1, 1)
    cam2.setPosition((0.0, 0.0, 0.2))
    compare_modes(ctx, sc.flatten(), cam2, st2)
    st2.bounceCount = 0
    compare_modes(ctx, sc.flatten(), cam2, st2)
    st2.bounceCount = 3; st2.maxDistance = 9.6; cam2.setPosition((0.0, 0.0, 10.0))
    compare_modes(ctx, sc.flatten(), cam2, st2)


def test_tiles_and_progressive(ctx):
    scene, cam, st = scenes.config2(192, 108, 4, 4, 64)
    flat = scene.flatten()
    ref, rid, nq = orc.render(flat, cam, st)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(192, 108, ids=True)
    total = 0
    for rect in [(0, 0, 100, 50), (100, 0, 192, 50), (0, 50, 37, 108), (37, 50, 192, 108)]:  # ragged tiles
        ctx.render(cam, st, fb, rect)
        total += ctx.stats().rays
    assert np.array_equal(ctx.read(fb, 192, 108), ref) and np.array_equal(ctx.read_ids(fb, 192, 108), rid) and total == nq
    # progressive running mean over three frames equals the oracle's
    acc = ref
    for k in (1, 2):
        st.frameCount = k; st.progressiveFrameCount = k
        acc, _, _ = orc.render(flat, cam, st, prev=acc)
        ctx.render(cam, st, fb)
    assert np.array_equal(ctx.read(fb, 192, 108), acc)
    ctx.framebuffer_destroy(fb)


def test_detmath_matches_oracle_bit_for_bit(ctx):
    L = orc.lib()
    rng = np.random.default_rng(8)
    u = rng.random(200000).astype(np.float32)
    cases = [(0, u, L.or_logf), (0, np.array([0.0, 1.0, 1e-45, np.inf, -1.0], np.float32), L.or_logf),
             (1, (u * np.float32(6.283184)).astype(np.float32), L.or_cosf),
             (1, (u * 2000 - 1000).astype(np.float32), L.or_cosf),
             (2, (-u * 110).astype(np.float32), L.or_expf), (2, (u * 95).astype(np.float32), L.or_expf),
             (3, u, lambda x: L.or_powf(x, 5.0)), (4, (1 - u * np.float32(0.03)).astype(np.float32), lambda x: L.or_powf(x, 1000.0)),
             (4, u[:50000], lambda x: L.or_powf(x, 1000.0)),                                         # the whole unit interval: almost all of it underflows to 0
             (4, np.nextafter(np.float32(0.875), np.float32([0, 1, 0.875])), lambda x: L.or_powf(x, 1000.0)),
             (5, (u * 100).astype(np.float32), L.or_rsqrtf)]
    for op, x, f in cases:
        got = ctx.detmath(op, x)
        exp = np.array([f(float(v)) for v in x], np.float32)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), op
    # the two division-free conversions against the divisions they stand for: every byte, every 32-bit value
    b = np.arange(256, dtype=np.uint32)
    assert np.array_equal(ctx.detmath(7, b.view(np.float32)), (b.astype(np.float32) / np.float32(255.0)))
    hi = np.arange(65536, dtype=np.uint32)
    assert ctx.detmath(8, hi.view(np.float32)).sum() == 0.0
    # the two shortcuts of the shading code against the definitions they stand for, for every binary32 in [0, 1]:
    # pow5_f == pow_f(x, 5) (Schlick's term) and pow1000_f == pow_f(x, 1000) (the sun term)
    hi01 = np.arange(0, 0x3F81, dtype=np.uint32)
    assert ctx.detmath(9, hi01.view(np.float32)).sum() == 0.0
    seeds = rng.integers(0, 2**32, 50000, dtype=np.uint64).astype(np.uint32)
    got = ctx.detmath(6, seeds.view(np.float32))
    import ctypes as C
    exp = np.array([L.or_random_value(C.byref(C.c_uint32(int(s)))) for s in seeds], np.float32)
    assert np.array_equal(got, exp)


def test_error_behaviour(ctx):
    c2 = capi.Context(0)
    try:
        cam = Camera(); cam.setWidth(8); cam.setHeight(8); cam.calcRays()
        fb = c2.framebuffer(8, 8)
        with pytest.raises(capi.RtError, match="before rt_scene_commit"):
            c2.render(cam, RenderSettings(), fb)
        c2.upload_scene(Scene().flatten())
        with pytest.raises(capi.RtError, match="ray table"):
            c2.render(cam, RenderSettings(), fb)
        c2.set_camera_rays(cam.preTransformRays)
        with pytest.raises(capi.RtError, match="tile outside"):
            c2.render(cam, RenderSettings(), fb, (0, 0, 9, 8))
        with pytest.raises(capi.RtError, match="invalid framebuffer"):
            c2.render(cam, RenderSettings(), 77)
        with pytest.raises(capi.RtError, match="sampleCount"):
            c2.render(cam, RenderSettings(sampleCount=0), fb)
        c2.render(cam, RenderSettings(), fb)   # still usable after errors
        assert c2.read(fb, 8, 8)[0, 0, 3] == 1.0
    finally:
        c2.close()


def test_full_size_properties(ctx):
    """BASELINE.json configs[1] at full size (1920x1080, 16 spp, depth 4): too large for the CPU
    oracle, so checked through size-independent properties: determinism, tile composite ==
    full frame, brute force == BVH on a band, and the oracle on a centred crop."""
    scene, cam, st = scenes.config2()
    flat = scene.flatten()
    W, H = cam.width, cam.height
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(W, H, ids=True)
    ctx.render(cam, st, fb)
    a, ia, rays = ctx.read(fb, W, H), ctx.read_ids(fb, W, H), ctx.stats().rays
    ctx.render(cam, st, fb)
    assert np.array_equal(ctx.read(fb, W, H), a) and ctx.stats().rays == rays
    fb2 = ctx.framebuffer(W, H, ids=True)
    from raytracer_amd import multigpu
    for r in range(3):
        for rect in multigpu.band_rects(W, H, r, 3):
            ctx.render(cam, st, fb2, rect)
    assert np.array_equal(ctx.read(fb2, W, H), a) and np.array_equal(ctx.read_ids(fb2, W, H), ia)
    # brute force on a band through the middle
    ctx.set_option(capi.OPT_TRAVERSAL, 1)
    ctx.upload_scene(flat)
    band = (0, 500, W, 532)
    ctx.render(cam, st, fb2, band)
    b, ib = ctx.read(fb2, W, H), ctx.read_ids(fb2, W, H)
    assert np.array_equal(b[500:532], a[500:532]) and np.array_equal(ib[500:532], ia[500:532])
    # oracle on a 64x64 crop
    x0, y0 = W // 2 - 32, H // 2 - 32
    ref, rid, _ = orc.render(flat, cam, st, rect=(x0, y0, x0 + 64, y0 + 64))
    assert np.array_equal(ref[y0:y0 + 64, x0:x0 + 64], a[y0:y0 + 64, x0:x0 + 64])
    assert np.array_equal(rid[y0:y0 + 64, x0:x0 + 64], ia[y0:y0 + 64, x0:x0 + 64])
    ctx.framebuffer_destroy(fb); ctx.framebuffer_destroy(fb2)


@pytest.mark.parametrize("name,args,rect", [
    ("config3", (256, 144, 4, 4, 128), None),
    ("config4", (256, 144, 2, 4, 64), None),
    ("config4", (3840, 2160, 1, 4, 64), (1900, 700, 1964, 764)),     # full 4K frame, oracle on a 64x64 crop
    ("config5", (256, 144, 1, 4, 300, 120), None),                   # reduced: 28 800-triangle terrain + 300 spheres
])
def test_larger_configs_against_oracle(ctx, name, args, rect):
    scene, cam, st = getattr(scenes, name)(*args)
    flat = scene.flatten()
    ref, rid, nq = orc.render(flat, cam, st, rect=rect)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    ctx.render(cam, st, fb)
    img, ids, stats = ctx.read(fb, cam.width, cam.height), ctx.read_ids(fb, cam.width, cam.height), ctx.stats()
    ctx.framebuffer_destroy(fb)
    if rect is None:
        check(ref, rid, nq, img, ids, stats)
        assert np.array_equal(img, ref, equal_nan=True)
    else:
        x0, y0, x1, y1 = rect
        assert np.array_equal(ids[y0:y1, x0:x1], rid[y0:y1, x0:x1])
        assert np.array_equal(img[y0:y1, x0:x1], ref[y0:y1, x0:x1], equal_nan=True)
        assert (rid[y0:y1, x0:x1, 0] != 0xFFFFFFFF).any()


@pytest.mark.parametrize("name,crop", [("c3", (856, 888, 904, 936)), ("c4", (2168, 1592, 2216, 1640))])
def test_reference_geometry_at_full_size(ctx, name, crop):
    """BASELINE.json configs[2] and [3] at full size on the reference's own geometry (tests/golden/ref_assets.npz:
    murasama.fbx and model.dae flattened by tests/golden/make_assets.py): C3 3840x2160 x 64 spp, C4 3840x2160 x 4 spp
    over 1 000 944 triangles.  Too large for the CPU oracle, so: the oracle on a 48x48 crop over the model, bit for
    bit; the interleaved bands of three ranks compose the frame; brute force equals the BVH on a strip (C3)."""
    assert scenes.reference_assets() is not None and "reference geometry" in scenes.workload_label(name)
    scene, cam, st = getattr(scenes, "config" + name[1:])()
    assert (cam.width, cam.height, st.sampleCount) == ((3840, 2160, 64) if name == "c3" else (3840, 2160, 4))
    flat = scene.flatten()
    W, H = cam.width, cam.height
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    assert ctx.stats().triangles == (8972 + 12 if name == "c3" else 331 * 3024 + 12)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(W, H, ids=True)
    ctx.render(cam, st, fb)
    a, ia, rays = ctx.read(fb, W, H), ctx.read_ids(fb, W, H), ctx.stats().rays
    x0, y0, x1, y1 = crop
    ref, rid, _ = orc.render(flat, cam, st, rect=crop)
    assert np.array_equal(rid[y0:y1, x0:x1], ia[y0:y1, x0:x1])
    assert np.array_equal(ref[y0:y1, x0:x1], a[y0:y1, x0:x1], equal_nan=True)
    model = rid[y0:y1, x0:x1, 0]
    assert ((model == 0) if name == "c3" else (model < 331)).mean() > 0.25, "the crop should look at the model"
    fb2 = ctx.framebuffer(W, H, ids=True)
    total = 0
    for r in range(3):
        ctx.render(cam, st, fb2, bands=(16, r, 3))
        total += ctx.stats().rays
    assert total == rays and np.array_equal(ctx.read(fb2, W, H), a) and np.array_equal(ctx.read_ids(fb2, W, H), ia)
    if name == "c3":
        ctx.set_option(capi.OPT_TRAVERSAL, 1)
        ctx.upload_scene(flat)
        sx0, sy0, sx1, sy1 = strip = (600, 900, 1100, 908)
        ctx.render(cam, st, fb2, strip)
        b, ib = ctx.read(fb2, W, H), ctx.read_ids(fb2, W, H)
        assert np.array_equal(b[sy0:sy1, sx0:sx1], a[sy0:sy1, sx0:sx1]) and np.array_equal(ib[sy0:sy1, sx0:sx1], ia[sy0:sy1, sx0:sx1])
        assert (ib[sy0:sy1, sx0:sx1, 0] == 0).any()
        ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.framebuffer_destroy(fb); ctx.framebuffer_destroy(fb2)
    ctx.upload_scene(Scene().flatten())


def test_config5_progressive_accumulation_at_full_size(ctx):
    """BASELINE.json configs[4] as stated: the full stress scene (9 999 392-triangle terrain + 10 000 spheres =
    13.5 M triangles, 10 001 objects) at 7680x4320, 256 samples per pixel as 256 progressive frames of 1 spp
    (progressiveFrameCount 0..255, GPURaytracer.cu:129-134).  Checked: the oracle's running mean over all 256 frames on a
    crop of the ACCUMULATED buffer, bit for bit; determinism of the whole sequence; interleaved bands compose a frame;
    and the multi-GPU split (multigpu.frame_indices: each rank sums its own clamped frames, sums added, divided once)
    against the running mean.  Bound of that last comparison: the running mean r' = (c + r n) / (n + 1) commits three
    roundings of a value <= 1 per frame, each damped by n / (n + 1) from then on, which leaves e_N = (1 / N) sum k d_k
    <= N / 2 x 1.8e-7 = 2.3e-5 after N = 256 frames (a third of that in the mean); the sum of 128 clamped frames per
    rank is exact to an ulp of 128 (1.5e-5) per addition before the division by 256.  Stated: 2e-5; measured: about 2e-6."""
    import torch
    from raytracer_amd import multigpu
    scene, cam, st = scenes.config5()
    W, H = cam.width, cam.height
    assert (W, H, st.sampleCount) == (7680, 4320, 1)
    flat = scene.flatten()
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(flat)
    assert ctx.stats().triangles == 9999392 + 10000 * 352 and len(flat.objects) == 10001
    ctx.set_camera_rays(cam.preTransformRays)
    dev = torch.device("cuda", 0)
    run = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)     # running mean, as the reference keeps it
    one = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)     # single frames
    fb_run, fb_one = ctx.framebuffer_wrap(run.data_ptr(), W, H), ctx.framebuffer_wrap(one.data_ptr(), W, H)
    frames, world = 256, 2
    sums = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(world)]
    owner = {f: r for r in range(world) for f in multigpu.frame_indices(0, frames, r, world)}
    rays = []
    for k in range(frames):
        st.frameCount, st.progressiveFrameCount = k, k
        ctx.render(cam, st, fb_run)
        rays.append(ctx.stats().rays)
        st.progressiveFrameCount = 0
        ctx.render(cam, st, fb_one)
        ctx.sync()
        sums[owner[k]] += one
    torch.cuda.synchronize()
    split_mean = (sums[0] + sums[1]) / frames
    split_err = float((split_mean[..., :3] - run[..., :3]).abs().max())
    assert split_err < 2e-5, split_err
    first = run.clone()
    del sums, split_mean
    # the oracle over the same 256 frames on a crop of the accumulated buffer (brute force over 13.5 M triangles per
    # segment: one column of 16 pixels, a row per thread), rendered in place
    x0, y0, x1, y1 = crop = (5074, 207, 5075, 223)   # crosses the silhouette of sphere 2884 against the terrain
    osc = orc.OracleScene(flat)
    acc = np.zeros((H, W, 4), np.float32)
    rid = np.full((H, W, 4), 0xFFFFFFFF, np.uint32)
    for k in range(frames):
        st.frameCount, st.progressiveFrameCount = k, k
        orc.render(osc, cam, st, rect=crop, out=acc, ids_out=rid)
    got = run[y0:y1, x0:x1].cpu().numpy()
    assert np.array_equal(got, acc[y0:y1, x0:x1])
    prim = rid[y0:y1, x0:x1, 0]
    assert (prim != 0xFFFFFFFF).any() and len(np.unique(prim)) >= 2
    del acc, rid, osc
    # determinism: the whole sequence again
    for k in range(frames):
        st.frameCount, st.progressiveFrameCount = k, k
        ctx.render(cam, st, fb_run)
        if k % 16 == 0:
            assert ctx.stats().rays == rays[k]
    ctx.sync()
    assert torch.equal(run, first)
    # one frame as the interleaved bands of three ranks
    st.frameCount, st.progressiveFrameCount = 7, 0
    ctx.render(cam, st, fb_one)
    ctx.sync()
    whole = one.clone()
    one.zero_()
    for r in range(3):
        ctx.render(cam, st, fb_one, bands=(multigpu.BAND_ROWS, r, 3))
    ctx.sync()
    assert torch.equal(one, whole)
    ctx.framebuffer_destroy(fb_run); ctx.framebuffer_destroy(fb_one)
    ctx.upload_scene(Scene().flatten())


def test_custom_model_from_asset_files(ctx, tmp_path):
    """A CustomModel flattened from model files written here (binary FBX + OBJ) renders like the oracle
    in all three modes (the flatteners feed the same upload path)."""
    import test_assets as ta
    from raytracer_amd.scene import CustomModel
    ta.write_fbx(str(tmp_path / "m.fbx"))
    (tmp_path / "q.obj").write_text("v -2 -1 -1\nv 2 -1 -1\nv 2 -1 3\nv -2 -1 3\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 1 0\n"
                                    "f 1/1/1 2/2/1 3/3/1 4/4/1\n")
    sc = Scene()
    a = CustomModel(str(tmp_path / "m.fbx"), position=(-1.0, -0.5, 0.0), scale=(1.2, 1.2, 1.2))
    b = CustomModel(str(tmp_path / "q.obj"))
    assert a.getLoadSuccess() and b.getLoadSuccess()
    sc.addToScene("fbx", a)
    sc.addToScene("floor", b)
    cam = Camera(position=(0.5, 1.5, 6.0), fov=45.0)
    cam.setWidth(96); cam.setHeight(64); cam.calcRays()
    st = RenderSettings(sampleCount=3, bounceCount=3)
    _, rid, _ = compare_modes(ctx, sc.flatten(), cam, st)
    # the floor is flat, hence invisible (object-space cull with strict <): only the FBX model is hit
    assert (rid[..., 0] != 0xFFFFFFFF).mean() > 0.05 and not (rid[..., 0] == 1).any()


@pytest.mark.parametrize("refit", [150, 1, 0], ids=["refit-150", "refit-default", "rebuild"])
@pytest.mark.parametrize("builder", [0, 1], ids=["ploc", "radix-tree"])
def test_dynamic_scene_recommit_matches_oracle(ctx, builder, refit):
    """the reference rebuilds its instance table every frame (Scene.cpp:3-29).  Here the
    transforms are re-committed on the handles already uploaded (no mesh / texture re-upload; the record
    arrays and the build's scratch arena are reused), the acceleration structure of the first commit is refitted
    (RT_OPT_REFIT, the default) or built again, and the frame must match the oracle's for the moved scene -- also
    after a move so large that the refitted boxes pass the growth limit and the commit falls back to a build, after a
    rescale, and in the brute-force and persistent modes on the refitted records."""
    sc, cam = pbr_scene(seed=5, n_obj=12)
    cam.setWidth(128); cam.setHeight(80); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=2)
    flat = sc.flatten()
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.set_option(capi.OPT_BUILDER, builder)   # 1: the builder meant for scenes re-committed every frame
    ctx.set_option(capi.OPT_REFIT, refit)
    try:
        ids = ctx.upload_scene(flat)
        assert ctx.commit_kind() == 1
        ctx.set_camera_rays(cam.preTransformRays)
        fb = ctx.framebuffer(cam.width, cam.height, ids=True)
        first_ms = ctx.stats().commit_ms
        kinds = []
        for step in range(5):
            for k, (name, obj) in enumerate(sc.getObjects().items()):
                p = np.asarray(obj.Position, np.float32)
                if step < 3:
                    obj.setPosition((float(p[0]) + 0.07 * (k % 3 - 1), float(p[1]) + 0.05 * step, float(p[2]) - 0.04 * k))
                elif step == 3:   # scattered far and wide: the tree of the first commit no longer fits
                    obj.setPosition((float(p[0]) * 9.0 + 3.0 * (k % 5), float(p[1]) * 7.0, float(p[2]) * 9.0 - 40.0 * (k % 2)))
                else:             # ... and gathered again, at other sizes
                    obj.setPosition((float(p[0]) / 9.0, float(p[1]) / 7.0, float(p[2]) / 30.0))
                    obj.setScale(tuple(np.asarray(obj.Scale, np.float32) * (0.6 + 0.1 * (k % 4))))
            flat = sc.flatten()
            ctx.commit(ids["obj"], [np.asarray(o["model"], np.float32) for o in flat.objects],
                       [np.asarray(o["invModel"], np.float32) for o in flat.objects])
            kinds.append(ctx.commit_kind())
            st.frameCount = step
            ref, rid, nq = orc.render(flat, cam, st)
            for mode in (0, 2, 1) if step in (1, 4) else (0,):
                ctx.set_option(capi.OPT_TRAVERSAL, mode)
                ctx.render(cam, st, fb)
                img, gid, stats = ctx.read(fb, cam.width, cam.height), ctx.read_ids(fb, cam.width, cam.height), ctx.stats()
                check(ref, rid, nq, img, gid, stats)
                assert np.array_equal(img, ref)
            ctx.set_option(capi.OPT_TRAVERSAL, 0)
        print("commit ms: first %.2f, re-commit %.2f; kinds %s" % (first_ms, ctx.stats().commit_ms, kinds))
        if refit == 150:
            assert kinds[:3] == [2, 2, 2] and 1 in kinds[3:], kinds   # small moves refit; the scatter rebuilds
        elif refit:   # the default growth limit: the first small move is within it, the scatter is not
            assert kinds[0] == 2 and 1 in kinds[3:], kinds
        else:
            assert kinds == [1] * 5, kinds
        ctx.framebuffer_destroy(fb)
    finally:
        ctx.set_option(capi.OPT_BUILDER, 0)
        ctx.set_option(capi.OPT_REFIT, 1)


@pytest.mark.parametrize("lanes,chunk", [(1, 1024), (3, 2048), (4, 1 << 20)])
def test_wavefront_chunks_and_lanes(ctx, lanes, chunk):
    """The wavefront pipeline cut into many chunks over several HIP streams (the sizes a small test frame
    would never reach by itself): same pixels, ids and query count as the oracle, ragged edge tiles included."""
    sc, cam = pbr_scene(seed=7, n_obj=16)
    cam.setWidth(203); cam.setHeight(117); cam.calcRays()
    st = RenderSettings(bounceCount=4, sampleCount=5, frameCount=3)
    flat = sc.flatten()
    ref, rid, nq = orc.render(flat, cam, st)
    ctx.set_option(capi.OPT_WF_LANES, lanes)
    ctx.set_option(capi.OPT_WF_CHUNK, chunk)
    try:
        img, ids, stats = render_gpu(ctx, flat, cam, st, 0)
        check(ref, rid, nq, img, ids, stats)
        assert np.array_equal(img, ref)
        # a sub-rectangle with odd bounds on top of a previous frame (progressive mean)
        rect = (13, 9, 150, 101)
        st.progressiveFrameCount = 2
        prev = np.random.default_rng(1).random((cam.height, cam.width, 4)).astype(np.float32)
        ref2, rid2, nq2 = orc.render(flat, cam, st, rect=rect, prev=prev)
        img2, ids2, stats2 = render_gpu(ctx, flat, cam, st, 0, rect=rect, prev=prev)
        x0, y0, x1, y1 = rect
        assert np.array_equal(img2[y0:y1, x0:x1], ref2[y0:y1, x0:x1]) and stats2.rays == nq2
        assert np.array_equal(ids2[y0:y1, x0:x1], rid2[y0:y1, x0:x1])
        assert np.array_equal(img2[:y0], prev[:y0]) and np.array_equal(img2[:, x1:], prev[:, x1:])
    finally:
        ctx.set_option(capi.OPT_WF_LANES, 0)
        ctx.set_option(capi.OPT_WF_CHUNK, 0)


def test_thousands_of_coincident_triangles(ctx):
    """6 000 copies of one triangle: every box of that subtree is hit by every ray that hits the triangle, all
    hits tie exactly, and the lowest triangle index has to win (8a-Q3) -- through the four-wide BVH with its
    quantised boxes and sorted child order just as in the reference's linear scan."""
    n = 6000
    tri = np.array([[-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [0.0, 1.2, 0.3]], np.float32)
    verts = np.tile(tri, (n, 1))
    idx = np.arange(3 * n, dtype=np.int32)
    pile = Mesh(verts.reshape(-1), idx, None, None)
    sc = Scene()
    sc.addToScene("pile", MeshObject([(pile, 0)], [Material("m", (0.7, 0.5, 0.3, 1.0), 0.6, 0.1)]))
    s = Sphere(position=(1.6, 0.2, -0.5)); s.setMaterial(Material("s", (0.3, 0.6, 0.8, 1.0), 0.3, 0.5))
    sc.addToScene("ball", s)
    cam = Camera((0.2, 0.1, 5.0)); cam.setWidth(96); cam.setHeight(72); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=2)
    for builder in (0, 1):   # equal boxes everywhere: PLOC's tie-break has to pair them up, the radix tree splits by position
        ctx.set_option(capi.OPT_BUILDER, builder)
        try:
            _, rid, _ = compare_modes(ctx, sc.flatten(), cam, st, modes=(0, 2))
        finally:
            ctx.set_option(capi.OPT_BUILDER, 0)
        hit = rid[..., 0] == 0
        assert hit.any() and (rid[..., 2][hit] == 0).all()


def test_axis_parallel_rays(ctx):
    """Rays with exactly zero direction components that run inside the planes shared by many boxes
    (cube faces on integer coordinates, origins on them): the box test must neither lose a hit nor,
    as it once did through NaN slab distances, visit every box in the ray's plane."""
    sc = Scene()
    m = Material("a", (0.7, 0.5, 0.3, 1), 0.5, 0.1)
    k = 0
    for x in range(-6, 6):
        for z in range(-6, 6):
            c = Cube(); c.setPosition((x + 0.5, 0.5 * ((x + z) % 3), z + 0.5)); c.setMaterial(m); sc.addToScene(f"c{k}", c); k += 1
    flat = sc.flatten()
    rng = np.random.default_rng(3)
    n = 30000
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [0, -0.6, -0.8], [0.6, 0, 0.8], [-0.8, 0.6, 0], [0, -0.0, -1.0]], np.float32)
    r = np.zeros((n, 8), np.float32)
    r[:, :3] = np.round(rng.uniform(-8, 8, (n, 3)) * 2) / 2   # origins on the half-integer lattice: on faces and edges
    r[:, 1] += 3.0 * (rng.random(n) < 0.3)
    r[:, 3] = 100.0
    r[:, 4:7] = axes[rng.integers(0, len(axes), n)]
    ref = orc.trace_rays(flat, r)
    for mode in (1, 0):
        ctx.set_option(capi.OPT_TRAVERSAL, mode)
        ctx.upload_scene(flat)
        got = ctx.trace_rays(r)
        assert np.array_equal(got, ref), mode
    assert (ref[:, 0] != 0xFFFFFFFF).mean() > 0.1
    # an axis-aligned camera on the symmetry plane: the centre column looks along x = 0 exactly
    cam = Camera((0.0, 2.0, 14.0), (0.0, 0.0, 0.0, 1.0), 45.0, 1.0)
    cam.setWidth(96); cam.setHeight(64); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=2, antiAliasingEnabled=False)
    _, _, stats = compare_modes(ctx, flat, cam, st, modes=(0, 2))
    assert stats.node_visits < 150 * stats.rays, "box tests per ray out of proportion: axis-parallel rays are visiting whole planes"


def test_non_finite_vertices(ctx):
    """A mesh with NaN and infinite coordinates among ordinary triangles: the reference's tests simply never
    accept such a triangle; the build has to terminate on boxes without a finite area (both builders) and the
    traversal must still find every finite triangle."""
    rng = np.random.default_rng(11)
    n = 60
    v = rng.uniform(-1.5, 1.5, (n * 3, 3)).astype(np.float32)
    v[7, 1] = np.nan; v[40, 0] = np.inf; v[41, 2] = -np.inf; v[100, :] = np.nan; v[150, 0] = np.float32(3e38)
    mesh = Mesh(v.reshape(-1), np.arange(n * 3, dtype=np.int32), rng.random(n * 6).astype(np.float32), None)
    o = MeshObject([(mesh, 0)], [Material("m", (0.6, 0.6, 0.3, 1.0), 0.5, 0.2)])
    o.compositeMinBounds = np.array([-2, -2, -2], np.float32); o.compositeMaxBounds = np.array([2, 2, 2], np.float32)
    sc = Scene(); sc.addToScene("soup", o)
    c = Cube(); c.setPosition((0.0, -2.2, 0.0)); c.setMaterial(Material("c", (0.3, 0.5, 0.7, 1.0), 0.4, 0.0)); sc.addToScene("cube", c)
    cam = Camera((0.3, 0.4, 6.0)); cam.setWidth(80); cam.setHeight(60); cam.calcRays()
    st = RenderSettings(bounceCount=3, sampleCount=2)
    flat = sc.flatten()
    for builder in (0, 1):
        ctx.set_option(capi.OPT_BUILDER, builder)
        try:
            _, rid, _ = compare_modes(ctx, flat, cam, st)
        finally:
            ctx.set_option(capi.OPT_BUILDER, 0)
    assert (rid[..., 0] == 0).any() and (rid[..., 0] == 1).any()
