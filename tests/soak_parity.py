"""Random-scene soak of the HIP path against the CPU oracle: every seed draws a scene (object kinds, counts, scales over
several decades, rotations, touching / duplicated / flat objects, materials with any subset of the maps, emitters), a
camera, a frame size, sample count, depth, maxDistance, anti-aliasing, a progressive blend and a sub-rectangle, and the
frame, the primary ids, the path checksum and the ray count must equal the oracle's bit for bit in the traversal modes.

    python tests/soak_parity.py --first 0 --count 200        # on the GPU box; prints one line per seed, exits 1 on a mismatch

tests/test_parity_gpu.py::test_random_scene_soak runs a fixed range of seeds with the suite."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle_py as orc   # noqa: E402
from raytracer_amd import capi, scenes   # noqa: E402
from raytracer_amd.scene import Camera, Cube, Material, MeshObject, RenderSettings, Scene, Sphere, Texture   # noqa: E402


def _quat(rng, mostly_axis=False):
    if mostly_axis and rng.random() < 0.5:   # exact axis-aligned orientations: coplanar faces, axis-parallel edges
        k = rng.integers(0, 4)
        return [(0, 0, 0, 1), (0.70710678, 0, 0, 0.70710678), (0, 0.70710678, 0, 0.70710678), (0, 0, 1, 0)][k]
    q = rng.normal(size=4)
    return tuple(q / np.linalg.norm(q))


SCALE = 1   # --scale: frame extents multiplied (same scenes, frames of several hundred pixels: many pieces per stripe, full batches)


def draw(seed):
    """(scene, camera, settings, rect, prev) of a seed."""
    rng = np.random.default_rng(1000 + seed)
    style = seed % 6 if seed < 100000 else (seed % 8 if seed < 200000 else seed % 10)   # (earlier seeds keep the styles they were first run with)
                              # 0 ordinary, 1 decades of scale, 2 touching grid, 3 duplicates, 4 flat + tiny, 5 dense pile,
                              # 6 everything axis-aligned on a half-unit grid with an axis-aligned camera, 7 ordinary with a collapsed axis,
                              # 8 the reference's own meshes (sword, bin: slivers and triangles of no area) in random poses, 9 far from the origin
    n_obj = int(rng.integers(1, 40 if style != 5 else 90))
    if style == 8:
        n_obj = int(rng.integers(1, 5))
    real = None
    if style == 8 and scenes.reference_assets() is not None:
        bins, slots, _ = scenes.trashcan_meshes()
        real = [[(scenes.sword_mesh(), 0)], [(m, 0) for m in bins]]
    shift = np.zeros(3)
    if style == 9:      # the whole scene, camera included, far out: every coordinate loses low bits
        shift = rng.choice([-1.0, 1.0], 3) * float(rng.choice([300.0, 5000.0, 40000.0]))
    tex = int(rng.choice([32, 33, 64]))
    mats = []
    for k in range(int(rng.integers(1, 6))):
        m = Material(f"m{k}", tuple(rng.random(3) * 0.9 + 0.05) + (1.0,), float(rng.random()), float(rng.random()),
                     tuple(rng.random(3)), float(rng.random() * 3 * (rng.random() < 0.3)))
        if rng.random() < 0.5:
            m.setAlbedoTexture(Texture(scenes.procedural_texture(tex, seed + k, "albedo")))
        if rng.random() < 0.4:
            m.setNormalTexture(Texture(scenes.procedural_texture(tex, seed + 10 + k, "normal")))
        if rng.random() < 0.4:
            m.setRoughnessTexture(Texture(scenes.procedural_texture(32, seed + 20 + k, "scalar")[..., 0]))
        if rng.random() < 0.4:
            m.setMetalTexture(Texture(scenes.procedural_texture(tex, seed + 30 + k, "scalar")[..., :3]))
        mats.append(m)
    ell = scenes.merge_meshes([scenes.ellipsoid_mesh(0.5, 0.35, 0.45, int(rng.integers(4, 14)), int(rng.integers(3, 9)))])
    box = scenes.merge_meshes([scenes.box_mesh((-0.4, -0.3, -0.5), (0.4, 0.3, 0.5))])
    quad = scenes.merge_meshes([scenes.box_mesh((-0.5, 0.0, -0.5), (0.5, 0.0, 0.5))])   # zero thickness: the slab cull rejects it
    sc = Scene()
    extent = float(rng.choice([1.5, 4.0, 12.0]))
    placed = []
    for i in range(n_obj):
        mat = mats[int(rng.integers(0, len(mats)))]
        if style == 1:
            scl = tuple(10.0 ** rng.uniform(-1.5, 1.2, 3))
        elif style == 4 and rng.random() < 0.3:
            scl = tuple(rng.uniform(0.01, 0.05, 3))
        else:
            scl = tuple(rng.uniform(0.4, 2.0, 3)) if rng.random() < 0.7 else (float(rng.uniform(0.4, 2.0)),) * 3
        if style == 2:       # unit cells of a grid: faces of neighbours coincide exactly
            pos = tuple(np.round(rng.uniform(-2, 2, 3)))
            q, scl = (0, 0, 0, 1), (1.0, 1.0, 1.0)
        elif style == 3 and placed and rng.random() < 0.4:   # the same object again, same place (or the same place, other scale)
            pos, q, scl0 = placed[int(rng.integers(0, len(placed)))]
            scl = scl0 if rng.random() < 0.6 else scl
        elif style == 5:
            pos, q = tuple(rng.normal(0, 0.6, 3)), _quat(rng)
        elif style == 6:
            pos = tuple(np.round(rng.uniform(-3, 3, 3) * 2) / 2)
            q = [(0, 0, 0, 1), (0.70710678, 0, 0, 0.70710678), (0, 0.70710678, 0, 0.70710678), (0, 0, 0.70710678, 0.70710678), (0, 0, 1, 0), (1, 0, 0, 0)][int(rng.integers(0, 6))]
            scl = tuple(rng.choice([0.5, 1.0, 1.0, 2.0], 3))
        else:
            pos, q = tuple(rng.uniform(-extent, extent, 3) * (1, 0.5, 1)), _quat(rng, mostly_axis=True)
        if style == 7 and rng.random() < 0.15:     # one axis collapsed: a singular model matrix, a non-finite inverse
            scl = tuple(0.0 if k == int(rng.integers(0, 3)) else v for k, v in enumerate(scl))
        pos = tuple(np.asarray(pos, np.float64) + shift)
        placed.append((pos, q, scl))
        kind = int(rng.integers(0, 6)) if style != 2 else 1
        if real is not None:
            o = MeshObject(real[int(rng.integers(0, 2))], [mat], pos, _quat(rng, mostly_axis=True), tuple(rng.uniform(0.3, 1.5, 3)))
            sc.addToScene(f"o{i}", o)
            continue
        if kind == 0:
            o = Sphere(pos, q, scl); o.setMaterial(mat)
        elif kind == 1:
            o = Cube(); o.setPosition(pos); o.setRotation(q); o.setScale(scl); o.setMaterial(mat)
        elif kind == 2:
            o = MeshObject([(ell, 1), (box, 0)], [mat, mats[0]], pos, q, scl)
        elif kind == 3:
            o = MeshObject([(box, 0)], [mat], pos, q, scl)
        elif kind == 4:
            o = MeshObject([(ell, 0)], [mat], pos, q, scl)
        else:
            o = MeshObject([(quad, 0), (box, 0)] if rng.random() < 0.5 else [(quad, 0)], [mat], pos, q, scl)
        sc.addToScene(f"o{i}", o)
    if rng.random() < 0.5:   # a floor everything stands in
        fl = MeshObject([(scenes.merge_meshes([scenes.box_mesh((-30, -1.0, -30), (30, -0.7, 30))]), 0)], [mats[0]], tuple(shift))
        sc.addToScene("floor", fl)
    # camera: somewhere around, looking at a point near the middle
    eye = rng.uniform(-1, 1, 3); eye = eye / np.linalg.norm(eye) * extent * rng.uniform(0.2, 3.0)
    if style == 5 and rng.random() < 0.5:
        eye = rng.normal(0, 0.2, 3)          # inside the pile
    tgt = rng.normal(0, 0.3 * extent, 3)
    f = tgt - eye; f /= np.linalg.norm(f)
    # quaternion turning -z into f
    z = np.array([0.0, 0.0, -1.0]); c = float(np.dot(z, f))
    if c < -0.9999:
        q = np.array([0.0, 1.0, 0.0, 0.0])
    else:
        ax = np.cross(z, f); q = np.array([ax[0], ax[1], ax[2], 1.0 + c]); q /= np.linalg.norm(q)
    if style == 6:   # on the grid, looking along an axis: the centre rays are axis-parallel and meet vertices, edges and faces exactly
        eye = np.round(rng.uniform(-4, 4, 3) * 2) / 2
        eye[int(rng.integers(0, 3))] = float(rng.choice([-8.0, 8.0, 6.5]))
        q = np.array([(0, 0, 0, 1), (0, 1, 0, 0), (0, 0.70710678, 0, 0.70710678), (0, -0.70710678, 0, 0.70710678),
                      (0.70710678, 0, 0, 0.70710678), (-0.70710678, 0, 0, 0.70710678)][int(rng.integers(0, 6))])
    eye = eye + shift
    cam = Camera(tuple(eye.astype(np.float32)), tuple(q.astype(np.float32)), float(rng.uniform(25, 100)), float(rng.uniform(0.2, 3.0)))
    W, H = int(rng.integers(9, 140)), int(rng.integers(9, 100))
    if style == 6 and rng.random() < 0.5:
        W, H = W | 1, H | 1          # odd extents: a pixel centre on the optical axis
    W, H = W * SCALE, H * SCALE
    cam.setWidth(W); cam.setHeight(H); cam.calcRays()
    st = RenderSettings(bounceCount=int(rng.choice([0, 1, 2, 3, 4, 4, 5, 7])), sampleCount=int(rng.integers(1, 6)),
                        antiAliasingEnabled=bool(rng.random() < (0.3 if style == 6 else 0.7)), frameCount=int(rng.integers(0, 2 ** 31)),
                        maxDistance=float(rng.choice([100.0, 100.0, 30.0, 6.0, 1.5])), aoIntensity=float(rng.choice([0.01, 0.0, 0.3])))
    prev = None
    if rng.random() < 0.3:
        st.progressiveFrameCount = int(rng.integers(1, 9))
        prev = rng.random((H, W, 4)).astype(np.float32)
    rect = None
    if rng.random() < 0.3:
        x0, y0 = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 4))
        rect = (x0, y0, int(rng.integers(x0 + 1, W + 1)), int(rng.integers(y0 + 1, H + 1)))
    return sc, cam, st, rect, prev


def run_seed(ctx, seed, modes=(0, 2)):
    """None when the seed passes, else a description of the first difference."""
    sc, cam, st, rect, prev = draw(seed)
    flat = sc.flatten()
    ref, rid, nq = orc.render(flat, cam, st, rect=rect, prev=prev)
    W, H = cam.width, cam.height
    x0, y0, x1, y1 = rect or (0, 0, W, H)
    vr = np.random.default_rng(777 + seed)
    for mode in modes:
        ctx.set_option(capi.OPT_TRAVERSAL, mode)
        ctx.upload_scene(flat)
        ctx.set_camera_rays(cam.preTransformRays)
        fb = ctx.framebuffer(W, H, ids=True)
        if prev is not None:
            ctx.write(fb, prev)
        # the wavefront pipeline also through its other entry points and launch shapes, by the seed
        how = ("plain", "noalloc", "bands", "chunks")[(seed // 6) % 4] if mode == 0 else "plain"
        if how == "bands":
            band_rows, world = int(vr.choice([8, 16, 64])), int(vr.integers(2, 6))
            rays = 0
            for r in range(world):
                ctx.render(cam, st, fb, rect, bands=(band_rows, r, world))
                rays += ctx.stats().rays
        else:
            if how == "chunks":
                ctx.set_option(capi.OPT_WF_CHUNK, int(vr.choice([2048, 8192, 65536])))
                ctx.set_option(capi.OPT_WF_LANES, int(vr.integers(0, 5)))
            ctx.render(cam, st, fb, rect, noalloc=(how == "noalloc" and st.bounceCount <= 16))
            rays = ctx.stats().rays
            if how == "chunks":
                ctx.set_option(capi.OPT_WF_CHUNK, 0); ctx.set_option(capi.OPT_WF_LANES, 0)
        img, ids = ctx.read(fb, W, H), ctx.read_ids(fb, W, H)
        ctx.framebuffer_destroy(fb)
        mode = f"{mode} ({how})"
        inside = (slice(y0, y1), slice(x0, x1))
        if prev is not None:   # outside the rectangle the previous frame stays
            keep = np.ones((H, W), bool); keep[inside] = False
            if not np.array_equal(img[keep], prev[keep]):
                return f"mode {mode}: pixels outside the rectangle changed"
        if not np.array_equal(ids[inside], rid[inside]):
            bad = np.argwhere((ids[inside] != rid[inside]).any(-1))
            return f"mode {mode}: ids differ at {len(bad)} pixels, first {tuple(bad[0] + (y0, x0))}"
        if not np.array_equal(img[inside], ref[inside], equal_nan=True):
            d = np.abs(img[inside].astype(np.float64) - ref[inside]); bad = np.argwhere(d.max(-1) > 0)
            return f"mode {mode}: radiance differs at {len(bad)} pixels, max {d.max():.3g}, first {tuple(bad[0] + (y0, x0))}"
        if rays != nq:
            return f"mode {mode}: {rays} rays, oracle {nq}"
    # rt_trace_rays on the same scene: rays between random points, rays aimed exactly at world-space vertices and edge
    # midpoints, axis-parallel rays through them, several maxDistance values
    n = 1500
    verts = []
    for o in flat.objects:
        m = np.asarray(o["model"], np.float64).reshape(4, 4).T
        for mid in o["meshIds"]:
            v = flat.meshes[mid].vertices.reshape(-1, 3).astype(np.float64)
            verts.append((v @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
    verts = np.concatenate(verts)
    a, b = verts[vr.integers(0, len(verts), n)], verts[vr.integers(0, len(verts), n)]
    tgt = np.where(vr.random((n, 1)) < 0.5, a, (0.5 * (a + b)).astype(np.float32))
    centre = np.median(verts, axis=0).astype(np.float64) if seed >= 200000 else np.zeros(3)   # (the far-out scenes of style 9)
    tgt[: n // 3] = (centre + vr.uniform(-4, 4, (n // 3, 3))).astype(np.float32)
    org = (centre + vr.uniform(-9, 9, (n, 3))).astype(np.float32)
    axis = vr.random(n) < 0.25
    k = vr.integers(0, 3, n)
    org[axis] = tgt[axis]
    org[axis, k[axis]] += vr.choice([-7.0, 7.0], int(axis.sum())).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    nz = np.linalg.norm(d, axis=1) > 0
    d[nz] /= np.linalg.norm(d[nz], axis=1, keepdims=True)
    d[~nz] = (0.0, 0.0, 1.0)
    rays_q = np.zeros((n, 8), np.float32)
    rays_q[:, :3] = org; rays_q[:, 3] = vr.choice([100.0, 100.0, 7.5, 3.0], n); rays_q[:, 4:7] = d
    want = orc.trace_rays(flat, rays_q)
    for tm in modes:
        ctx.set_option(capi.OPT_TRAVERSAL, tm)
        ctx.upload_scene(flat)
        got = ctx.trace_rays(rays_q)
        bad = np.nonzero((got != want).any(axis=1))[0]
        if bad.size:
            return f"trace_rays mode {tm}: {bad.size} of {n} rays differ, first {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
    return None


def run_sequence(ctx, seed, steps=7):
    """A live scene edited between frames, rendered through the GPURaytracer mirror (the reference's per-frame use: setters,
    raytrace(), counters that advance): objects moved, rescaled, removed and added, material constants changed, maps bound,
    swapped and cleared, the camera moved, progressive rendering switched on and off.  Every frame must equal the oracle's
    for the scene as it stands.  None when the seed passes."""
    import copy
    rng = np.random.default_rng(5000 + seed)
    sc, cam, st0, _, _ = draw(seed)
    rt = capi.GPURaytracer(ctx)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    rt.setBounceCount(min(st0.bounceCount, 5)); rt.setSampleCount(st0.sampleCount); rt.setMaxDistance(st0.maxDistance)
    rt.setAntiAliasingEnabled(st0.antiAliasingEnabled); rt.setAOIntensity(st0.aoIntensity)
    W, H = cam.width, cam.height
    fb = ctx.framebuffer(W, H, ids=True)
    prev = np.zeros((H, W, 4), np.float32)
    extra = 0
    try:
        for step in range(steps):
            objs = list(sc.getObjects().items())
            for _ in range(int(rng.integers(0, 4))):
                kind = int(rng.integers(0, 9))
                if kind == 0 and objs:
                    objs[int(rng.integers(0, len(objs)))][1].setPosition(tuple(rng.uniform(-3, 3, 3)))
                elif kind == 1 and objs:
                    objs[int(rng.integers(0, len(objs)))][1].setRotation(_quat(rng))
                elif kind == 2 and objs:
                    objs[int(rng.integers(0, len(objs)))][1].setScale(tuple(rng.uniform(0.3, 2.5, 3)))
                elif kind == 3 and len(objs) > 1:
                    sc.removeFromSceen(objs[int(rng.integers(0, len(objs)))][0])
                elif kind == 4:
                    o = Cube() if rng.random() < 0.5 else Sphere()
                    o.setPosition(tuple(rng.uniform(-2, 2, 3))); o.setScale(tuple(rng.uniform(0.3, 1.5, 3)))
                    o.setMaterial(Material(f"x{extra}", tuple(rng.random(3)) + (1.0,), float(rng.random()), float(rng.random())))
                    sc.addToScene(f"extra{extra}", o); extra += 1
                elif kind == 5 and objs:
                    m = objs[int(rng.integers(0, len(objs)))][1].getMaterials()[0]
                    m.simpleAlbedo = np.append(rng.random(3), 1.0).astype(np.float32); m.simpleRoughness = np.float32(rng.random())
                    m.simpleMetal = np.float32(rng.random()); m.emissionStrength = np.float32(rng.random() * 2 * (rng.random() < 0.3))
                elif kind == 6 and objs:
                    m = objs[int(rng.integers(0, len(objs)))][1].getMaterials()[0]
                    t = Texture(scenes.procedural_texture(32, int(rng.integers(0, 1000)), "albedo"))
                    [m.setAlbedoTexture, m.setNormalTexture, m.setMetalTexture][int(rng.integers(0, 3))](t)
                elif kind == 7 and objs:
                    m = objs[int(rng.integers(0, len(objs)))][1].getMaterials()[0]
                    [m.setAlbedoTexture, m.setNormalTexture, m.setRoughnessTexture, m.setMetalTexture][int(rng.integers(0, 4))](None)
                else:
                    cam.setPosition(cam.Position + rng.normal(0, 0.3, 3).astype(np.float32))
                objs = list(sc.getObjects().items())
            if rng.random() < 0.4:
                rt.setProgressiveRendering(not rt.getProgressiveRendering())
            st = copy.deepcopy(rt.st)          # the counters this frame renders with
            flat = sc.flatten()
            ref, rid, nq = orc.render(flat, cam, st, prev=prev if st.progressiveFrameCount > 0 else None)
            rt.raytrace(sc, cam, fb)
            img, ids, rays = ctx.read(fb, W, H), ctx.read_ids(fb, W, H), ctx.stats().rays
            if not np.array_equal(ids, rid):
                return f"step {step}: ids differ"
            if not np.array_equal(img, ref, equal_nan=True):
                return f"step {step}: radiance differs, max {np.abs(img.astype(np.float64) - ref).max():.3g}"
            if rays != nq:
                return f"step {step}: {rays} rays, oracle {nq}"
            prev = img
    finally:
        ctx.framebuffer_destroy(fb)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--brute", action="store_true", help="also the reference-order brute-force mode")
    ap.add_argument("--sequence", action="store_true", help="edited live scenes through the GPURaytracer mirror instead of single frames")
    ap.add_argument("--scale", type=int, default=1, help="multiply the frame extents (the oracle then needs seconds per seed)")
    a = ap.parse_args()
    global SCALE
    SCALE = max(1, a.scale)
    import torch
    torch.cuda.is_available()      # torch's HIP runtime first (tests/conftest.py)
    ctx = capi.Context(0)
    failed = 0
    t0 = time.time()
    for seed in range(a.first, a.first + a.count):
        why = run_sequence(ctx, seed) if a.sequence else run_seed(ctx, seed, (0, 2, 1) if a.brute else (0, 2))
        sc, cam, st, rect, prev = draw(seed)
        print(f"seed {seed:5d} {'ok  ' if why is None else 'FAIL'} {cam.width}x{cam.height} spp {st.sampleCount} "
              f"depth {st.bounceCount} maxd {st.maxDistance:g} {'prog ' if prev is not None else ''}{'rect ' if rect else ''}"
              f"{'' if why is None else why}  [{time.time() - t0:.0f} s]", flush=True)
        failed += why is not None
    ctx.close()
    print(f"{a.count - failed} of {a.count} seeds equal the oracle bit for bit")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
