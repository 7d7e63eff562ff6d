"""Scenes whose camera rays are aimed, by construction, at the lines that collinear triangles lie on.

The reference's triangle test (GPURaytracer.cu:551-637) decides by the signs of three edge functions in binary32.  When a ray
passes within rounding of the LINE through a collinear triangle, all three are rounding noise and the test reports a hit a
good part of the time -- wherever along that line the triangle sits, so possibly far from the ray, where no walk of a BVH
looks.  rt_scene_commit lists such triangles (rt_commit.hip, thin_kernel) and wf_gen / path_kernel send every camera ray that
comes near a listed line to the exact walks, which test the listed triangles (rt_trace.h, bvh_walk).  Camera rays are the rays
that can be aimed at a line by construction; here that is done on purpose: anti-aliasing off, and for a set of pixels an
object is placed so that the line of its collinear triangles crosses the pixel's ray.  (Measured with the oracle: a third of
the rays that cross such a line inside the span of a collinear triangle "hit" it -- never a triangle whose span does not
hold the crossing point, 0 of 54 000 -- but the distance reported is a mean of the three vertex depths under weights of
rounding noise: anywhere along the sliver, not where the ray meets it, and not inside the part of the ray that the sliver's
bounding box holds.)

Two kinds of object:
 * a needle: nine vertices exactly on one axis and all 84 triangles of three of them, axis-aligned and uniformly scaled (the
   soak found the effect on objects with two axes scaled to zero; their singular matrices switch the hot kernels' pruning off
   altogether, so the render-path test uses well-conditioned instances);
 * the reference's own sword (murasama.fbx, tests/golden/ref_assets.npz), which holds ten collinear triangles, two of them of
   no area at all, under an ordinary rigid transform.
"""
import numpy as np

from raytracer_amd import hostmath, scenes
from raytracer_amd.scene import Camera, Cube, Material, MeshObject, RenderSettings, Scene, Sphere

F = np.float32


def primary_dir(cam, x, y):
    """The camera ray of pixel (x, y) without anti-aliasing, setupRay (GPURaytracer.cu:140-163) in binary32 operation for
    operation: bilinear fetch of the ray table at integer coordinates (the 50/50 blend of texels x-1 and x),
    quaternion rotation (:700-717), normalize with 1/sqrt (:807-811)."""
    T = cam.preTransformRays
    H, W = T.shape[:2]
    x0, x1 = min(max(x - 1, 0), W - 1), min(max(x, 0), W - 1)
    y0, y1 = min(max(y - 1, 0), H - 1), min(max(y, 0), H - 1)
    q = F(0.25)
    v = [((q * T[y0, x0, c] + q * T[y0, x1, c]) + q * T[y1, x0, c]) + q * T[y1, x1, c] for c in range(3)]
    qx, qy, qz, qw = (F(c) for c in cam.Rotation)
    t2, t3, t4 = qw * qx, qw * qy, qw * qz
    t5, t6, t7 = -qx * qx, qx * qy, qx * qz
    t8, t9, t10 = -qy * qy, qy * qz, -qz * qz
    two = F(2)
    r = [two * (((t8 + t10) * v[0] + (t6 - t4) * v[1]) + (t3 + t7) * v[2]) + v[0],
         two * (((t4 + t6) * v[0] + (t5 + t10) * v[1]) + (t9 - t2) * v[2]) + v[1],
         two * (((t7 - t3) * v[0] + (t2 + t9) * v[1]) + (t5 + t8) * v[2]) + v[2]]
    inv = F(1) / np.sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2], dtype=np.float32)
    return np.array([r[0] * inv, r[1] * inv, r[2] * inv], np.float32)


def collinear_triangles(mesh, rel=2.0 ** -17):
    """Indices of the triangles of a mesh whose altitude is below rel x its largest coordinate, with the line each lies on
    (end of the longest edge, unit direction, length)."""
    v = mesh.vertices.reshape(-1, 3).astype(np.float64)
    idx = mesh.indices.reshape(-1, 3)
    p = v[idx]
    out = []
    lim = rel * np.abs(v).max()
    for t in range(len(idx)):
        a, b, c = p[t]
        edges = [(np.linalg.norm(b - a), a, b), (np.linalg.norm(c - b), b, c), (np.linalg.norm(a - c), c, a)]
        ln, e0, e1 = max(edges, key=lambda e: e[0])
        if ln == 0.0:
            continue
        if np.linalg.norm(np.cross(b - a, c - a)) / ln < lim:
            out.append((t, e0, (e1 - e0) / ln, ln))
    return out


AXIS_TURNS = [(0, 0, 0, 1), (0.70710678, 0, 0, 0.70710678), (0, 0.70710678, 0, 0.70710678), (0, 0, 0.70710678, 0.70710678), (0, 0, 1, 0), (1, 0, 0, 0)]


NEEDLE_X = np.array([0.0, 0.1, 0.13, 0.3, 0.35, 0.5, 0.62, 0.8, 1.0]) - 0.5


def needle_meshes():
    """(needle, side box): nine vertices on the x axis and every triangle of three of them; a small box beside it."""
    import itertools
    from raytracer_amd.scene import Mesh
    v = np.stack([NEEDLE_X, np.zeros_like(NEEDLE_X), np.zeros_like(NEEDLE_X)], 1).astype(np.float32)
    tris = np.array(list(itertools.combinations(range(len(NEEDLE_X)), 3)), np.int32)
    uv = np.random.default_rng(0).random((len(NEEDLE_X), 2)).astype(np.float32)
    nrm = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(NEEDLE_X), 1))
    return Mesh(v.reshape(-1), tris.reshape(-1), uv.reshape(-1), nrm.reshape(-1)), scenes.merge_meshes([scenes.box_mesh((-0.05, 0.3, -0.05), (0.05, 0.4, 0.05))])


def _rot(q):
    return hostmath.mat4_cast(q)[:3, :3].astype(np.float64).T   # GLM m[col][row] -> ordinary 3x3


def aimed_scene(seed=5, width=96, height=72, swords=True):
    """(scene, camera, settings, aimed): `aimed` maps pixel (x, y) -> name of the object placed for it."""
    rng = np.random.default_rng(seed)
    q = np.array([-0.06, 0.04, 0.01, 1.0], np.float32); q /= np.linalg.norm(q)
    cam = Camera((0.3, 0.4, 9.0), q, 40.0, 1.0)
    cam.setWidth(width); cam.setHeight(height); cam.calcRays()
    eye = cam.Position.astype(np.float64)
    mats = [Material("a", (0.7, 0.3, 0.3, 1), 0.4, 0.2), Material("b", (0.3, 0.7, 0.3, 1), 0.6, 0.0),
            Material("glow", (0.2, 0.2, 0.8, 1), 0.5, 0.0, (1.0, 0.8, 0.2), 1.5)]
    sc = Scene()
    # something ordinary behind it all, so that the aimed rays have a hit the BVH does find
    back = Cube(); back.setPosition((0.0, 0.0, -4.0)); back.setScale((14.0, 10.0, 1.0)); back.setMaterial(mats[0])
    sc.addToScene("back", back)
    ball = Sphere((-1.0, 0.2, 0.5), (0, 0, 0, 1), (2.0, 2.0, 2.0)); ball.setMaterial(mats[1]); sc.addToScene("ball", ball)
    sword = scenes.sword_mesh() if swords and scenes.reference_assets() is not None else None
    thin = collinear_triangles(sword) if sword is not None else []
    needle, side = needle_meshes()
    aimed = {}
    pixels = [(x, y) for y in range(4, height - 4, 5) for x in range(4, width - 4, 6)]
    for k, (x, y) in enumerate(pixels):
        d = primary_dir(cam, x, y).astype(np.float64)
        P = eye + rng.uniform(4.0, 9.0) * d                         # where the line crosses the ray
        quat = rng.normal(size=4); quat /= np.linalg.norm(quat)
        R = _rot(quat)
        name = f"aim{k}"
        if sword is not None and k % 8 == 7:
            t, a, u, ln = thin[int(rng.integers(0, len(thin)))]
            local = a + rng.uniform(0.1, 0.9) * ln * u               # a point of the sliver's own line
            o = MeshObject([(sword, 0)], [mats[k % 3]], tuple(P - R @ local), quat, (1.0, 1.0, 1.0))
        else:
            # A needle: a mesh whose vertices lie exactly on its x axis (84 overlapping collinear triangles; a small box off to
            # the side gives the object the thickness without which the reference's cull never lets it through),
            # under an exact axis-aligned turn and a uniform scale -- an ordinary, well-conditioned instance whose triangles
            # stay exactly collinear in world space.  The ray is aimed at one of the needle's inner vertices, where dozens of
            # the slivers overlap.  The distance the reference reports for a noise hit is a mean of the sliver's three vertex
            # depths under weights of rounding noise, up to 1e-2 before or behind the crossing point and outside the stretch of
            # the ray that the sliver's box holds (a tenth of such rays, measured with the oracle): a walk that has found one of
            # the overlapping slivers first has no reason to open the box of the one the reference reports.
            quat = np.array(AXIS_TURNS[int(rng.integers(0, len(AXIS_TURNS)))], np.float64)
            R = _rot(quat)
            s = float(rng.uniform(0.8, 3.0))
            lam = float(NEEDLE_X[int(rng.integers(1, len(NEEDLE_X) - 1))])   # the crossing point is an inner vertex of the needle
            o = MeshObject([(needle, 0), (side, 0)], [mats[k % 3]], tuple(P - R[:, 0] * (s * lam)), tuple(quat), (s, s, s))
        sc.addToScene(name, o)
        aimed[(x, y)] = name
    st = RenderSettings(bounceCount=3, sampleCount=2, antiAliasingEnabled=False, frameCount=7)
    return sc, cam, st, aimed


def beyond_any_box(sc, cam, st, aimed, pad=1e-3):
    """Aimed pixels whose primary hit, per the oracle, is a triangle of the object placed for them reported at a distance
    more than `pad` outside the stretch of the pixel's ray that the triangle's own bounding box (widened by `pad`) holds: a
    hit that no bounding box stands for."""
    from oracle import oracle_py as orc
    names = list(sc.getObjects().keys())
    eye = cam.Position.astype(np.float64)
    flat = sc.flatten()
    px = list(aimed.keys())
    rays = np.zeros((len(px), 8), np.float32)
    for i, (x, y) in enumerate(px):
        rays[i, :3] = cam.Position; rays[i, 3] = st.maxDistance; rays[i, 4:7] = primary_dir(cam, x, y)
    res = orc.trace_rays(flat, rays)
    out = []
    for i, (x, y) in enumerate(px):
        if res[i, 0] == 0xFFFFFFFF or names[int(res[i, 0])] != aimed[(x, y)]:
            continue
        o = sc.getObjects()[aimed[(x, y)]]
        mesh = o.meshes[int(res[i, 1])][0]
        M = o.getModelMatrix().astype(np.float64)   # [col][row]
        tri = mesh.vertices.reshape(-1, 3).astype(np.float64)[mesh.indices.reshape(-1, 3)[int(res[i, 2])]] @ M[:3, :3] + M[3, :3]
        d = rays[i, 4:7].astype(np.float64)
        lo, hi = tri.min(0) - pad, tri.max(0) + pad
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (lo - eye) / d, (hi - eye) / d
        tn, tf = np.nanmax(np.minimum(t0, t1)), np.nanmin(np.maximum(t0, t1))
        t = float(res[i, 3:4].view(np.float32)[0])
        if not (tn <= tf) or t < tn - pad or t > tf + pad:
            out.append((x, y))
    return out
