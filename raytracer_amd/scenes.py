"""Builders for the BASELINE.json configurations (synthetic where the reference checkout lacks
the asset).  Pure host code (numpy)."""
import numpy as np

from .scene import (Camera, Cube, Material, Mesh, MeshObject, RenderSettings, Scene, Sphere, Texture,
                    default_material)


def config1(width=256, height=256, spp=1, depth=4):
    """C1: default Cube (identity) + Sphere at (1.5,0,0), both `Default` material,
    object order [cube, sphere]; camera (0,0,10), identity rotation, FOV 45, exposure 1; sky
    defaults; AA on; frameCount 0."""
    scene = Scene()
    mat = default_material()
    cube = Cube()
    cube.setMaterial(mat)
    sphere = Sphere(position=(1.5, 0.0, 0.0))
    sphere.setMaterial(mat)
    scene.addToScene("cube", cube)
    scene.addToScene("sphere", sphere)
    cam = Camera((0.0, 0.0, 10.0), (0.0, 0.0, 0.0, 0.0), 45.0, 1.0)
    cam.setWidth(width)
    cam.setHeight(height)
    cam.calcRays()
    st = RenderSettings(bounceCount=depth, sampleCount=spp)
    return scene, cam, st


# ---------------------------------------------------------------------------------------------
# procedural building blocks (synthetic stand-ins; the reference checkout lacks the OBJ and every
# normal map, SURVEY.md 2.3, and /root/reference does not exist on the GPU box)

def procedural_texture(size=1024, seed=7, kind="albedo"):
    """Deterministic RGBA8 texture: value-noise colour panels (albedo), a bumpy tangent-space
    normal map (normal), or a smooth scalar map replicated to RGB (scalar)."""
    rng = np.random.default_rng(seed)
    n = 16
    g = rng.random((n + 1, n + 1, 3)).astype(np.float32)
    t = np.linspace(0, n, size, endpoint=False, dtype=np.float32)
    i = np.floor(t).astype(np.int32)
    f = (t - i).astype(np.float32)
    f = f * f * (3 - 2 * f)
    a = g[i][:, i] * (1 - f)[None, :, None] + g[i][:, i + 1] * f[None, :, None]
    b = g[i + 1][:, i] * (1 - f)[None, :, None] + g[i + 1][:, i + 1] * f[None, :, None]
    v = a * (1 - f)[:, None, None] + b * f[:, None, None]
    if kind == "albedo":
        stripes = ((np.arange(size) // (size // 32)) % 2).astype(np.float32)
        rgb = 0.25 + 0.7 * v
        rgb[..., 0] *= 0.8 + 0.2 * stripes[None, :]
    elif kind == "normal":
        h = v[..., 0]
        dx = np.roll(h, -1, 1) - np.roll(h, 1, 1)
        dy = np.roll(h, -1, 0) - np.roll(h, 1, 0)
        nx, ny, nz = -dx * 24, -dy * 24, np.ones_like(h)
        l = np.sqrt(nx * nx + ny * ny + nz * nz)
        rgb = np.stack([nx / l, ny / l, nz / l], -1) * 0.5 + 0.5
    else:
        rgb = np.repeat(v[..., :1], 3, -1)
    out = np.empty((size, size, 4), np.uint8)
    out[..., :3] = np.clip(rgb * 255 + 0.5, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def ellipsoid_mesh(rx, ry, rz, nu=32, nv=16, center=(0, 0, 0)):
    """Closed UV ellipsoid with normals and UVs: 2*nu*(nv-1) triangles."""
    v, n, uv, idx = [], [], [], []
    for j in range(nv + 1):
        th = np.pi * j / nv
        for i in range(nu + 1):
            ph = 2 * np.pi * i / nu
            d = np.array([np.sin(th) * np.cos(ph), -np.cos(th), np.sin(th) * np.sin(ph)])
            v.append(d * (rx, ry, rz) + center)
            nn = d / (rx, ry, rz)
            n.append(nn / np.linalg.norm(nn))
            uv.append((i / nu, j / nv))
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            b = a + nu + 1
            if j > 0:
                idx += [a, a + 1, b + 1]
            if j < nv - 1:
                idx += [a, b + 1, b]
    return (np.array(v, np.float32), np.array(idx, np.int32), np.array(uv, np.float32), np.array(n, np.float32))


def box_mesh(lo, hi):
    """Axis-aligned box, 12 triangles, per-face normals and UVs."""
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = cube_like = []
    faces = [((0, 0, -1), (1, 0, 0), (0, 1, 0)), ((0, 0, 1), (-1, 0, 0), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, -1), (0, 1, 0)), ((1, 0, 0), (0, 0, 1), (0, 1, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 1, 0), (1, 0, 0), (0, 0, -1))]
    v, n, uv, idx = [], [], [], []
    ctr, half = (lo + hi) / 2, (hi - lo) / 2
    for nrm, ta, tb in faces:
        nrm, ta, tb = (np.array(x, np.float32) for x in (nrm, ta, tb))
        base = len(v)
        for (sa, sb), t in zip(((-1, -1), (1, -1), (1, 1), (-1, 1)), ((0, 0), (1, 0), (1, 1), (0, 1))):
            v.append(ctr + half * (nrm + sa * ta + sb * tb))
            n.append(nrm)
            uv.append(t)
        idx += [base, base + 2, base + 1, base, base + 3, base + 2]
    return (np.array(v, np.float32), np.array(idx, np.int32), np.array(uv, np.float32), np.array(n, np.float32))


def merge_meshes(parts):
    vs, ids, uvs, ns, off = [], [], [], [], 0
    for v, i, uv, n in parts:
        vs.append(v); ids.append(i + off); uvs.append(uv); ns.append(n)
        off += len(v)
    return Mesh(np.concatenate(vs).reshape(-1), np.concatenate(ids), np.concatenate(uvs).reshape(-1),
                np.concatenate(ns).reshape(-1))


def lowpoly_plane_mesh():
    """Stand-in for res/basiclowpoly's Airplane.obj (absent from the reference checkout): a
    low-poly aeroplane of ~2.9k triangles with normals and a UV atlas."""
    parts = [
        ellipsoid_mesh(2.6, 0.42, 0.42, 40, 20),                       # fuselage
        ellipsoid_mesh(0.9, 0.07, 2.9, 24, 12, (0.2, -0.05, 0.0)),     # main wing
        ellipsoid_mesh(0.45, 0.05, 1.1, 16, 8, (-2.2, 0.1, 0.0)),      # tail plane
        ellipsoid_mesh(0.5, 0.6, 0.05, 16, 8, (-2.2, 0.55, 0.0)),      # fin
        ellipsoid_mesh(0.45, 0.22, 0.22, 16, 8, (0.5, -0.3, 1.3)),     # engines
        ellipsoid_mesh(0.45, 0.22, 0.22, 16, 8, (0.5, -0.3, -1.3)),
    ]
    return merge_meshes(parts)


def config2(width=1920, height=1080, spp=16, depth=4, tex_size=1024):
    """C2 (BASELINE.json configs[1]): low-poly model + 1024^2 albedo texture, 1080p, 16 spp,
    depth 4.  The aeroplane OBJ is not in the reference checkout, so the mesh is the synthetic
    stand-in above and the texture is procedural (same size/format as `Airplane Texture.png`).
    A thin ground slab (a zero-thickness quad is invisible in the reference) and
    two props give the bounce rays something to hit."""
    scene = Scene()
    paint = Material("PlanePaint", (0.8, 0.8, 0.8, 1.0), 0.45, 0.1)
    paint.setAlbedoTexture(Texture(procedural_texture(tex_size, 7, "albedo")))
    ground = Material("Ground", (0.45, 0.5, 0.4, 1.0), 0.9, 0.0)
    lamp = Material("Lamp", (1.0, 0.9, 0.7, 1.0), 0.5, 0.0, (1.0, 0.85, 0.6), 4.0)
    chrome = Material("Chrome", (0.9, 0.9, 0.95, 1.0), 0.15, 1.0)

    plane = MeshObject([(lowpoly_plane_mesh(), 0)], [paint], position=(0.0, 0.6, 0.0),
                       rotation=(0.0, 0.38268343, 0.0, 0.92387953))
    slab = MeshObject([(merge_meshes([box_mesh((-12, -0.2, -12), (12, 0.0, 12))]), 0)], [ground])
    bulb = Sphere(position=(2.5, 2.6, 1.5), scale=(1.2, 1.2, 1.2)); bulb.setMaterial(lamp)
    ball = Sphere(position=(-2.8, 0.6, 2.2), scale=(1.2, 1.2, 1.2)); ball.setMaterial(chrome)
    crate = Cube(); crate.setPosition((2.6, 0.5, -2.4)); crate.setMaterial(ground)
    for name, o in (("plane", plane), ("ground", slab), ("bulb", bulb), ("ball", ball), ("crate", crate)):
        scene.addToScene(name, o)
    # camera: pitched down ~17 degrees about x
    cam = Camera((0.0, 3.2, 9.0), (-0.14780941, 0.0, 0.0, 0.98901586), 45.0, 1.0)
    cam.setWidth(width); cam.setHeight(height); cam.calcRays()
    st = RenderSettings(bounceCount=depth, sampleCount=spp)
    return scene, cam, st
