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
        stripes = ((np.arange(size) // max(1, size // 32)) % 2).astype(np.float32)
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
    depth 4.  The aeroplane OBJ is not in the reference checkout (*.obj is git-ignored there), so the
    mesh is the synthetic stand-in above; the albedo map is the reference's own
    res/basiclowpoly/Airplane Texture.png (the bytes its stbi_load returns, tests/golden/ref_assets.npz;
    Texture upload: Material.cu:179-205) at the default size, a procedural map at any other size or
    without the fixture.  A thin ground slab (a zero-thickness quad is invisible in the reference) and
    two props give the bounce rays something to hit."""
    scene = Scene()
    paint = Material("PlanePaint", (0.8, 0.8, 0.8, 1.0), 0.45, 0.1)
    ra = reference_assets()
    if ra is not None and tex_size == 1024 and "airplane_albedo" in ra:
        paint.setAlbedoTexture(Texture(ra["airplane_albedo"]))
    else:
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


def _quat_axis_angle(axis, deg):
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
    h = np.radians(deg) / 2
    return tuple(np.float32(v) for v in (*(a * np.sin(h)), np.cos(h)))


def _look_quat(pitch_deg, yaw_deg=0.0):
    """Camera quaternion: yaw about +y, then pitch about the camera's x axis."""
    qy = np.array(_quat_axis_angle((0, 1, 0), yaw_deg), np.float64)
    qx = np.array(_quat_axis_angle((1, 0, 0), pitch_deg), np.float64)
    x1, y1, z1, w1 = qy; x2, y2, z2, w2 = qx
    q = (w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
         w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2)
    return tuple(np.float32(v) for v in q)


_REF_ASSETS = None


def reference_assets():
    """The reference's own model files flattened by tests/golden/make_assets.py (murasama.fbx, model.dae, four 8-bit
    gray maps), or None when the fixture is absent.  The file travels with the repository, so the GPU box renders
    the real geometry without /root/reference."""
    global _REF_ASSETS
    if _REF_ASSETS is None:
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_assets.npz")
        # RT_NO_REF_ASSETS=1: the synthetic stand-ins of round 1, to compare builds on the same scene
        _REF_ASSETS = dict(np.load(path)) if os.path.exists(path) and not os.environ.get("RT_NO_REF_ASSETS") else False
    return _REF_ASSETS or None


def workload_label(name):
    """bench.py's config.workload string: says whether the reference's geometry or a stand-in is rendered."""
    real = reference_assets() is not None
    return {
        "c2": "C2 low-poly aeroplane (synthetic stand-in mesh: the OBJ is absent from the reference checkout) + "
              + ("the reference's 1024^2 Airplane Texture.png as albedo" if real else "a procedural 1024^2 albedo map"),
        "c3": "C3 pbrsword: " + ("reference geometry (murasama.fbx flattened, 8 972 tris) + its Base_Color / Roughness / Metallic maps, procedural normal map (the reference's is a missing blob)"
                                 if real else "synthetic stand-in mesh and maps") + ", 2048^2 x 4 maps",
        "c4": "C4 331 x lowpolypbrtrash: " + ("reference geometry (model.dae flattened, 6 meshes, 3 024 tris) x 331 = 1 000 944 tris, metalmaterial roughness / metalness maps"
                                                if real else "synthetic stand-in meshes") + ", 2 PBR material sets x 4 maps",
        "c5": "C5 stress: 10 M-triangle terrain + 10 000 tessellated spheres (13.5 M tris)",
    }.get(name, name)


def sword_mesh():
    """The reference's res/pbrsword/source/murasama.fbx as flattened into the fixture (8 972 triangles); without the
    fixture, a stand-in for res/pbrsword/source/murasama.fbx (4 671 control points, ~8 972 triangles after
    fan triangulation; the FBX cannot travel to the GPU box): a katana-like blade, guard and
    wrapped handle with about the same triangle count and extents (x in [-2.5, 2.5])."""
    ra = reference_assets()
    if ra is not None:
        return Mesh(ra["sword_pos"].reshape(-1), ra["sword_idx"].reshape(-1), ra["sword_uv"].reshape(-1), ra["sword_nrm"].reshape(-1))
    parts = [
        ellipsoid_mesh(1.9, 0.17, 0.035, 96, 24, (0.6, 0.0, 0.0)),     # blade   4416
        ellipsoid_mesh(0.06, 0.26, 0.2, 32, 12, (-1.3, 0.0, 0.0)),     # guard    704
        ellipsoid_mesh(0.62, 0.09, 0.07, 64, 16, (-1.9, 0.0, 0.0)),    # handle  1920
        ellipsoid_mesh(0.1, 0.11, 0.09, 32, 16, (-2.45, 0.0, 0.0)),    # pommel   960
        ellipsoid_mesh(0.3, 0.06, 0.05, 32, 16, (-1.75, 0.0, 0.08)),   # ornament 960
    ]
    return merge_meshes(parts)


def config3(width=3840, height=2160, spp=64, depth=4, tex_size=2048):
    """C3 (BASELINE.json configs[2]): the PBR sword with albedo / normal / roughness / metal 2048^2 textures,
    3840x2160, 64 spp.  With tests/golden/ref_assets.npz the geometry is the reference's murasama.fbx and the
    base colour / roughness / metallic maps are its own (the 16-bit base colour as stb_image reduces it, >> 8);
    the normal map (missing from the reference checkout) is procedural.  Without the fixture everything is a stand-in."""
    width, height, spp, depth = width or 3840, height or 2160, spp or 64, depth or 4
    scene = Scene()
    steel = Material("Steel", (0.8, 0.8, 0.85, 1.0), 0.3, 1.0)
    ra = reference_assets()
    if ra is not None and tex_size == 2048 and "sword_base_color" in ra:
        steel.setAlbedoTexture(Texture(ra["sword_base_color"]))
    else:
        steel.setAlbedoTexture(Texture(procedural_texture(tex_size, 11, "albedo")))
    steel.setNormalTexture(Texture(procedural_texture(tex_size, 12, "normal")))
    if ra is not None and tex_size == 2048:
        steel.setRoughnessTexture(Texture(ra["sword_roughness"]))
        steel.setMetalTexture(Texture(ra["sword_metallic"]))
    else:
        steel.setRoughnessTexture(Texture(procedural_texture(tex_size, 13, "scalar")[..., 0]))
        steel.setMetalTexture(Texture(procedural_texture(tex_size, 14, "scalar")[..., 0]))
    stone = Material("Stone", (0.5, 0.48, 0.45, 1.0), 0.8, 0.0)
    if ra is not None:
        # the FBX's blade is broad in its z: stood on edge (z -> y) and tilted, broadside to a close camera.  A 13:1 object
        # cannot fill 30 % of a 16:9 frame (SURVEY 8d asks for it): sword 6 %, sword + plinth 47 % of the primary hits
        rot = tuple(np.float32(v) for v in _quat_mul(_quat_axis_angle((0, 0, 1), 20.0), _quat_axis_angle((1, 0, 0), -90.0)))
        sword = MeshObject([(sword_mesh(), 0)], [steel], position=(0.0, 0.9, 0.0), rotation=rot)
        cam = Camera((0.0, 1.3, 3.3), _look_quat(-12.0), 45.0, 1.0)
    else:
        sword = MeshObject([(sword_mesh(), 0)], [steel], position=(0.0, 0.45, 0.0), rotation=_quat_axis_angle((0, 0, 1), 12.0))
        cam = Camera((0.0, 1.4, 5.2), _look_quat(-12.0), 45.0, 1.0)
    plinth = MeshObject([(merge_meshes([box_mesh((-3.2, -0.3, -1.2), (3.2, 0.0, 1.2))]), 0)], [stone])
    scene.addToScene("sword", sword)
    scene.addToScene("plinth", plinth)
    cam.setWidth(width); cam.setHeight(height); cam.calcRays()
    return scene, cam, RenderSettings(bounceCount=depth, sampleCount=spp)


def trashcan_meshes():
    """res/lowpolypbrtrash/source/model.dae as flattened into the fixture: 6 meshes, 640 + 896 + 4 x 372 = 3 024
    triangles, Z up, node transforms ignored as in the reference (CustomModel.cpp:135-145); returns (meshes,
    material slot per mesh, up-axis fix).  Without the fixture: stand-ins with the same counts, Y up."""
    ra = reference_assets()
    if ra is not None:
        meshes = [Mesh(ra[f"trash{k}_pos"].reshape(-1), ra[f"trash{k}_idx"].reshape(-1), ra[f"trash{k}_uv"].reshape(-1),
                       ra[f"trash{k}_nrm"].reshape(-1)) for k in range(6)]
        return meshes, [int(v) for v in ra["trash_material"]], (-0.70710678, 0.0, 0.0, 0.70710678)
    body = merge_meshes([ellipsoid_mesh(0.45, 0.6, 0.45, 32, 11, (0, 0.6, 0))])        # 640
    lid = merge_meshes([ellipsoid_mesh(0.5, 0.12, 0.5, 32, 15, (0, 1.25, 0))])          # 896
    wheels = [merge_meshes([ellipsoid_mesh(0.12, 0.12, 0.06, 31, 7, (sx * 0.35, 0.12, sz * 0.42))])
              for sx in (-1, 1) for sz in (-1, 1)]                                        # 4 x 372
    return [body, lid] + wheels, [0, 0, 1, 1, 1, 1], (0.0, 0.0, 0.0, 1.0)


def _quat_mul(a, b):
    x1, y1, z1, w1 = a; x2, y2, z2, w2 = b
    return (w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2)


def config4(width=3840, height=2160, spp=4, depth=4, tex_size=2048, instances=331):
    """C4 (BASELINE.json configs[3]): 331 instances of a 3 024-triangle, 6-mesh model = 1 000 944
    triangles on a jittered grid (LCG seed 1), PBR material sets with four textures each, 4K.
    With tests/golden/ref_assets.npz the model is the reference's lowpolypbrtrash model.dae (stood upright:
    it is Z up) and the second material set carries metalmaterial's roughness / metalness maps; albedo and
    normal maps are procedural (the reference's are JPEG / missing).  Without the fixture: stand-in meshes."""
    width, height, spp, depth = width or 3840, height or 2160, spp or 4, depth or 4
    scene = Scene()
    mats = []
    for k, (rough, metal) in enumerate(((0.7, 0.0), (0.25, 1.0))):
        m = Material(f"Set{k}", (0.8, 0.8, 0.8, 1.0), rough, metal)
        m.setAlbedoTexture(Texture(procedural_texture(tex_size, 21 + 10 * k, "albedo")))
        m.setNormalTexture(Texture(procedural_texture(tex_size, 22 + 10 * k, "normal")))
        m.setRoughnessTexture(Texture(procedural_texture(tex_size // 2, 23 + 10 * k, "scalar")[..., 0]))
        m.setMetalTexture(Texture(procedural_texture(tex_size // 2, 24 + 10 * k, "scalar")[..., 0]))
        mats.append(m)
    ground = Material("Floor", (0.4, 0.4, 0.42, 1.0), 0.9, 0.0)
    meshes, slots, upright = trashcan_meshes()
    ra = reference_assets()
    if ra is not None and tex_size == 2048:
        mats[1].setRoughnessTexture(Texture(ra["metal_roughness"]))
        mats[1].setMetalTexture(Texture(ra["metal_metalness"]))
    side = int(np.ceil(np.sqrt(instances)))
    state = 1
    def lcg():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return state / 4294967296.0
    for i in range(instances):
        gx, gz = i % side, i // side
        pos = ((gx - side / 2) * 1.6 + (lcg() - 0.5) * 0.5, 0.0, (gz - side / 2) * 1.6 + (lcg() - 0.5) * 0.5)
        rot = tuple(np.float32(v) for v in _quat_mul(_quat_axis_angle((0, 1, 0), lcg() * 360.0), upright))
        s = (0.85 + 0.3 * lcg()) * (1.5 if ra is not None else 1.0)   # the DAE's pile is ~0.55 x 1.3 x 0.6
        k = i % 2
        o = MeshObject([(m, slots[j] % 2) for j, m in enumerate(meshes)], [mats[k], mats[1 - k]], pos, rot, (s, s, s))
        scene.addToScene(f"can{i}", o)
    ext = side * 0.8 + 2
    scene.addToScene("floor", MeshObject([(merge_meshes([box_mesh((-ext, -0.25, -ext), (ext, 0.0, ext))]), 0)], [ground]))
    cam = Camera((0.0, 9.0, side * 0.8 + 9.0), _look_quat(-28.0), 45.0, 1.0)
    cam.setWidth(width); cam.setHeight(height); cam.calcRays()
    return scene, cam, RenderSettings(bounceCount=depth, sampleCount=spp)


def terrain_mesh(n=2236, size=400.0, seed=2):
    """Height-field grid: 2*n*n triangles (n = 2236 -> 9 999 392)."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-size / 2, size / 2, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(g, g, indexing="xy")
    Y = np.zeros_like(X)
    for k in range(1, 7):
        fx, fz, ph = rng.uniform(0.01, 0.12) * k, rng.uniform(0.01, 0.12) * k, rng.uniform(0, 6.28, 2)
        Y += (3.0 / k) * np.sin(fx * X + ph[0]) * np.cos(fz * Z + ph[1])
    Y = Y.astype(np.float32)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3)
    dy_dx = np.gradient(Y, g, axis=1); dy_dz = np.gradient(Y, g, axis=0)
    nr = np.stack([-dy_dx, np.ones_like(Y), -dy_dz], -1).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=-1, keepdims=True)
    uv = np.stack([(X / size + 0.5) * 8 % 1.0, (Z / size + 0.5) * 8 % 1.0], -1).astype(np.float32).reshape(-1, 2)
    i = (np.arange(n)[:, None] * (n + 1) + np.arange(n)[None, :]).reshape(-1).astype(np.int32)
    idx = np.stack([i, i + n + 2, i + 1, i, i + n + 1, i + n + 2], -1).reshape(-1)
    return Mesh(v.reshape(-1), idx, uv.reshape(-1), nr.reshape(-1, 3).reshape(-1))


def config5(width=7680, height=4320, spp=1, depth=4, spheres=10000, grid=2236):
    """C5 (BASELINE.json configs[4]): ~10 M-triangle terrain + 10 000 tessellated spheres (352 triangles
    each, seed 2), 7680x4320; 256 spp are accumulated as progressive frames split over the GPUs."""
    width, height, spp, depth = width or 7680, height or 4320, spp or 1, depth or 4
    scene = Scene()
    rock = Material("Rock", (0.55, 0.5, 0.45, 1.0), 0.85, 0.0)
    rock.setAlbedoTexture(Texture(procedural_texture(2048, 41, "albedo")))
    shiny = [Material(f"Ball{k}", tuple(c) + (1.0,), r, m) for k, (c, r, m) in enumerate(
        (((0.9, 0.3, 0.3), 0.3, 0.0), ((0.9, 0.9, 0.95), 0.1, 1.0), ((0.3, 0.6, 0.9), 0.6, 0.0), ((1.0, 0.8, 0.4), 0.2, 1.0)))]
    shiny[0].emissionStrength = np.float32(0.0)
    tm = terrain_mesh(grid)
    scene.addToScene("terrain", MeshObject([(tm, 0)], [rock]))
    rng = np.random.default_rng(2)
    Y = tm.vertices.reshape(grid + 1, grid + 1, 3)
    for i in range(spheres):
        gx, gz = rng.integers(0, grid + 1, 2)
        p = Y[gz, gx]
        r = float(rng.uniform(0.4, 1.5))
        s = Sphere(position=(float(p[0]), float(p[1]) + r * 0.45, float(p[2])), scale=(r, r, r))
        s.setMaterial(shiny[i % 4])
        scene.addToScene(f"s{i}", s)
    cam = Camera((0.0, 60.0, 230.0), _look_quat(-18.0), 45.0, 1.0)
    cam.setWidth(width); cam.setHeight(height); cam.calcRays()
    return scene, cam, RenderSettings(bounceCount=depth, sampleCount=spp, maxDistance=1000.0)
