"""Debug helper: the non-finite-vertex scene of tests/test_parity_gpu.py, per mode and builder: how many primary ids differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as orc
from raytracer_amd import capi
from raytracer_amd.scene import Camera, Cube, Material, Mesh, MeshObject, RenderSettings, Scene
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
ref, rid, nq = orc.render(flat, cam, st)
ctx = capi.Context(0)
for builder in (0, 1):
    ctx.set_option(capi.OPT_BUILDER, builder)
    for mode in (1, 0, 2):
        ctx.set_option(capi.OPT_TRAVERSAL, mode)
        ctx.upload_scene(flat); ctx.set_camera_rays(cam.preTransformRays)
        fb = ctx.framebuffer(cam.width, cam.height, ids=True)
        ctx.render(cam, st, fb)
        img = ctx.read(fb, cam.width, cam.height); ids = ctx.read_ids(fb, cam.width, cam.height); s = ctx.stats()
        ctx.framebuffer_destroy(fb)
        d = np.any(ids[..., :3] != rid[..., :3], axis=-1)
        print("builder", builder, "mode", mode, "id diffs", int(d.sum()), "checksum diffs", int((ids[..., 3] != rid[..., 3]).sum()), "rays", s.rays, nq)
        ys, xs = np.nonzero(d)
        for y, x in list(zip(ys, xs))[:6]:
            print("   ", x, y, "gpu", ids[y, x, :3], "oracle", rid[y, x, :3])
ctx.close()
