"""C3 (the reference's sword, which holds triangles of no area) at reduced size: the BVH pipeline against the brute-force
mode (the reference's own loop) -- how many pixels differ.  RT_AMD_LIB selects the library build."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes
sc, cam, st = scenes.config3(width=int(sys.argv[1]), height=int(sys.argv[2]), spp=int(sys.argv[3]))
ctx = capi.Context(0)
ctx.upload_scene(sc.flatten())
ctx.set_camera_rays(cam.preTransformRays)
out = {}
for mode in (1, 0):
    ctx.set_option(capi.OPT_TRAVERSAL, mode)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    ctx.render(cam, st, fb)
    out[mode] = (ctx.read(fb, cam.width, cam.height), ctx.read_ids(fb, cam.width, cam.height), ctx.stats().rays)
    ctx.framebuffer_destroy(fb)
d = (out[0][0] != out[1][0]).any(-1)
print("pixels", d.size, "differ in radiance", int(d.sum()), "in ids/checksum", int((out[0][1] != out[1][1]).any(-1).sum()), "rays", out[0][2], out[1][2])
