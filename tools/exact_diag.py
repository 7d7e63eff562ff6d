"""With raytracer_amd/diag_rt_diag_exact.so (make -C raytracer_amd/csrc diag_rt_diag_exact.so; RT_AMD_LIB=...): how many
tie-band rays per frame, and how many of them the two-object table could not settle (reported in rt_stats.hits)."""
import sys
sys.path.insert(0, ".")
from raytracer_amd import capi, scenes
for name in sys.argv[1:] or ["c2", "c4", "c5"]:
    scene, cam, st = getattr(scenes, "config" + name[1:])()
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb)
    s = ctx.stats()
    print(name, "rays %d  band rays %d  table fallbacks %d" % (s.rays, s.slow_path, s.hits))
    ctx.close()
