"""Share of rays inside the tie band (re-traced by wf_exact) per workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name in ("config2", "config3", "config4", "config5"):
    scene, cam, st = getattr(scenes, name)()
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    print(name, "rays %.3g band rays %d (%.3f%%) kernel_ms %.2f" % (s.rays, s.slow_path, 100.0 * s.slow_path / max(1, s.rays), s.kernel_ms))
    ctx.close()
