"""Diagnostic (RT_PHASE_TIMERS build): share of wave cycles inside the traversal burst vs outside."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, args in (("config1", (1920, 1080, 16, 4)), ("config2", ()), ("config2", (1920, 1080, 16, 1)), ("config4", ()), ("config5", (3840, 2160, 1, 4))):
    scene, cam, st = getattr(scenes, name)(*args)
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    a, b = s.idle_node_lanes, s.idle_tri_lanes
    print(name, args, "kernel_ms %.2f  traversal %.1f%%  shade+gen+pool %.1f%%  rays %.3g" % (s.kernel_ms, 100.0 * a / max(1, a + b), 100.0 * b / max(1, a + b), s.rays))
    ctx.close()
