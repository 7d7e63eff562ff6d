"""Diagnostic (RT_PHASE_TIMERS build): share of wave time per phase of the path kernel's main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, args in (("config1", (1920, 1080, 16, 4)), ("config2", ()), ("config2", (1920, 1080, 16, 1)), ("config4", ())):
    scene, cam, st = getattr(scenes, name)(*args)
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    burst, shade, pool, gen, iters = s.idle_node_lanes, s.idle_tri_lanes, s.tri_tests, s.hits, max(1, s.slow_path)
    tot = max(1, burst + shade + pool + gen)
    print(name, args, "kernel_ms %.2f rays %.3g | burst %.1f%% shade %.1f%% pool %.1f%% raygen %.1f%% | main-loop iterations/wave %.0f, ticks/iter: burst %.0f shade %.0f pool %.0f gen %.0f, rays per iteration %.1f" % (
        s.kernel_ms, s.rays, 100.0 * burst / tot, 100.0 * shade / tot, 100.0 * pool / tot, 100.0 * gen / tot, iters / 3072.0,
        64.0 * burst / iters, 64.0 * shade / iters, 64.0 * pool / iters, 64.0 * gen / iters, s.rays / iters))
    ctx.close()
