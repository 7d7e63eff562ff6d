"""Diagnostic (RT_LOOP_TIMERS build): wave cycles and wave-level iterations of the node and triangle loops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, args in (("config2", ()), ("config2", (1920, 1080, 16, 1)), ("config4", ()), ("config5", (3840, 2160, 1, 4))):
    scene, cam, st = getattr(scenes, name)(*args)
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    cycA, cycB = s.idle_node_lanes * 16.0, s.idle_tri_lanes * 16.0
    itA, itB = max(1, s.slow_path), max(1, s.hits)
    print(name, args, "kernel_ms %.2f rays %.3g | node loop: %.3g wave-iters, %.0f cyc/iter, %.1f lanes/iter | tri loop: %.3g wave-iters, %.0f cyc/iter, %.1f lanes/iter | wave-cycles node %.3g tri %.3g" % (
        s.kernel_ms, s.rays, itA, cycA / itA, s.node_visits / itA, itB, cycB / itB, s.tri_tests / itB, cycA, cycB))
    ctx.close()
