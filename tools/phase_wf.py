"""Diagnostic (RT_PHASE_TIMERS build, wavefront pipeline): share of wf_trace wave time per phase.
Only wf_trace fills these slots in that build when the pipeline is the wavefront one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, args in (("config2", (1920, 1080, 16, 1)), ("config4", (3840, 2160, 4, 1)), ("config5", (7680, 4320, 1, 1))):
    scene, cam, st = getattr(scenes, name)(*args)
    ctx = capi.Context(0)
    ctx.set_option(capi.OPT_WF_LANES, 1)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    burst, write, refill, lanes, iters = s.idle_node_lanes, s.idle_tri_lanes, s.tri_tests, s.hits, max(1, s.slow_path)
    tot = max(1, burst + write + refill)
    print(name, args, "kernel_ms %.2f | wf_trace: burst %.1f%% write %.1f%% refill %.1f%% | main-loop iterations %.3g, ticks/iter burst %.0f write %.0f refill %.0f, diagB %d" % (
        s.kernel_ms, 100.0 * burst / tot, 100.0 * write / tot, 100.0 * refill / tot, iters, 64.0 * burst / iters, 64.0 * write / iters, 64.0 * refill / iters, lanes))
    ctx.close()
