"""Diagnostic (RT_PHASE_TIMERS build, wavefront pipeline): share of wave time per phase, for the packet
kernel (level 0: a frame of depth 1) and for wf_trace (levels 1+: full depth minus depth 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes

def run(name, depth):
    scene, cam, st = getattr(scenes, name)(depth=depth)
    ctx = capi.Context(0)
    ctx.set_option(capi.OPT_WF_LANES, 1)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    ctx.close()
    return dict(burst=s.idle_node_lanes, write=s.idle_tri_lanes, refill=s.tri_tests, aux=s.hits, iters=s.slow_path, ms=s.kernel_ms)

def show(tag, d):
    tot = max(1, d["burst"] + d["write"] + d["refill"]); it = max(1, d["iters"])
    print(tag, "walk/burst %.1f%% write %.1f%% refill %.1f%% | iterations %.3g, ticks/iter burst %.0f write %.0f refill %.0f, aux/iter %.1f | wave-seconds at 2.4 GHz %.2f" % (
        100.0 * d["burst"] / tot, 100.0 * d["write"] / tot, 100.0 * d["refill"] / tot, it, 64.0 * d["burst"] / it,
        64.0 * d["write"] / it, 64.0 * d["refill"] / it, d["aux"] / it, 64.0 * tot / 2.4e9))

for name in sys.argv[1:] or ("config2", "config4", "config5"):
    d1, d4 = run(name, 1), run(name, 4)
    show(name + " packet (depth 1, frame %.2f ms):" % d1["ms"], d1)
    show(name + " wf_trace (levels 1-3, frame %.2f ms):" % d4["ms"], {k: d4[k] - d1[k] for k in d1})
