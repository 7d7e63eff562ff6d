"""Diagnostic (RT_LOOP_TIMERS build): wave cycles and wave-level iterations of wf_trace's node and
triangle steps over levels 1+ (full depth minus a depth-1 frame, which only runs the packet kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes

def run(name, depth):
    scene, cam, st = getattr(scenes, name)(depth=depth)
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync()
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    ctx.close()
    return dict(cycA=s.idle_node_lanes * 16.0, cycB=s.idle_tri_lanes * 16.0, itA=s.slow_path, itB=s.hits, nodes=s.node_visits, tris=s.tri_tests, rays=s.rays)

for name in sys.argv[1:] or ("config2", "config4", "config5"):
    a, b = run(name, 1), run(name, 4)
    d = {k: b[k] - a[k] for k in a}
    print(name, "levels 1+: rays %.3g | node steps: %.3g wave-iters, %.0f cyc/iter, %.1f lanes/iter, %.1f per ray | tri steps: %.3g wave-iters, %.0f cyc/iter, %.1f lanes/iter, %.1f per ray | wave-cycles node %.3g tri %.3g" % (
        d["rays"], d["itA"], d["cycA"] / max(1, d["itA"]), d["nodes"] / max(1, d["itA"]), d["nodes"] / max(1, d["rays"]),
        d["itB"], d["cycB"] / max(1, d["itB"]), d["tris"] / max(1, d["itB"]), d["tris"] / max(1, d["rays"]), d["cycA"], d["cycB"]))
