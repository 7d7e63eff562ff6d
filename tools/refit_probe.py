"""Frame time of a refitted tree against the growth of its node area: a scene's objects nudged step by step, re-committed
(RT_OPT_REFIT limit from REFIT=percent; RT_AMD_DEBUG_BUILD=1 prints the area ratio of every refit), a frame timed after each.
   REFIT=400 RT_AMD_DEBUG_BUILD=1 python tools/refit_probe.py config4 0.1     # and REFIT=0 for the rebuilt tree at the same poses"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes
name = sys.argv[1]
scene, cam, st = getattr(scenes, name)()
ctx = capi.Context(0)
if os.environ.get("REFIT"): ctx.set_option(capi.OPT_REFIT, int(os.environ["REFIT"]))   # growth limit in percent, 0 = always rebuild
flat = scene.flatten(); ids = ctx.upload_scene(flat); ctx.set_camera_rays(cam.preTransformRays)
fb = ctx.framebuffer(cam.width, cam.height)
def frame_ms():
    for _ in range(2): ctx.render(cam, st, fb)
    ctx.sync(); ctx.event_record(0)
    for _ in range(4): ctx.render(cam, st, fb)
    ctx.event_record(1); ctx.sync()
    return ctx.event_elapsed_ms(0, 1) / 4
print("built: frame %.2f ms" % frame_ms())
rng = np.random.default_rng(1)
objs = list(scene.getObjects().values())
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
for k in range(8):
    for o in objs[1:]:
        o.setPosition(tuple(np.asarray(o.Position, np.float32) + rng.uniform(-amp, amp, 3).astype(np.float32)))
    fl = scene.flatten()
    ctx.commit(ids["obj"], [np.asarray(o["model"], np.float32) for o in fl.objects], [np.asarray(o["invModel"], np.float32) for o in fl.objects])
    print("step %d kind %d commit %.2f ms frame %.2f ms" % (k, ctx.commit_kind(), ctx.stats().commit_ms, frame_ms()), flush=True)
