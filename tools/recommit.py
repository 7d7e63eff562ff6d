"""Wall time of rt_scene_commit when a committed scene's transforms change (dynamic scenes)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, builder in [(n, b) for n in ("config2", "config4", "config5") for b in (0, 1)]:
    scene, cam, st = getattr(scenes, name)()
    flat = scene.flatten()
    ctx = capi.Context(0)
    ctx.set_option(capi.OPT_BUILDER, builder)
    t0 = time.perf_counter(); ids = ctx.upload_scene(flat); ctx.sync(); t_up = time.perf_counter() - t0
    models = [np.asarray(o["model"], np.float32) for o in flat.objects]
    invs = [np.asarray(o["invModel"], np.float32) for o in flat.objects]
    ts = []
    for k in range(5):
        t0 = time.perf_counter(); ctx.commit(ids["obj"], models, invs); ctx.sync(); ts.append(time.perf_counter() - t0)
    print(name, "builder", ("PLOC", "radix")[builder], "tris %d  first upload+commit %.1f ms  re-commit %s ms (GPU part %.2f ms)" % (
        ctx.stats().triangles, t_up * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts), ctx.stats().commit_ms))
    ctx.close()
