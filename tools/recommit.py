"""Wall time of rt_scene_commit when a committed scene's transforms change (dynamic scenes): refit of the tree
of the first commit (RT_OPT_REFIT, the default) against a full build, both builders; and what the refitted tree costs a frame."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes, hostmath
for name in (sys.argv[1:] or ("config2", "config4", "config5")):
    scene, cam, st = getattr(scenes, name)()
    for builder, refit in ((0, 1), (0, 0), (1, 0)):
        ctx = capi.Context(0)
        ctx.set_option(capi.OPT_BUILDER, builder)
        ctx.set_option(capi.OPT_REFIT, refit)
        flat = scene.flatten()
        t0 = time.perf_counter(); ids = ctx.upload_scene(flat); ctx.sync(); t_up = time.perf_counter() - t0
        ctx.set_camera_rays(cam.preTransformRays)
        fb = ctx.framebuffer(cam.width, cam.height)
        def frame_ms():
            for _ in range(2): ctx.render(cam, st, fb)
            ctx.sync(); ctx.event_record(0)
            for _ in range(5): ctx.render(cam, st, fb)
            ctx.event_record(1); ctx.sync()
            return ctx.event_elapsed_ms(0, 1) / 5
        f0 = frame_ms()
        ts, kinds = [], []
        rng = np.random.default_rng(1)
        objs = list(scene.getObjects().values())
        for k in range(5):   # every object but the largest nudged by up to 0.3 units, five times over
            for o in objs[1:] if len(objs) > 1 else objs:
                o.setPosition(tuple(np.asarray(o.Position, np.float32) + rng.uniform(-0.3, 0.3, 3).astype(np.float32)))
            fl = scene.flatten()
            models = np.stack([np.asarray(o["model"], np.float32).reshape(16) for o in fl.objects])
            invs = np.stack([np.asarray(o["invModel"], np.float32).reshape(16) for o in fl.objects])
            oid = np.asarray(ids["obj"], np.int32)
            t0 = time.perf_counter(); ctx.commit(oid, models, invs); ctx.sync(); ts.append(time.perf_counter() - t0)   # the C ABI call and its GPU work
            kinds.append(ctx.commit_kind())
        f1 = frame_ms()
        print(name, "builder", ("PLOC", "radix")[builder], "refit" if refit else "rebuild", "tris %d  first upload+commit %.1f ms  re-commit %s ms (GPU part %.2f ms) kinds %s  frame before %.2f ms, after the moves %.2f ms" % (
            ctx.stats().triangles, t_up * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts), ctx.stats().commit_ms, kinds, f0, f1))
        ctx.close()
        scene, cam, st = getattr(scenes, name)()
