"""Frame time of rt_render_noalloc (the entry point that allocates nothing: workspace = a 64 MiB static device array, the
frame as consecutive chunks of ~350 000 samples on the context stream) against rt_render, per workload:
   python tools/noalloc_time.py [c2 c4 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes
for name in (sys.argv[1:] or ("c2", "c4")):
    sc, cam, st = getattr(scenes, "config" + name[1:])()
    ctx = capi.Context(0)
    ctx.upload_scene(sc.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    out = []
    for noalloc in (False, True):
        for _ in range(2):
            ctx.render(cam, st, fb, noalloc=noalloc)
        ctx.sync(); ctx.event_record(0)
        for _ in range(4):
            ctx.render(cam, st, fb, noalloc=noalloc)
        ctx.event_record(1); ctx.sync()
        out.append(ctx.event_elapsed_ms(0, 1) / 4)
    samples = cam.width * cam.height * st.sampleCount
    print("%s: rt_render %.2f ms, rt_render_noalloc %.2f ms (%d samples, about %d chunks)" % (name, out[0], out[1], samples, (samples + 349999) // 350000))
    ctx.close()
