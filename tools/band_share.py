"""One rank's share of the C4 tile split on one GPU: time of rt_render_bands(16, r, N) for N = 1, 2, 4, 8 against frame / N
(what the strong-scaling curve can be at best, the exchange aside)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes, multigpu
sc, cam, st = scenes.config4()
ctx = capi.Context(0)
ctx.upload_scene(sc.flatten())
ctx.set_camera_rays(cam.preTransformRays)
fb = ctx.framebuffer(cam.width, cam.height)
if os.environ.get('WF_CHUNK'): ctx.set_option(capi.OPT_WF_CHUNK, int(os.environ['WF_CHUNK'])); ctx.set_option(capi.OPT_WF_LANES, int(os.environ.get('WF_LANES', '0')))
base = None
for n in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1, 2, 4, 8)):
    for r in sorted({0, n - 1}):
        for _ in range(3): ctx.render(cam, st, fb, bands=(multigpu.BAND_ROWS, r, n) if n > 1 else None)
        ctx.sync(); ctx.event_record(0)
        for _ in range(10): ctx.render(cam, st, fb, bands=(multigpu.BAND_ROWS, r, n) if n > 1 else None)
        ctx.event_record(1); ctx.sync()
        ms = ctx.event_elapsed_ms(0, 1) / 10
        base = base or ms
        print(f"N {n} rank {r}: {ms:.3f} ms   frame / N = {base / n:.3f} ms   efficiency {base / n / ms:.2f}")
