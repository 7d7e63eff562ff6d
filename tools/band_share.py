"""One rank's share of the C4 tile split on one GPU: time of rt_render_bands(16, r, N) for N = 1, 2, 4, 8 against frame / N
(what the strong-scaling curve can be at best, the exchange aside) -- one call after the other on one stream ("serial"), and
with two frames in flight on two lanes and two streams (RT_OPT_WF_LANE, "2 in flight": what bench.py's tiles mode does).
   python tools/band_share.py [N] [workload]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.is_available()
from raytracer_amd import capi, scenes, multigpu
sc, cam, st = getattr(scenes, "config" + (sys.argv[2][1:] if len(sys.argv) > 2 else "4"))()
ctx = capi.Context(0)
ctx.upload_scene(sc.flatten())
ctx.set_camera_rays(cam.preTransformRays)
fbs = [ctx.framebuffer(cam.width, cam.height) for _ in range(4)]
if os.environ.get('WF_CHUNK'): ctx.set_option(capi.OPT_WF_CHUNK, int(os.environ['WF_CHUNK'])); ctx.set_option(capi.OPT_WF_LANES, int(os.environ.get('WF_LANES', '0')))
main = torch.cuda.Stream()
lanes = [torch.cuda.Stream() for _ in range(4)]
REPS = 24


def run(n, r, inflight):
    bands = (multigpu.BAND_ROWS, r, n) if n > 1 else None
    def frame(i):
        if inflight == 1:
            ctx.render(cam, st, fbs[0], bands=bands)
        else:
            ctx.set_option(capi.OPT_WF_LANE, i % inflight); ctx.set_stream(lanes[i % inflight].cuda_stream)
            ctx.render(cam, st, fbs[i % inflight], bands=bands)
    ctx.set_option(capi.OPT_WF_LANE, 0); ctx.set_stream(main.cuda_stream)
    for i in range(4): frame(i)
    torch.cuda.synchronize()
    ctx.set_option(capi.OPT_WF_LANE, 0); ctx.set_stream(main.cuda_stream)
    ctx.event_record(0)
    for s in lanes: s.wait_stream(main)
    for i in range(REPS): frame(i)
    for s in lanes: main.wait_stream(s)
    ctx.set_option(capi.OPT_WF_LANE, 0); ctx.set_stream(main.cuda_stream)
    ctx.event_record(1); ctx.sync()
    return ctx.event_elapsed_ms(0, 1) / REPS


base = run(1, 0, 1)   # the whole frame, one call after the other
for n in ([1, int(sys.argv[1])] if len(sys.argv) > 1 and sys.argv[1] != "all" else (1, 2, 4, 8)):
    for r in sorted({0, n - 1}):
        ms1 = run(n, r, 1)
        more = ""
        if os.environ.get('BS_MODE') != 'serial':   # BS_MODE=serial: tools/band_share_prof.sh reads the last pass
            for k in (2, 3, 4):
                ms = run(n, r, k)
                more += f"   {k} in flight {ms:.3f} ms per frame ({base / n / ms:.2f})"
        print(f"N {n} rank {r}: serial {ms1:.3f} ms (efficiency {base / n / ms1:.2f}){more}   frame / N = {base / n:.3f} ms")
