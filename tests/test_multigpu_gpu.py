"""The N > 1 exchange steps of bench.py on the real backend: a one-rank RCCL group makes the same calls on the same streams
as N ranks do (TileCompositor's pack / all_gather_into_tensor / unpack on its side stream, FrameAccumulator's asynchronous
reduce), so a one-GPU box can check that frames rendered through them equal frames rendered directly, bit for bit, with
two frames in flight.  The band arithmetic for world > 1 is covered on CPU (tests/test_host.py, gloo, world_size 2) and
by rt_render_bands' own test (tests/test_api_gpu.py)."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import capi, multigpu
from raytracer_amd.scene import RenderSettings
from test_api_gpu import small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield torch, dist
    dist.destroy_process_group()


def test_exchange_steps_on_rccl_keep_the_frames(ctx, rccl):
    torch, dist = rccl
    dev = torch.device("cuda", 0)
    sc, cam = small_scene()
    cam.setWidth(150); cam.setHeight(117); cam.calcRays()     # 8 bands of 16 rows, the last one ragged
    W, H = cam.width, cam.height
    st = RenderSettings(bounceCount=3, sampleCount=2, frameCount=0, progressiveFrameCount=0)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(sc.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(W, H)
    want = []
    for f in range(5):
        st.frameCount = f
        ctx.render(cam, st, fb)
        want.append(ctx.read(fb, W, H))
    ctx.framebuffer_destroy(fb)

    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    try:
        # tiles: two frames in flight, composite on the side stream
        tc = multigpu.TileCompositor(torch, dist, W, H, 0, 1, dev, buffers=2, exchange=True)
        assert tc.side is not None
        fbs = [ctx.framebuffer_wrap(f.data_ptr(), W, H) for f in tc.frames]
        got = {}
        for f in range(5):
            st.frameCount = f
            b = tc.acquire(f)
            if f >= 2:      # the buffer's previous frame is complete once acquire returns (in stream order)
                got[f - 2] = tc.frames[b].clone()
            tc.frames[b].zero_()
            ctx.render(cam, st, fbs[b], bands=(multigpu.BAND_ROWS, 0, 1))
            tc.submit(b)
        tc.drain()
        got[3], got[4] = tc.frames[1].clone(), tc.frames[0].clone()
        torch.cuda.synchronize()
        for f in range(5):
            assert np.array_equal(got[f].cpu().numpy(), want[f]), f
        for h in fbs:
            ctx.framebuffer_destroy(h)

        # frames: asynchronous reduce of one buffer beside the render of the other
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        fbs = [ctx.framebuffer_wrap(t.data_ptr(), W, H) for t in bufs]
        fa = multigpu.FrameAccumulator(dist, 1, 0, bufs, acc, exchange=True)
        for f in range(5):
            st.frameCount = f
            b = fa.acquire(f)
            ctx.render(cam, st, fbs[b])
            fa.submit(b)
            assert fa.pending[b] is not None
        fa.drain()
        torch.cuda.synchronize()
        total = np.zeros((H, W, 4), np.float32)
        for f in (0, 1, 2, 4, 3):      # the order the accumulator adds them in: drain() finishes buffer 0 (frame 4) first
            total += want[f]
        assert float(np.abs(acc.cpu().numpy() - total).max()) == 0.0
        for h in fbs:
            ctx.framebuffer_destroy(h)
    finally:
        ctx.sync()
        ctx.set_stream(0)
        torch.cuda.set_stream(torch.cuda.default_stream(dev))


def test_exchange_steps_with_two_frames_in_flight_on_lanes(ctx, rccl):
    """The same exchange steps the way bench.py drives them since round 3: frame i is rendered on workspace lane i % 2
    (RT_OPT_WF_LANE) under torch stream lanes[i % 2], the compositor's / accumulator's calls made under that stream, the
    lanes forked from and joined to the main stream."""
    torch, dist = rccl
    dev = torch.device("cuda", 0)
    sc, cam = small_scene()
    cam.setWidth(150); cam.setHeight(117); cam.calcRays()
    W, H = cam.width, cam.height
    st = RenderSettings(bounceCount=3, sampleCount=2, frameCount=0, progressiveFrameCount=0)
    ctx.set_option(capi.OPT_TRAVERSAL, 0)
    ctx.upload_scene(sc.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(W, H)
    n = 7
    want = []
    for f in range(n):
        st.frameCount = f
        ctx.render(cam, st, fb)
        want.append(ctx.read(fb, W, H))
    ctx.framebuffer_destroy(fb)

    main = torch.cuda.Stream(device=dev)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.set_stream(main)
    ctx.set_stream(main.cuda_stream)

    def on_lane(f):
        ctx.set_option(capi.OPT_WF_LANE, f % 2)
        ctx.set_stream(lanes[f % 2].cuda_stream)
        return torch.cuda.stream(lanes[f % 2])

    def join():
        for s in lanes:
            main.wait_stream(s)
        ctx.set_option(capi.OPT_WF_LANE, 0)
        ctx.set_stream(main.cuda_stream)

    try:
        # tiles
        tc = multigpu.TileCompositor(torch, dist, W, H, 0, 1, dev, buffers=2, exchange=True)
        fbs = [ctx.framebuffer_wrap(f.data_ptr(), W, H) for f in tc.frames]
        got = {}
        for s in lanes:
            s.wait_stream(main)
        for f in range(n):
            st.frameCount = f
            with on_lane(f):
                b = tc.acquire(f)
                if f >= 2:
                    got[f - 2] = tc.frames[b].clone()
                tc.frames[b].zero_()
                ctx.render(cam, st, fbs[b], bands=(multigpu.BAND_ROWS, 0, 1))
                tc.submit(b)
        tc.drain()
        join()
        got[n - 2], got[n - 1] = tc.frames[(n - 2) % 2].clone(), tc.frames[(n - 1) % 2].clone()
        torch.cuda.synchronize()
        for f in range(n):
            assert np.array_equal(got[f].cpu().numpy(), want[f]), f
        for h in fbs:
            ctx.framebuffer_destroy(h)

        # frames
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        fbs = [ctx.framebuffer_wrap(t.data_ptr(), W, H) for t in bufs]
        fa = multigpu.FrameAccumulator(dist, 1, 0, bufs, acc, exchange=True, streams=lanes)
        for s in lanes:
            s.wait_stream(main)
        for f in range(n):
            st.frameCount = f
            with on_lane(f):
                b = fa.acquire(f)
                ctx.render(cam, st, fbs[b])
                fa.submit(b)
        fa.drain()
        join()
        torch.cuda.synchronize()
        even = np.zeros((H, W, 4), np.float32)
        odd = np.zeros((H, W, 4), np.float32)
        for f in range(0, n, 2):
            even += want[f]
        for f in range(1, n, 2):
            odd += want[f]
        assert float(np.abs(acc.cpu().numpy() - (even + odd)).max()) == 0.0   # one partial sum per buffer, folded in buffer order
        for h in fbs:
            ctx.framebuffer_destroy(h)
    finally:
        ctx.sync()
        ctx.set_option(capi.OPT_WF_LANE, 0)
        ctx.set_stream(0)
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
