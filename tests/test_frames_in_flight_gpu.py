"""Frames in flight (RT_OPT_WF_LANE, include/rt_api.h): consecutive rt_render / rt_render_bands calls on different workspace
lanes and different streams may overlap on the GPU.  Every frame must come out as it does alone -- pixels, primary ids and,
summed, the query counts -- with two and with four frames in flight, for whole frames and for one rank's bands; lanes outside
0..3 are refused; and multigpu.FrameAccumulator(streams=...) sums the frames of two lanes like the frames rendered one by one."""
import numpy as np
import pytest

from raytracer_amd import capi, multigpu, scenes

pytestmark = pytest.mark.gpu


def _ctx_scene(width=448, height=256, spp=3, depth=4):
    ctx = capi.Context(0)
    scene, cam, st = scenes.config2(width, height, spp, depth, 128)
    ctx.upload_scene(scene.flatten())
    ctx.set_camera_rays(cam.preTransformRays)
    return ctx, cam, st


@pytest.mark.parametrize("bands", [None, (16, 1, 3)], ids=["frames", "bands"])
def test_frames_in_flight_equal_frames_alone(bands):
    import torch
    ctx, cam, st = _ctx_scene()
    try:
        W, H, n = cam.width, cam.height, 8
        alone, rays_alone = [], 0
        fb = ctx.framebuffer(W, H, ids=True)
        for i in range(n):
            st.frameCount = i
            ctx.write(fb, np.zeros((H, W, 4), np.float32))
            ctx.render(cam, st, fb, bands=bands)
            alone.append((ctx.read(fb, W, H), ctx.read_ids(fb, W, H)))
            rays_alone += ctx.stats().rays
        ctx.framebuffer_destroy(fb)
        assert (alone[0][1][..., 0] != 0xFFFFFFFF).mean() > 0.02
        for k in (2, 4):
            streams = [torch.cuda.Stream() for _ in range(k)]
            fbs = [ctx.framebuffer(W, H, ids=True) for _ in range(k)]
            for f in fbs:
                ctx.write(f, np.zeros((H, W, 4), np.float32))
            ctx.sync()
            ctx.set_option(capi.OPT_ACCUMULATE_STATS, 1)
            for base in range(0, n, k):
                for j in range(k):       # k frames enqueued back to back, none waited for
                    st.frameCount = base + j
                    ctx.set_option(capi.OPT_WF_LANE, j)
                    ctx.set_stream(streams[j].cuda_stream)
                    ctx.render(cam, st, fbs[j], bands=bands)
                torch.cuda.synchronize()
                for j in range(k):
                    ctx.set_stream(streams[j].cuda_stream)
                    img, ids = ctx.read(fbs[j], W, H), ctx.read_ids(fbs[j], W, H)
                    assert np.array_equal(img, alone[base + j][0]) and np.array_equal(ids, alone[base + j][1]), (k, base + j)
                    if bands is not None:    # the next frame on this lane starts from a clean buffer again
                        ctx.write(fbs[j], np.zeros((H, W, 4), np.float32))
            ctx.set_option(capi.OPT_WF_LANE, 0)
            ctx.set_stream(None)
            torch.cuda.synchronize()
            assert ctx.stats().rays == rays_alone
            ctx.set_option(capi.OPT_ACCUMULATE_STATS, 0)
            for f in fbs:
                ctx.framebuffer_destroy(f)
        for bad in (-1, 4):
            with pytest.raises(capi.RtError):
                ctx.set_option(capi.OPT_WF_LANE, bad)
    finally:
        ctx.close()


def test_frame_accumulator_over_two_lanes():
    import torch
    ctx, cam, st = _ctx_scene(320, 184, 2, 3)
    try:
        W, H, n = cam.width, cam.height, 7
        dev = torch.device("cuda", 0)
        main = torch.cuda.Stream()
        lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        fbs = [ctx.framebuffer_wrap(b.data_ptr(), W, H) for b in bufs]
        want = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        with torch.cuda.stream(main):
            ctx.set_stream(main.cuda_stream)
            for i in range(n):           # one by one: each frame added before the next one is rendered
                st.frameCount = i
                ctx.render(cam, st, fbs[0])
                want.add_(bufs[0])
            main.synchronize()
            acc = torch.zeros_like(want)
            fa = multigpu.FrameAccumulator(None, 1, 0, bufs, acc, streams=lanes)
            for s in lanes:
                s.wait_stream(main)
            for i in range(n):
                lane = i & 1
                ctx.set_option(capi.OPT_WF_LANE, lane)
                ctx.set_stream(lanes[lane].cuda_stream)
                with torch.cuda.stream(lanes[lane]):
                    st.frameCount = i
                    b = fa.acquire(i)
                    assert b == lane
                    ctx.render(cam, st, fbs[b])
                    fa.submit(b)
            fa.drain()
            ctx.set_option(capi.OPT_WF_LANE, 0)
            ctx.set_stream(main.cuda_stream)
            main.synchronize()
        # the partial sums add the frames in another order than the running sum does: equal to rounding, not to the bit
        assert torch.allclose(acc, want, rtol=0, atol=2e-6 * n)
        assert float(want.abs().max()) > 0.1
        ctx.set_stream(None)
    finally:
        ctx.close()
