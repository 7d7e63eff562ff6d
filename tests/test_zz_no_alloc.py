"""rt_render_noalloc (include/rt_api.h): the variant of rt_render whose wavefront workspace is a static
device array, for callers that cannot allocate.  Checked through the C ABI against rt_render on the same
context: same pixels, hit ids, path checksums and query counts, bit for bit -- on two calls in a row, the
second with a smaller input (the array keeps the first call's contents), then on a frame that needs several
chunks of the array, and its limits are refused with an error.  And the property it exists for: the device's free
memory is the same before and after its calls (rt_render on the same frame, right after, does take memory)."""
import numpy as np
import pytest

from raytracer_amd import capi, scenes
from raytracer_amd.scene import RenderSettings

pytestmark = pytest.mark.gpu


def frame(ctx, cam, st, noalloc, rect=None, prev=None):
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    if prev is not None:
        ctx.write(fb, prev)
    ctx.render(cam, st, fb, rect, noalloc=noalloc)
    img, ids, rays = ctx.read(fb, cam.width, cam.height), ctx.read_ids(fb, cam.width, cam.height), ctx.stats().rays
    ctx.framebuffer_destroy(fb)
    return img, ids, rays


def test_noalloc_matches_rt_render_on_two_calls_in_a_row():
    ctx = capi.Context(0)
    try:
        ctx.set_option(capi.OPT_TRAVERSAL, 0)
        scene, cam, st = scenes.config2(320, 184, 4, 4, 128)
        ctx.upload_scene(scene.flatten())
        ctx.set_camera_rays(cam.preTransformRays)
        st.frameCount = 7
        # No allocation: free device memory before and after the entry point's first calls on this context.  (One tiny call
        # first: the HIP runtime sets up the scratch memory of kernels that spill on their first launch in a process, which is
        # not this library's doing.)  rt_render has not run on this context, so there is no workspace from it to fall back on.
        import torch
        fb0 = ctx.framebuffer(cam.width, cam.height, ids=True)
        ctx.render(cam, st, fb0, (0, 0, 8, 8), noalloc=True)
        ctx.sync()
        free0 = torch.cuda.mem_get_info()[0]
        ctx.render(cam, st, fb0, noalloc=True)
        ctx.render(cam, RenderSettings(bounceCount=9, sampleCount=7, frameCount=3), fb0, noalloc=True)
        ctx.sync()
        assert torch.cuda.mem_get_info()[0] == free0
        # first call compared: the no-alloc entry point before rt_render has ever run on this context
        got1 = frame(ctx, cam, st, True)
        # second call: a smaller input (fewer samples, shallower, a ragged sub-rectangle, progressive blend)
        st2 = RenderSettings(bounceCount=3, sampleCount=2, frameCount=8, progressiveFrameCount=3)
        rect = (21, 10, 203, 131)
        prev = np.random.default_rng(5).random((cam.height, cam.width, 4)).astype(np.float32)
        got2 = frame(ctx, cam, st2, True, rect, prev)
        ctx.render(cam, st, fb0)          # the allocating entry point on the same frame: its workspace shows in the free memory
        ctx.sync()
        assert torch.cuda.mem_get_info()[0] < free0 - (1 << 20)
        ctx.framebuffer_destroy(fb0)
        want1 = frame(ctx, cam, st, False)
        want2 = frame(ctx, cam, st2, False, rect, prev)
        for got, want in ((got1, want1), (got2, want2)):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
        assert (want1[1][..., 0] != 0xFFFFFFFF).mean() > 0.2
        # a frame of several chunks of the static array: 1.4 M samples at depth 4 against ~350 k per chunk
        scene, cam, st = scenes.config2(640, 360, 6, 4, 128)
        ctx.set_camera_rays(cam.preTransformRays)
        a, b = frame(ctx, cam, st, True), frame(ctx, cam, st, False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        # limits
        fb = ctx.framebuffer(cam.width, cam.height)
        with pytest.raises(capi.RtError, match="bounceCount beyond"):
            ctx.render(cam, RenderSettings(bounceCount=17), fb, noalloc=True)
        with pytest.raises(capi.RtError, match="sampleCount beyond"):
            ctx.render(cam, RenderSettings(sampleCount=20000), fb, noalloc=True)
        ctx.render(cam, RenderSettings(bounceCount=16, sampleCount=1), fb, (0, 0, 64, 64), noalloc=True)
        ctx.sync()
    finally:
        ctx.close()
