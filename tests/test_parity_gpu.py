"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bars: primary hit ids and the all-segment checksum bit-exact; linear
radiance max-abs <= 1e-3 and PSNR >= 60 dB (in practice the two agree bit for bit)."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from raytracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 999.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def render_gpu(ctx, flat, cam, st, mode, rect=None, prev=None):
    ctx.set_option(capi.OPT_TRAVERSAL, mode)
    ctx.upload_scene(flat)
    ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height, ids=True)
    if prev is not None:
        ctx.write(fb, prev)
    ctx.render(cam, st, fb, rect)
    img = ctx.read(fb, cam.width, cam.height)
    ids = ctx.read_ids(fb, cam.width, cam.height)
    stats = ctx.stats()
    ctx.framebuffer_destroy(fb)
    return img, ids, stats


def check(ref, rid, nq, img, ids, stats):
    assert np.array_equal(ids[..., :3], rid[..., :3]), "primary hit ids differ"
    assert np.array_equal(ids[..., 3], rid[..., 3]), "path checksum differs"
    assert stats.rays == nq
    err = np.abs(img[..., :3] - ref[..., :3]).max()
    assert err <= 1e-3, f"max abs radiance error {err}"
    assert psnr(img[..., :3], ref[..., :3]) >= 60.0


@pytest.mark.parametrize("mode", [1, 0], ids=["brute", "bvh"])
@pytest.mark.parametrize("spp,depth", [(1, 4), (2, 3), (3, 4), (4, 4), (6, 2), (9, 4)])
def test_config1(ctx, mode, spp, depth):
    scene, cam, st = scenes.config1(256, 256, spp, depth)
    flat = scene.flatten()
    ref, rid, nq = orc.render(flat, cam, st)
    img, ids, stats = render_gpu(ctx, flat, cam, st, mode)
    check(ref, rid, nq, img, ids, stats)
    assert np.array_equal(img, ref), "radiance is expected to match bit for bit"
