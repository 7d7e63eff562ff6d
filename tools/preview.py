"""Render a workload on the GPU at reduced size and save a PNG under gpurun_out/ (framing check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from raytracer_amd import capi, scenes
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fn = getattr(scenes, "config" + name[1:])
scene, cam, st = fn(w, h, spp, 4)
ctx = capi.Context(0)
ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
fb = ctx.framebuffer(w, h)
ctx.render(cam, st, fb); img = ctx.read(fb, w, h); s = ctx.stats()
print(name, "tris", s.triangles, "rays", s.rays, "kernel_ms", s.kernel_ms, "build_ms", s.commit_ms, "nodes/ray", s.node_visits / max(1, s.rays), "slow", s.slow_path)
x = np.clip(img[::-1, :, :3], 0, 1)
x = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
Image.fromarray((x * 255 + 0.5).astype(np.uint8)).save(f"gpurun_out/preview_{name}.png")
