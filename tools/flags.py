"""Diagnostic (-DRT_DEBUG_FLAGS build): prints every pixel the wavefront pipeline hands to the redo launch on C2 and why
(stack overflow or a failed object cull of the winner).  C2: 2 samples of one pixel per frame, both cull failures."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
scene, cam, st = scenes.config2()
ctx = capi.Context(0)
ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
fb = ctx.framebuffer(cam.width, cam.height)
ctx.render(cam, st, fb); ctx.sync()
ctx.close()
