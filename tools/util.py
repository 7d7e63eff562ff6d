import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi, scenes
for name, args in (("config2", ()), ("config4", ()), ("config5", (3840, 2160, 1, 4))):
    scene, cam, st = getattr(scenes, name)(*args)
    ctx = capi.Context(0)
    ctx.upload_scene(scene.flatten()); ctx.set_camera_rays(cam.preTransformRays)
    fb = ctx.framebuffer(cam.width, cam.height)
    ctx.render(cam, st, fb); ctx.sync(); s = ctx.stats()
    print(name, "node-loop util %.2f  tri-loop util %.2f  nodes %.3g tris %.3g" % (
        s.node_visits / max(1, s.node_visits + s.idle_node_lanes), s.tri_tests / max(1, s.tri_tests + s.idle_tri_lanes), s.node_visits, s.tri_tests))
    ctx.close()
