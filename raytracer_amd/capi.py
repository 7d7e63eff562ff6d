"""ctypes binding of librt_amd.so (include/rt_api.h) plus `GPURaytracer`, the Python mirror of
the reference's renderer front-end (src/processor/Raytracer.hpp:18-63,
src/processor/GPU/GPURaytracer.cu:24-74).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible, loading
or context creation raises."""
import ctypes as C
import os

import numpy as np

from .scene import Camera, FlatScene, RenderSettings, Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "librt_amd.so"))  # override: timing-only ablation builds
_LIB = None


class RtError(RuntimeError):
    pass


class CameraParams(C.Structure):
    _fields_ = [("origin", C.c_float * 4), ("rotation", C.c_float * 4), ("exposure", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class RendererParams(C.Structure):
    _fields_ = [("bounceCount", C.c_int32), ("maxDistance", C.c_float), ("aoIntensity", C.c_float),
                ("frameCount", C.c_uint32), ("progressiveFrameCount", C.c_uint32),
                ("antiAliasingEnabled", C.c_int32), ("sampleCount", C.c_uint32)]


class SkyParams(C.Structure):
    _fields_ = [("direction", C.c_float * 4), ("lightColor", C.c_float * 4), ("skyColor", C.c_float * 4)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class MaterialDesc(C.Structure):
    _fields_ = [("albedo", C.c_float * 4), ("roughness", C.c_float), ("metal", C.c_float),
                ("emissionColor", C.c_float * 3), ("emissionStrength", C.c_float),
                ("texNormal", C.c_int32), ("texAlbedo", C.c_int32), ("texRoughness", C.c_int32), ("texMetal", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("kernel_ms", C.c_float), ("triangles", C.c_uint32), ("bvh_nodes", C.c_uint32),
                ("commit_ms", C.c_float), ("slow_path", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("hits", C.c_uint64),
                ("packet_node_tests", C.c_uint64), ("packet_node_fetches", C.c_uint64)]

    # diagnostic builds of the library report other tallies in the last two slots (tools/)
    idle_node_lanes = property(lambda self: self.packet_node_tests)
    idle_tri_lanes = property(lambda self: self.packet_node_fetches)


EXPORTS = [
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_set_stream", "rt_set_option", "rt_sync", "rt_get_stats",
    "rt_mesh_create", "rt_mesh_destroy", "rt_texture_create", "rt_texture_destroy", "rt_material_create",
    "rt_material_update", "rt_material_destroy", "rt_object_create", "rt_object_destroy", "rt_scene_commit", "rt_scene_commit_kind",
    "rt_camera_set_rays", "rt_framebuffer_create", "rt_framebuffer_wrap", "rt_framebuffer_destroy",
    "rt_framebuffer_read", "rt_framebuffer_write", "rt_present", "rt_srgb8_thresholds", "rt_framebuffer_enable_ids", "rt_framebuffer_read_ids",
    "rt_image_load", "rt_image_free", "rt_render", "rt_render_bands", "rt_render_noalloc", "rt_trace_rays", "rt_event_record", "rt_event_elapsed_ms", "rt_detmath_eval", "rt_version",
]

OPT_TRAVERSAL, OPT_COUNTERS, OPT_ACCUMULATE_STATS, OPT_WF_CHUNK, OPT_WF_LANES, OPT_BUILDER, OPT_WF_LANE, OPT_REFIT, OPT_THIN_LIST = 1, 2, 3, 4, 5, 6, 7, 8, 9


def lib():
    """Load librt_amd.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_version.restype = C.c_char_p
        L.rt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rt_ctx_destroy.argtypes = [C.c_void_p]
        L.rt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rt_sync.argtypes = [C.c_void_p]
        L.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rt_mesh_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_int32)]
        L.rt_mesh_destroy.argtypes = [C.c_void_p, C.c_int32]
        L.rt_texture_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.rt_texture_destroy.argtypes = [C.c_void_p, C.c_int32]
        L.rt_material_create.argtypes = [C.c_void_p, C.POINTER(MaterialDesc), C.POINTER(C.c_int32)]
        L.rt_material_update.argtypes = [C.c_void_p, C.c_int32, C.POINTER(MaterialDesc)]
        L.rt_material_destroy.argtypes = [C.c_void_p, C.c_int32]
        L.rt_object_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.rt_object_destroy.argtypes = [C.c_void_p, C.c_int32]
        L.rt_scene_commit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.rt_scene_commit_kind.argtypes = [C.c_void_p]
        L.rt_camera_set_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rt_framebuffer_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.rt_framebuffer_wrap.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.rt_framebuffer_destroy.argtypes = [C.c_void_p, C.c_int32]
        L.rt_framebuffer_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rt_framebuffer_write.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rt_framebuffer_enable_ids.argtypes = [C.c_void_p, C.c_int32, C.c_int]
        L.rt_framebuffer_read_ids.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(CameraParams), C.POINTER(RendererParams), C.POINTER(SkyParams),
                                Rect, C.c_int32]
        L.rt_image_load.restype = C.POINTER(C.c_uint8)
        L.rt_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rt_image_free.argtypes = [C.POINTER(C.c_uint8)]
        L.rt_render_noalloc.argtypes = L.rt_render.argtypes
        L.rt_render_bands.argtypes = [C.c_void_p, C.POINTER(CameraParams), C.POINTER(RendererParams), C.POINTER(SkyParams),
                                      Rect, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.rt_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.rt_event_record.argtypes = [C.c_void_p, C.c_int]
        L.rt_event_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.rt_detmath_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_void_p]
        _LIB = L
    return _LIB


def image_load(path):
    """rt_image_load: HxWxC uint8 array of a PNG / baseline JPEG file (the reference's stbi_load bytes), or None."""
    L = lib()
    w, h, n = C.c_int32(), C.c_int32(), C.c_int32()
    p = L.rt_image_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(n))
    if not p:
        return None
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).copy()
    finally:
        L.rt_image_free(p)


def _ptr(a):
    return None if a is None else a.ctypes.data


class Context:
    """One rt_ctx (one GPU)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.rt_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RtError(f"rt_ctx_create({device}) failed ({rc}): {self.L.rt_last_error(None).decode()}")
        self.h = h
        self._scene_handles = None

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise RtError(f"librt_amd error {rc}: {self.L.rt_last_error(self.h).decode()}")

    # ---- scene ----
    def upload_scene(self, flat: FlatScene):
        """Create meshes / textures / materials / objects from a FlatScene and commit them."""
        self.release_scene()
        L, h = self.L, self.h
        ids = dict(mesh=[], tex=[], mat=[], obj=[])
        out = C.c_int32()
        for m in flat.meshes:
            self.check(L.rt_mesh_create(h, _ptr(m.vertices), m.vertices.size // 3, _ptr(m.indices), m.triangleCount,
                                        _ptr(m.uvCoords), _ptr(m.normals), C.byref(out)))
            ids["mesh"].append(out.value)
        for t in flat.textures:
            t = np.ascontiguousarray(t, np.uint8)
            self.check(L.rt_texture_create(h, t.ctypes.data, t.shape[1], t.shape[0], C.byref(out)))
            ids["tex"].append(out.value)

        for m in flat.materials:
            d = self.material_desc(m, ids["tex"])
            self.check(L.rt_material_create(h, C.byref(d), C.byref(out)))
            ids["mat"].append(out.value)
        models, invs = [], []
        for o in flat.objects:
            mi = np.array([ids["mesh"][i] for i in o["meshIds"]], np.int32)
            mx = np.array(o["materialIndices"], np.int32)
            md = np.array([ids["mat"][i] for i in o["materialIds"]], np.int32)
            bmin = np.ascontiguousarray(o["bmin"], np.float32)
            bmax = np.ascontiguousarray(o["bmax"], np.float32)
            self.check(L.rt_object_create(h, _ptr(mi), _ptr(mx), len(mi), _ptr(md), len(md), _ptr(bmin), _ptr(bmax),
                                          C.byref(out)))
            ids["obj"].append(out.value)
            models.append(np.asarray(o["model"], np.float32))
            invs.append(np.asarray(o["invModel"], np.float32))
        self._scene_handles = ids
        self.commit(ids["obj"], models, invs)
        return ids

    @staticmethod
    def material_desc(m: dict, tex_handles) -> "MaterialDesc":
        """rt_material_desc of a FlatScene material; texture indices become handles."""
        def tx(i):
            return -1 if i < 0 else tex_handles[i]
        return MaterialDesc((C.c_float * 4)(*m["albedo"]), float(m["roughness"]), float(m["metal"]),
                            (C.c_float * 3)(*m["emissionColor"]), float(m["emissionStrength"]),
                            tx(m["texNormal"]), tx(m["texAlbedo"]), tx(m["texRoughness"]), tx(m["texMetal"]))

    def update_material(self, handle, m: dict, tex_handles):
        """Material::sendToGPU on a live material (Material.hpp:155-159): patches the committed scene in place."""
        d = self.material_desc(m, tex_handles)
        self.check(self.L.rt_material_update(self.h, handle, C.byref(d)))

    def commit(self, object_ids, models, invs):
        n = len(object_ids)
        oid = np.ascontiguousarray(object_ids, np.int32)
        # (n, 16) arrays are taken as they are: stacking 10 000 small arrays costs more than the library's commit of them
        mm = np.ascontiguousarray(models if isinstance(models, np.ndarray) else (np.stack(models) if n else np.zeros((0, 16))), np.float32).reshape(n, 16)
        im = np.ascontiguousarray(invs if isinstance(invs, np.ndarray) else (np.stack(invs) if n else np.zeros((0, 16))), np.float32).reshape(n, 16)
        self.check(self.L.rt_scene_commit(self.h, _ptr(oid) if n else None, _ptr(mm) if n else None,
                                          _ptr(im) if n else None, n))

    def commit_kind(self):
        """1 = the last commit built the acceleration structure, 2 = it refitted the previous one."""
        return int(self.L.rt_scene_commit_kind(self.h))

    def release_scene(self):
        ids = self._scene_handles
        if not ids:
            return
        L, h = self.L, self.h
        for i in ids["obj"]:
            L.rt_object_destroy(h, i)
        for i in ids["mat"]:
            L.rt_material_destroy(h, i)
        for i in ids["tex"]:
            L.rt_texture_destroy(h, i)
        for i in ids["mesh"]:
            L.rt_mesh_destroy(h, i)
        self._scene_handles = None

    def set_camera_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        self.check(self.L.rt_camera_set_rays(self.h, rays.ctypes.data, rays.shape[1], rays.shape[0]))

    # ---- output ----
    def framebuffer(self, w, h, ids=False):
        out = C.c_int32()
        self.check(self.L.rt_framebuffer_create(self.h, w, h, C.byref(out)))
        if ids:
            self.check(self.L.rt_framebuffer_enable_ids(self.h, out.value, 1))
        return out.value

    def framebuffer_wrap(self, device_ptr, w, h):
        out = C.c_int32()
        self.check(self.L.rt_framebuffer_wrap(self.h, C.c_void_p(device_ptr), w, h, C.byref(out)))
        return out.value

    def framebuffer_destroy(self, fb):
        self.check(self.L.rt_framebuffer_destroy(self.h, fb))

    def read(self, fb, w, h):
        a = np.empty((h, w, 4), np.float32)
        self.check(self.L.rt_framebuffer_read(self.h, fb, a.ctypes.data))
        return a

    def write(self, fb, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        self.check(self.L.rt_framebuffer_write(self.h, fb, arr.ctypes.data))

    def read_ids(self, fb, w, h):
        a = np.empty((h, w, 4), np.uint32)
        self.check(self.L.rt_framebuffer_read_ids(self.h, fb, a.ctypes.data))
        return a

    def render(self, camera: Camera, st: RenderSettings, fb, rect=None, noalloc=False, bands=None):
        """rt_render; `bands` = (bandRows, firstBand, bandStride) renders only those row bands of `rect`
        (rt_render_bands: one rank's share of a frame split over several GPUs)."""
        cam = CameraParams((C.c_float * 4)(*camera.Position, 1.0), (C.c_float * 4)(*camera.Rotation),
                           float(camera.exposureValue), camera.width, camera.height)
        rnd = RendererParams(int(st.bounceCount), float(st.maxDistance), float(st.aoIntensity),
                             int(st.frameCount) & 0xFFFFFFFF, int(st.progressiveFrameCount),
                             int(bool(st.antiAliasingEnabled)), int(st.sampleCount))
        sky = SkyParams((C.c_float * 4)(*st.lightDirection), (C.c_float * 4)(*st.lightColor),
                        (C.c_float * 4)(*st.skyColor))
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, camera.width, camera.height)
        if bands is not None:
            self.check(self.L.rt_render_bands(self.h, C.byref(cam), C.byref(rnd), C.byref(sky), Rect(x0, y0, x1, y1),
                                              int(bands[0]), int(bands[1]), int(bands[2]), fb))
            return
        fn = self.L.rt_render_noalloc if noalloc else self.L.rt_render
        self.check(fn(self.h, C.byref(cam), C.byref(rnd), C.byref(sky), Rect(x0, y0, x1, y1), fb))

    def sync(self):
        self.check(self.L.rt_sync(self.h))

    def stats(self) -> Stats:
        s = Stats()
        self.check(self.L.rt_get_stats(self.h, C.byref(s)))
        return s

    def set_option(self, opt, val):
        self.check(self.L.rt_set_option(self.h, opt, val))

    def set_stream(self, stream_ptr):
        self.check(self.L.rt_set_stream(self.h, C.c_void_p(stream_ptr)))

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.empty((rays.shape[0], 4), np.uint32)
        self.check(self.L.rt_trace_rays(self.h, rays.ctypes.data, rays.shape[0], out.ctypes.data))
        return out

    def event_record(self, slot):
        self.check(self.L.rt_event_record(self.h, slot))

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self.check(self.L.rt_event_elapsed_ms(self.h, a, b, C.byref(ms)))
        return ms.value

    def detmath(self, op, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.check(self.L.rt_detmath_eval(self.h, op, x.ctypes.data, x.size, out.ctypes.data))
        return out


class GPURaytracer:
    """Python mirror of Raytracer + GPURaytracer: same setters, same frame-counter rule."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.st = RenderSettings()
        self._scene = None
        self._key = None
        self._mat_keys = []
        self._keep = None
        self._ids = None         # the handles of this raytracer's own upload
        self._xform_key = None   # the matrices of the last commit

    def setBounceCount(self, n): self.st.bounceCount = int(n)
    def setMaxDistance(self, d): self.st.maxDistance = float(d)
    def setAOIntensity(self, a): self.st.aoIntensity = float(a)
    def setProgressiveRendering(self, p): self.st.progressiveRendering = bool(p)
    def setSampleCount(self, n): self.st.sampleCount = int(n)
    def setAntiAliasingEnabled(self, e): self.st.antiAliasingEnabled = bool(e)
    def getBounceCount(self): return self.st.bounceCount
    def getMaxDistance(self): return self.st.maxDistance
    def getAOIntensity(self): return self.st.aoIntensity
    def getProgressiveRendering(self): return self.st.progressiveRendering
    def getSampleCount(self): return self.st.sampleCount
    def getAntiAliasingEnabled(self): return self.st.antiAliasingEnabled

    def setSkyLight(self, pitch, yaw, light, sky):
        from . import hostmath
        self.st.lightColor = np.asarray(light, np.float32)
        self.st.skyColor = np.asarray(sky, np.float32)
        self.st.lightDirection = hostmath.sky_direction(pitch, yaw)

    def raytrace(self, scene: Scene, camera: Camera, fb, rect=None):
        """GPURaytracer::raytrace (GPURaytracer.cu:24-74).  The reference rebuilds its instance table from
        the live objects on every call (Scene.cpp:3-29), so edits to the Scene between two calls show in the
        next frame.  Here the scene is flattened on every call too (host work only); when nothing changed, nothing
        is sent; when only transforms and material values changed, the meshes and textures already on the device
        are kept, changed materials are patched (rt_material_update) and moved objects re-committed on their handles
        (rt_scene_commit, which refits the acceleration structure); otherwise everything is uploaded again."""
        self.commit_scene(scene)
        camera.update()
        self.ctx.set_camera_rays(camera.preTransformRays)
        self.ctx.render(camera, self.st, fb, rect)
        self.ctx.sync()
        self.st.frameCount += 1
        if self.st.progressiveRendering:
            self.st.progressiveFrameCount += 1
        else:
            self.st.progressiveFrameCount = 0

    @staticmethod
    def _structure(flat: FlatScene):
        """What has to be equal for the device meshes / textures / objects to be reusable."""
        return (tuple(id(m) for m in flat.meshes), tuple(id(t) for t in flat.textures), len(flat.materials),
                tuple((tuple(o["meshIds"]), tuple(o["materialIndices"]), tuple(o["materialIds"]),
                       o["bmin"].tobytes(), o["bmax"].tobytes()) for o in flat.objects))

    @staticmethod
    def _material_key(m: dict):
        return (np.asarray(m["albedo"], np.float32).tobytes(), float(m["roughness"]), float(m["metal"]),
                np.asarray(m["emissionColor"], np.float32).tobytes(), float(m["emissionStrength"]),
                m["texNormal"], m["texAlbedo"], m["texRoughness"], m["texMetal"])

    def commit_scene(self, scene: Scene):
        flat = scene.flatten()
        key = self._structure(flat)
        ids = self.ctx._scene_handles
        # the fast paths use the handles of THIS raytracer's own upload: another GPURaytracer, or a direct
        # ctx.upload_scene(), may have replaced the context's scene since
        if self._scene is scene and ids is not None and ids is self._ids and key == self._key and len(ids["obj"]) == len(flat.objects):
            for k, m in enumerate(flat.materials):
                mk = self._material_key(m)
                if mk != self._mat_keys[k]:
                    self.ctx.update_material(ids["mat"][k], m, ids["tex"])
                    self._mat_keys[k] = mk
            models = [np.asarray(o["model"], np.float32) for o in flat.objects]
            xkey = b"".join(m.tobytes() for m in models)
            if xkey != self._xform_key:   # nothing moved: the committed scene stands (scene->dirty())
                self.ctx.commit(ids["obj"], models, [np.asarray(o["invModel"], np.float32) for o in flat.objects])
                self._xform_key = xkey
        else:
            self._ids = self.ctx.upload_scene(flat)
            self._key = key
            self._mat_keys = [self._material_key(m) for m in flat.materials]
            self._xform_key = b"".join(np.asarray(o["model"], np.float32).tobytes() for o in flat.objects)
            self._keep = flat      # the ids in the key stay unique while these arrays are alive
        self._scene = scene
