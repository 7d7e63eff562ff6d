"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by raytracer_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrMesh(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("idx", C.c_void_p), ("uv", C.c_void_p), ("nrm", C.c_void_p),
                ("nVerts", C.c_int), ("nTris", C.c_int)]


class OrTexture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("w", C.c_int), ("h", C.c_int)]


class OrMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 4), ("roughness", C.c_float), ("metal", C.c_float),
                ("emissionColor", C.c_float * 3), ("emissionStrength", C.c_float),
                ("texNormal", C.c_int), ("texAlbedo", C.c_int), ("texRoughness", C.c_int), ("texMetal", C.c_int)]


class OrObject(C.Structure):
    _fields_ = [("model", C.c_float * 16), ("invModel", C.c_float * 16),
                ("bmin", C.c_float * 3), ("bmax", C.c_float * 3),
                ("meshCount", C.c_int), ("meshIds", C.c_void_p), ("materialIndices", C.c_void_p),
                ("materialCount", C.c_int), ("materialIds", C.c_void_p)]


class OrScene(C.Structure):
    _fields_ = [("nMeshes", C.c_int), ("meshes", C.c_void_p), ("nTextures", C.c_int), ("textures", C.c_void_p),
                ("nMaterials", C.c_int), ("materials", C.c_void_p), ("nObjects", C.c_int), ("objects", C.c_void_p)]


class OrCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 4), ("rotation", C.c_float * 4), ("exposure", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("rays", C.c_void_p)]


class OrRenderer(C.Structure):
    _fields_ = [("bounceCount", C.c_int), ("maxDistance", C.c_float), ("aoIntensity", C.c_float),
                ("frameCount", C.c_uint32), ("progressiveFrameCount", C.c_uint32),
                ("antiAliasingEnabled", C.c_int), ("sampleCount", C.c_uint32)]


class OrSky(C.Structure):
    _fields_ = [("direction", C.c_float * 4), ("lightColor", C.c_float * 4), ("skyColor", C.c_float * 4)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.or_render.restype = C.c_uint64
        L.or_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int]
        L.or_trace_rays.restype = None
        L.or_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.or_random_value.restype = C.c_float
        L.or_random_value.argtypes = [C.POINTER(C.c_uint32)]
        L.or_dist_to_triangle.restype = C.c_int
        L.or_dist_to_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_float), C.c_void_p]
        L.or_slab.restype = C.c_int
        L.or_slab.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_tex2d.restype = None
        L.or_tex2d.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.or_tex2d_rays.restype = None
        L.or_tex2d_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.or_sky_light.restype = None
        L.or_sky_light.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_exposure.restype = None
        L.or_exposure.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        for n in ("or_logf", "or_cosf", "or_expf", "or_rsqrtf"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.or_powf.restype = C.c_float
        L.or_powf.argtypes = [C.c_float, C.c_float]
        L.or_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data


class OracleScene:
    """Marshals a raytracer_amd.scene.FlatScene into or_scene; keeps the arrays alive."""

    def __init__(self, flat):
        self._keep = []
        k = self._keep
        meshes = (OrMesh * max(1, len(flat.meshes)))()
        for i, m in enumerate(flat.meshes):
            k.extend([m.vertices, m.indices, m.uvCoords, m.normals])
            meshes[i] = OrMesh(_ptr(m.vertices), _ptr(m.indices), _ptr(m.uvCoords), _ptr(m.normals),
                               m.vertices.size // 3, m.triangleCount)
        texs = (OrTexture * max(1, len(flat.textures)))()
        for i, t in enumerate(flat.textures):
            t = np.ascontiguousarray(t, np.uint8)
            k.append(t)
            texs[i] = OrTexture(t.ctypes.data, t.shape[1], t.shape[0])
        mats = (OrMaterial * max(1, len(flat.materials)))()
        for i, m in enumerate(flat.materials):
            mats[i] = OrMaterial((C.c_float * 4)(*m["albedo"]), float(m["roughness"]), float(m["metal"]),
                                 (C.c_float * 3)(*m["emissionColor"]), float(m["emissionStrength"]),
                                 m["texNormal"], m["texAlbedo"], m["texRoughness"], m["texMetal"])
        objs = (OrObject * max(1, len(flat.objects)))()
        for i, o in enumerate(flat.objects):
            mi = np.array(o["meshIds"], np.int32)
            mx = np.array(o["materialIndices"], np.int32)
            md = np.array(o["materialIds"], np.int32)
            k.extend([mi, mx, md])
            objs[i] = OrObject((C.c_float * 16)(*o["model"]), (C.c_float * 16)(*o["invModel"]),
                               (C.c_float * 3)(*o["bmin"]), (C.c_float * 3)(*o["bmax"]),
                               len(mi), mi.ctypes.data, mx.ctypes.data, len(md), md.ctypes.data)
        k.extend([meshes, texs, mats, objs])
        self.c = OrScene(len(flat.meshes), C.addressof(meshes), len(flat.textures), C.addressof(texs),
                         len(flat.materials), C.addressof(mats), len(flat.objects), C.addressof(objs))


def render(flat, camera, settings, rect=None, prev=None, threads=0, want_ids=True, out=None, ids_out=None):
    """Render with the oracle.  camera: raytracer_amd.scene.Camera (rays computed);
    settings: RenderSettings.  Returns (fb[H,W,4] float32, ids[H,W,4] uint32 or None, nQueries).
    out / ids_out: contiguous arrays to render into in place (out also holds the previous frame of a progressive
    sequence): long sequences at large frame sizes then allocate and copy nothing per frame."""
    L = lib()
    sc = flat if isinstance(flat, OracleScene) else OracleScene(flat)
    W, H = camera.width, camera.height
    rays = np.ascontiguousarray(camera.preTransformRays, np.float32)
    cam = OrCamera((C.c_float * 4)(*camera.Position, 1.0), (C.c_float * 4)(*camera.Rotation),
                   float(camera.exposureValue), W, H, rays.ctypes.data)
    rnd = OrRenderer(int(settings.bounceCount), float(settings.maxDistance), float(settings.aoIntensity),
                     int(settings.frameCount) & 0xFFFFFFFF, int(settings.progressiveFrameCount),
                     int(bool(settings.antiAliasingEnabled)), int(settings.sampleCount))
    sky = OrSky((C.c_float * 4)(*settings.lightDirection), (C.c_float * 4)(*settings.lightColor),
                (C.c_float * 4)(*settings.skyColor))
    if out is not None:
        assert out.dtype == np.float32 and out.shape == (H, W, 4) and out.flags.c_contiguous
        fb = out
    else:
        fb = np.zeros((H, W, 4), np.float32) if prev is None else np.ascontiguousarray(prev, np.float32).copy()
    if ids_out is not None:
        assert ids_out.dtype == np.uint32 and ids_out.shape == (H, W, 4) and ids_out.flags.c_contiguous
        ids = ids_out
    else:
        ids = np.full((H, W, 4), 0xFFFFFFFF, np.uint32) if want_ids else None
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
    n = L.or_render(C.addressof(sc.c), C.addressof(cam), C.addressof(rnd), C.addressof(sky),
                    x0, y0, x1, y1, fb.ctypes.data, _ptr(ids), threads)
    return fb, ids, int(n)


def trace_rays(flat, rays, threads=0):
    """rays: [N,8] float32 (ox,oy,oz,maxDistance,dx,dy,dz,pad) -> [N,4] uint32."""
    L = lib()
    sc = flat if isinstance(flat, OracleScene) else OracleScene(flat)
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.empty((rays.shape[0], 4), np.uint32)
    L.or_trace_rays(C.addressof(sc.c), rays.ctypes.data, rays.shape[0], out.ctypes.data, threads)
    return out
