"""TEST INFRASTRUCTURE -- checker for the legacy CPU renderer of the C++ host mirror (rth::CPURaytracer).

A numpy restatement of the reference's CPURaytracer (src/processor/CPU/CPURaytracer.cpp:6-332, .hpp:44) written
independently of the C++ one: single seed walked through the frame in pixel order, diffuse bounces only, nearest-texel
material getters (src/Material.cu:58-110, 253-266), the first mesh of every object (src/DisplayObject.hpp:49-50), the
object pre-cull against the box spanned by the model-transformed min / max corners (:52-56).  Arithmetic is binary32 step by
step (np.float32 scalars / arrays, no fused operations); logf / cosf / expf are the C library's, called through ctypes,
because that is what the C++ mirror calls.  The reference ran MSVC's math library: parity with it is unpinned.
Only tests/ import this module."""
import ctypes as C
import ctypes.util

import numpy as np

F = np.float32
_m = C.CDLL(ctypes.util.find_library("m"))
for _n in ("logf", "cosf", "expf", "sqrtf"):
    getattr(_m, _n).restype = C.c_float
    getattr(_m, _n).argtypes = [C.c_float]
INF = F(np.inf)


def _mul_point(m, v, w):
    """glm mat4 * vec4 (detail/type_mat4x4.inl): (m0*x + m1*y) + (m2*z + m3*w); m is 4x4 with m[col][row]; v is (..., 3)."""
    v = np.asarray(v, F)
    out = np.empty(v.shape, F)
    for r in range(3):
        out[..., r] = (m[0][r] * v[..., 0] + m[1][r] * v[..., 1]) + (m[2][r] * v[..., 2] + m[3][r] * F(w))
    return out


def _normalize(v):
    s = F(1.0) / F(np.sqrt(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    return np.array([v[0] * s, v[1] * s, v[2] * s], F)


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], F)


class Rng:
    def __init__(self, seed=0):
        self.seed = np.uint32(seed)

    def value(self):  # :298-304
        with np.errstate(over="ignore"):
            self.seed = np.uint32(self.seed * np.uint32(747796405) + np.uint32(2891336453))
            r = np.uint32((self.seed >> np.uint32((self.seed >> np.uint32(28)) + np.uint32(4))) ^ self.seed) * np.uint32(277803737)
            r = np.uint32((r >> np.uint32(22)) ^ r)
        return F(float(r) / 4294967295.0)

    def normal(self):  # :306-312
        theta = F(2 * 3.1415926 * float(self.value()))
        r = self.value()
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = F(_m.sqrtf(F(F(-2.0) * F(_m.logf(r)))))
        return F(rho * F(_m.cosf(theta)))

    def hemisphere(self, n):  # :315-332
        x = self.normal(); y = self.normal(); z = self.normal()
        u = _normalize(np.array([x, y, z], F))
        d = F(F(u[0] * n[0] + u[1] * n[1]) + u[2] * n[2])
        return u if d > 0 else -u


def sample_texture(rgba8, x, y):
    """Texture::sampleTexture (Material.cu:253-266) on the 4-channel copy; reads past the array give 0."""
    h, w = rgba8.shape[:2]
    if x < 0 or x > 1 or y < 0 or y > 1:
        return np.zeros(4, F)
    ix, iy = int(F(x) * F(w)), int(F(y) * F(h))
    flat = rgba8.reshape(-1)
    idx = (iy * w + ix) * 4
    return np.array([F(flat[k]) / F(255.0) if k < flat.size else F(0) for k in range(idx, idx + 4)], F)


def _tri_distances(o, d, v0, v1, v2, max_distance):
    """distToTriangle (:211-296) for all triangles of a mesh at once; returns (t or inf, barycentrics)."""
    a, b, c = v0 - o, v1 - o, v2 - o
    ax, ay, az = abs(d[0]), abs(d[1]), abs(d[2])
    kz = (0 if ax > az else 2) if ax > ay else (1 if ay > az else 2)
    kx = 0 if kz == 2 else kz + 1
    ky = 0 if kx == 2 else kx + 1
    dx, dy, dz = d[kx], d[ky], d[kz]
    p0x, p0y, p0z = a[:, kx].copy(), a[:, ky].copy(), a[:, kz].copy()
    p1x, p1y, p1z = b[:, kx].copy(), b[:, ky].copy(), b[:, kz].copy()
    p2x, p2y, p2z = c[:, kx].copy(), c[:, ky].copy(), c[:, kz].copy()
    with np.errstate(all="ignore"):
        sz = F(1.0) / dz
        sx, sy = -dx * sz, -dy * sz
        p0x = p0x + sx * p0z; p0y = p0y + sy * p0z
        p1x = p1x + sx * p1z; p1y = p1y + sy * p1z
        p2x = p2x + sx * p2z; p2y = p2y + sy * p2z
        e0 = p1x * p2y - p1y * p2x
        e1 = p2x * p0y - p2y * p0x
        e2 = p0x * p1y - p0y * p1x
        z = (e0 == 0) | (e1 == 0) | (e2 == 0)
        if z.any():
            D = np.float64
            e0 = np.where(z, (p2y.astype(D) * p1x.astype(D) - p2x.astype(D) * p1y.astype(D)).astype(F), e0)
            e1 = np.where(z, (p0y.astype(D) * p2x.astype(D) - p0x.astype(D) * p2y.astype(D)).astype(F), e1)
            e2 = np.where(z, (p1y.astype(D) * p0x.astype(D) - p1x.astype(D) * p0y.astype(D)).astype(F), e2)
        bad = ((e0 < 0) | (e1 < 0) | (e2 < 0)) & ((e0 > 0) | (e1 > 0) | (e2 > 0))
        det = (e0 + e1) + e2
        bad |= det == 0
        ts = (e0 * (p0z * sz) + e1 * (p1z * sz)) + e2 * (p2z * sz)
        md = F(max_distance) * det
        bad |= (det < 0) & ((ts >= 0) | (ts < md))
        bad |= (det > 0) & ((ts <= 0) | (ts > md))
        inv = F(1.0) / det
        t = np.where(bad, INF, ts * inv).astype(F)
        bc = np.stack([e0 * inv, e1 * inv, e2 * inv], -1).astype(F)
    return t, bc


def render(objects, camera_pos, camera_rot, pre_rays, exposure, bounce_count=3, max_distance=100.0, ao_intensity=0.01, seed=0):
    """objects: list of dicts {model (4x4 [col][row]), vertices (V,3), indices (T,3), uvs (V,2) or None, min, max, material};
    material: dict {albedo(4), roughness, metal, emissionColor(3), emissionStrength, tex: {role: rgba8}}."""
    H, W = pre_rays.shape[:2]
    out = np.ones((H, W, 4), F)
    rng = Rng(seed)
    q = np.asarray(camera_rot[:3], F)
    qw = F(camera_rot[3])
    ao_intensity, exposure = F(ao_intensity), F(exposure)
    world = [(_mul_point(o["model"], o["vertices"][o["indices"][:, 0]], 1.0), _mul_point(o["model"], o["vertices"][o["indices"][:, 1]], 1.0),
              _mul_point(o["model"], o["vertices"][o["indices"][:, 2]], 1.0)) for o in objects]
    boxes = [(_mul_point(o["model"], np.asarray(o["min"], F), 1.0), _mul_point(o["model"], np.asarray(o["max"], F), 1.0)) for o in objects]

    def mat_get(m, role, const, u, v):
        t = m["tex"].get(role)
        return const if t is None else sample_texture(t, u, v)

    for y in range(H):
        for x in range(W):
            v = pre_rays[y, x, :3].astype(F)
            uv_ = _cross(q, v); uuv = _cross(q, uv_)
            d = (v + ((uv_ * qw) + uuv) * F(2.0)).astype(F)
            o = np.asarray(camera_pos, F)
            incoming = np.zeros(3, F); color = np.ones(3, F)
            did_hit = False; hit = None
            for i in range(bounce_count):
                for k, ob in enumerate(objects):
                    tmin, tmax = F(0.0), (hit["distance"] if did_hit else INF)
                    with np.errstate(all="ignore"):
                        for a in range(3):
                            inv = F(1.0) / d[a]
                            sign = np.signbit(d[a])
                            bmin = boxes[k][1][a] if sign else boxes[k][0][a]
                            bmax = boxes[k][0][a] if sign else boxes[k][1][a]
                            dmin, dmax = (bmin - o[a]) * inv, (bmax - o[a]) * inv
                            tmin = dmin if dmin > tmin else tmin
                            tmax = dmax if dmax < tmax else tmax
                    if not (tmin < tmax):
                        continue
                    t, bc = _tri_distances(o, d, *world[k], max_distance)
                    best = hit["distance"] if did_hit else INF
                    tri = -1
                    # the scan keeps the first triangle of the smallest distance below the current best (:127-144, strict '<')
                    tt = np.where(np.isnan(t), INF, t)
                    if tt.size and tt.min() < best:
                        tri = int(np.argmin(tt)); best = tt[tri]
                    if tri < 0:
                        continue
                    b = bc[tri]
                    idx = ob["indices"][tri]
                    v0, v1, v2 = (ob["vertices"][j].astype(F) for j in idx)
                    uvs = ob["uvs"] if ob["uvs"] is not None else np.zeros((len(ob["vertices"]), 2), F)
                    t0, t1, t2 = (uvs[j].astype(F) for j in idx)
                    e1, e2 = v1 - v0, v2 - v0
                    d1, d2 = t1 - t0, t2 - t0
                    with np.errstate(all="ignore"):
                        f = F(1.0) / (d1[0] * d2[1] - d2[0] * d1[1])
                        tg = f * (d2[1] * e1 - d1[1] * e2)
                        bt = f * (-d2[0] * e1 + d1[0] * e2)
                        T = _normalize(_mul_point(ob["model"], tg, 0.0)); B = _normalize(_mul_point(ob["model"], bt, 0.0))
                    N = _cross(T, B)
                    ip = (b[0] * v0 + b[1] * v1) + b[2] * v2
                    hit = dict(pos=_mul_point(ob["model"], ip, 1.0), distance=best, uv=(b[0] * t0 + b[1] * t1) + b[2] * t2,
                               material=ob["material"], T=T, B=B, N=N)
                    did_hit = True
                if not did_hit:
                    break
                m = hit["material"]; u, vv = hit["uv"]
                nl = mat_get(m, "normal", np.array([0, 0, 1, 0], F), u, vv)[:3]
                normal = ((hit["T"] * nl[0] + hit["B"] * nl[1]) + hit["N"] * nl[2]).astype(F)
                al = mat_get(m, "albedo", np.asarray(m["albedo"], F), u, vv)
                if i == 0:
                    ao = mat_get(m, "ao", np.array([1, 1, 1, 1], F), u, vv)[:3]
                    incoming = incoming + (ao * al[:3]) * ao_intensity
                em = mat_get(m, "emission", np.append(np.asarray(m["emissionColor"], F), F(0)), u, vv)[:3]
                incoming = incoming + (color * em) * F(m["emissionStrength"])
                color = color * al[:3]
                if i < bounce_count - 1:
                    o = hit["pos"]
                    d = rng.hemisphere(normal)
                    did_hit = False
            for ch in range(3):
                out[y, x, ch] = F(1.0) - F(_m.expf(F(-incoming[ch] * exposure)))
    return out
