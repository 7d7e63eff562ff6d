// rt_trace.h -- closest-hit for one ray (device code shared by the render kernel and
// rt_trace_rays).  Result semantics are the reference's getIntersectionPoint(ray, vector)
// (processor/GPU/GPURaytracer.cu:247-265): objects in array order, object-space AABB cull with
// the current ray.maxDistance (:267-315), brute force over every triangle (:379-428) with the
// watertight test (:551-637), strict '<' at triangle, mesh and object level, and
// ray.maxDistance narrowed by every accepted object hit.
//
// Two implementations return the same bits:
//  * trace_brute: that algorithm as written, over the flattened reference-order records;
//  * trace_bvh:   a BVH over the same world-space triangles finds the lexicographic minimum
//                 (t, gid) -- which is the reference's answer whenever no other object has a
//                 hit within a small band of it -- and falls back to an exact in-order
//                 emulation restricted to the objects inside the band otherwise
//                 (DESIGN.md section 5 gives the argument).
#pragma once
#include "rt_device.h"
#include "rt_detmath.h"
#include <float.h>

namespace rt {

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float maxDist;
};

struct Hit {
    float t;
    float b0, b1, b2;
    uint32_t gid;   // kNoHit on miss
};

// per-ray constants of the watertight test (:563-577), a pure function of the direction
struct TriRay {
    int kz;
    float sx, sy, sz;
};

__device__ __forceinline__ float sel3(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }

__device__ __forceinline__ TriRay make_tri_ray(const Ray& r)
{
    TriRay q;
    const float ax = __builtin_fabsf(r.dx), ay = __builtin_fabsf(r.dy), az = __builtin_fabsf(r.dz);
    q.kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    // kx = kz+1 mod 3, ky = kx+1 mod 3
    const float dX = sel3(q.kz, r.dy, r.dz, r.dx);
    const float dY = sel3(q.kz, r.dz, r.dx, r.dy);
    const float dZ = sel3(q.kz, r.dx, r.dy, r.dz);
    q.sz = 1.f / dZ;
    q.sx = -dX * q.sz;
    q.sy = -dY * q.sz;
    return q;
}

// :551-637.  Returns true and fills t / barycentrics when the triangle is hit within
// (0.001, maxDist]; bit-for-bit the oracle's dist_to_triangle.
__device__ __forceinline__ bool tri_test(const Ray& r, const TriRay& q, float maxDist,
                                         float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                         float v2x, float v2y, float v2z, float& tOut, float& b0Out,
                                         float& b1Out, float& b2Out)
{
    const float a0x = v0x - r.ox, a0y = v0y - r.oy, a0z = v0z - r.oz;
    const float a1x = v1x - r.ox, a1y = v1y - r.oy, a1z = v1z - r.oz;
    const float a2x = v2x - r.ox, a2y = v2y - r.oy, a2z = v2z - r.oz;

    float p0x = sel3(q.kz, a0y, a0z, a0x), p0y = sel3(q.kz, a0z, a0x, a0y), p0z = sel3(q.kz, a0x, a0y, a0z);
    float p1x = sel3(q.kz, a1y, a1z, a1x), p1y = sel3(q.kz, a1z, a1x, a1y), p1z = sel3(q.kz, a1x, a1y, a1z);
    float p2x = sel3(q.kz, a2y, a2z, a2x), p2y = sel3(q.kz, a2z, a2x, a2y), p2z = sel3(q.kz, a2x, a2y, a2z);

    p0x += q.sx * p0z; p0y += q.sy * p0z;
    p1x += q.sx * p1z; p1y += q.sy * p1z;
    p2x += q.sx * p2z; p2y += q.sy * p2z;

    float e0 = p1x * p2y - p1y * p2x;
    float e1 = p2x * p0y - p2y * p0x;
    float e2 = p0x * p1y - p0y * p1x;

    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double a = (double)p2x * (double)p1y, b = (double)p2y * (double)p1x;
        e0 = (float)(b - a);
        a = (double)p0x * (double)p2y; b = (double)p0y * (double)p2x;
        e1 = (float)(b - a);
        a = (double)p1x * (double)p0y; b = (double)p1y * (double)p0x;
        e2 = (float)(b - a);
    }

    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    const float det = e0 + e1 + e2;
    if (det == 0) return false;

    p0z *= q.sz; p1z *= q.sz; p2z *= q.sz;
    const float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0 && (tScaled >= 0 || tScaled < maxDist * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > maxDist * det)) return false;

    const float invDet = 1 / det;
    const float t = tScaled * invDet;
    if ((double)t < 0.001 || t > maxDist) return false;
    if (!(t < FLT_MAX)) return false; // reference: a FLT_MAX / NaN distance never wins '<' (:421)

    tOut = t; b0Out = e0 * invDet; b1Out = e1 * invDet; b2Out = e2 * invDet;
    return true;
}

// :728-743 (transpose == false), columns c0..c3 of a column-major 16-float matrix
__device__ __forceinline__ void mat_vec(const float* __restrict__ m, float x, float y, float z, float w,
                                        float& ox, float& oy, float& oz)
{
    ox = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    oy = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    oz = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
}

// :267-315: object-space slab test.  Split in two so that the band logic can reuse it:
// object_slab computes (tmin, boxFar) with tmax = +inf; the reference's answer for a given
// ray.maxDistance m is  tmin < fmin(boxFar, m)  (fminf is exact, so the nesting order of the
// reference's min chain does not matter).
__device__ __forceinline__ void object_slab(const Ray& r, const ObjCull& o, float& tminOut, float& farOut)
{
    float dx, dy, dz, ox, oy, oz;
    mat_vec(o.inv, r.dx, r.dy, r.dz, 0.0f, dx, dy, dz);
    mat_vec(o.inv, r.ox, r.oy, r.oz, 1.0f, ox, oy, oz);
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;

    float tmin = 0.0f;
    float tmax = __builtin_inff();

    bool sg = __builtin_signbit(dx);
    float bmin = sg ? o.bmax[0] : o.bmin[0];
    float bmax = sg ? o.bmin[0] : o.bmax[0];
    float dmin = (bmin - ox) * ix;
    float dmax = (bmax - ox) * ix;
    tmin = rtd::fmax_f(dmin, tmin);
    tmax = rtd::fmin_f(dmax, tmax);

    sg = __builtin_signbit(dy);
    bmin = sg ? o.bmax[1] : o.bmin[1];
    bmax = sg ? o.bmin[1] : o.bmax[1];
    dmin = (bmin - oy) * iy;
    dmax = (bmax - oy) * iy;
    tmin = rtd::fmax_f(dmin, tmin);
    tmax = rtd::fmin_f(dmax, tmax);

    sg = __builtin_signbit(dz);
    bmin = sg ? o.bmax[2] : o.bmin[2];
    bmax = sg ? o.bmin[2] : o.bmax[2];
    dmin = (bmin - oz) * iz;
    dmax = (bmax - oz) * iz;
    tmin = rtd::fmax_f(dmin, tmin);
    tmax = rtd::fmin_f(dmax, tmax);

    tminOut = tmin;
    farOut = tmax;
}

__device__ __forceinline__ bool object_cull_pass(float tmin, float boxFar, float maxDist)
{
    return tmin < rtd::fmin_f(boxFar, maxDist);
}

__device__ __forceinline__ ObjCull load_obj(const ObjCull* __restrict__ p)
{
    return *p;
}

// ------------------------------------------------------------------ brute force (reference order)
__device__ __forceinline__ Hit trace_brute(const SceneView& s, Ray& ray, uint32_t& triTests)
{
    Hit best; best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
    const TriRay q = make_tri_ray(ray);
    for (uint32_t i = 0; i < s.nObjects; i++) {
        const ObjCull o = load_obj(s.objects + i);
        float tmin, bfar;
        object_slab(ray, o, tmin, bfar);
        if (!object_cull_pass(tmin, bfar, ray.maxDist)) continue;
        Hit ob; ob.t = FLT_MAX; ob.gid = kNoHit; ob.b0 = ob.b1 = ob.b2 = 0.0f;
        const uint32_t end = o.triStart + o.triCount;
        for (uint32_t g = o.triStart; g < end; g++) {
            const float4* p = reinterpret_cast<const float4*>(s.triRef + g);
            const float4 A = p[0], B = p[1], C = p[2];
            float t, b0, b1, b2;
            triTests++;
            if (tri_test(ray, q, ray.maxDist, A.x, A.y, A.z, B.x, B.y, B.z, C.x, C.y, C.z, t, b0, b1, b2)) {
                if (t < ob.t) { ob.t = t; ob.b0 = b0; ob.b1 = b1; ob.b2 = b2; ob.gid = g; }
            }
        }
        if (ob.gid != kNoHit && ob.t < best.t) {
            best = ob;
            ray.maxDist = ob.t;
        }
    }
    return best;
}

// ------------------------------------------------------------------ BVH
// Ray-side conservative padding: the slab test is evaluated against two shifted origins, which
// widens every box by `pad` along each axis in t-space terms (pad * |1/d|) at no per-box cost.
struct BoxRay {
    float nox, noy, noz;   // origin for near planes
    float fox, foy, foz;   // origin for far planes
    float ix, iy, iz;
    bool sx, sy, sz;       // direction sign bits
};

__device__ __forceinline__ BoxRay make_box_ray(const Ray& r, float sceneMaxAbs)
{
    BoxRay b;
    const float scale = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.ox), __builtin_fabsf(r.oy)),
                                        __builtin_fmaxf(__builtin_fabsf(r.oz), sceneMaxAbs));
    const float pad = scale * 3.8146973e-06f + 1e-30f; // 2^-18 relative: 32 ulps of the largest coordinate
    b.ix = 1.0f / r.dx; b.iy = 1.0f / r.dy; b.iz = 1.0f / r.dz;
    b.sx = __builtin_signbit(r.dx); b.sy = __builtin_signbit(r.dy); b.sz = __builtin_signbit(r.dz);
    b.nox = b.sx ? r.ox - pad : r.ox + pad;  b.fox = b.sx ? r.ox + pad : r.ox - pad;
    b.noy = b.sy ? r.oy - pad : r.oy + pad;  b.foy = b.sy ? r.oy + pad : r.oy - pad;
    b.noz = b.sz ? r.oz - pad : r.oz + pad;  b.foz = b.sz ? r.oz + pad : r.oz - pad;
    return b;
}

// returns entry distance (clamped at 0) or +inf-like miss via the bool
__device__ __forceinline__ bool box_test(const BoxRay& b, const float* __restrict__ mn, const float* __restrict__ mx,
                                         float bound, float& tnear)
{
    const float nx = b.sx ? mx[0] : mn[0], fx = b.sx ? mn[0] : mx[0];
    const float ny = b.sy ? mx[1] : mn[1], fy = b.sy ? mn[1] : mx[1];
    const float nz = b.sz ? mx[2] : mn[2], fz = b.sz ? mn[2] : mx[2];
    const float t0x = (nx - b.nox) * b.ix, t1x = (fx - b.fox) * b.ix;
    const float t0y = (ny - b.noy) * b.iy, t1y = (fy - b.foy) * b.iy;
    const float t0z = (nz - b.noz) * b.iz, t1z = (fz - b.foz) * b.iz;
    // __builtin_fmaxf/fminf drop NaN operands (0 * inf when a zero direction component meets a
    // padded plane exactly); such a ray is `pad` away from anything inside the box.
    const float tn = __builtin_fmaxf(__builtin_fmaxf(t0x, t0y), __builtin_fmaxf(t0z, 0.0f));
    const float tf = __builtin_fminf(__builtin_fminf(t1x, t1y), __builtin_fminf(t1z, bound));
    tnear = tn;
    return tn <= tf;
}

struct BvhCounters { uint32_t nodes, tris; };

// Traversal stack: the first kLdsStack entries of every lane live in LDS, laid out
// [entry][lane] so that lanes at the same depth hit distinct banks (ds latency ~64 cycles vs a
// scratch round trip through L1/L2); deeper entries, rare, spill to a per-lane scratch array.
struct Stack {
    uint32_t* lds;   // already offset by the lane's slot
    uint32_t* ovf;   // kStackDepth - kLdsStack entries of private memory
    __device__ __forceinline__ void push(int& sp, uint32_t v) const
    {
        if (sp < kLdsStack) lds[sp * kBlock] = v;
        else if (sp < kStackDepth) ovf[sp - kLdsStack] = v;
        sp++;
    }
    __device__ __forceinline__ uint32_t pop(int& sp) const
    {
        --sp;
        if (sp < kLdsStack) return lds[sp * kBlock];
        return sp < kStackDepth ? ovf[sp - kLdsStack] : 0u;
    }
};

// Band inside which a hit on a different object triggers the exact in-order emulation.
__device__ __forceinline__ float band_of(float t, float rayScale) { return 2e-4f * (t + rayScale); }

// Generic stack traversal.  `Visit` is called for every triangle of every reached leaf with its
// leaf-order record; it returns the (possibly reduced) prune bound.
// "while-while" traversal (Aila & Laine 2009): a lane that reaches a leaf parks it and keeps
// descending until every lane of the wave still searching has a leaf too; then the wave runs the
// triangle tests together.  Node visits and triangle tests no longer serialise against each other
// inside a wave (VALU lane utilisation of the first kernel was 32 %).
template <class Visit>
__device__ __forceinline__ void bvh_walk(const SceneView& s, const BoxRay& br, float bound, const Stack& stack,
                                         BvhCounters& cnt, Visit&& visit)
{
    if (!s.rootIsValid) return;
    constexpr int32_t kEnd = 0x7FFFFFFF; // traversal finished / no leaf parked (internal nodes are < kEnd, leaves < 0)
    int sp = 0;
    int32_t node = 0; // root
    int32_t leaf = kEnd;
    while (node != kEnd) {
        while (node >= 0 && node != kEnd) {
            const float4* np = reinterpret_cast<const float4*>(s.nodes + node);
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            cnt.nodes++;
            const float lmin[3] = { n0.x, n0.y, n0.z }, lmax[3] = { n0.w, n1.x, n1.y };
            const float rmin[3] = { n1.z, n1.w, n2.x }, rmax[3] = { n2.y, n2.z, n2.w };
            const int32_t left = __float_as_int(n3.x), right = __float_as_int(n3.y);
            float tl, tr;
            const bool hl = box_test(br, lmin, lmax, bound, tl);
            const bool hr = box_test(br, rmin, rmax, bound, tr);
            if (hl && hr) {
                const bool leftFirst = tl <= tr;
                stack.push(sp, (uint32_t)(leftFirst ? right : left));
                node = leftFirst ? left : right;
            } else if (hl) node = left;
            else if (hr) node = right;
            else node = sp > 0 ? (int32_t)stack.pop(sp) : kEnd;
            if (node < 0 && leaf == kEnd) { // park the first leaf, keep searching
                leaf = node;
                node = sp > 0 ? (int32_t)stack.pop(sp) : kEnd;
            }
            if (!__any(leaf == kEnd)) break; // every searching lane holds a leaf
        }
        while (leaf != kEnd) {
            const uint32_t code = (uint32_t)(~leaf);
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                const float4* p = reinterpret_cast<const float4*>(s.triLeaf + first + k);
                const float4 A = p[0], B = p[1], C = p[2];
                cnt.tris++;
                bound = visit(A, B, C, bound);
            }
            leaf = kEnd;
            if (node < 0) { // a second leaf was reached while the first was parked
                leaf = node;
                node = sp > 0 ? (int32_t)stack.pop(sp) : kEnd;
            }
        }
    }
}

struct CullCache {
    uint32_t obj;
    float tmin, bfar;
};

__device__ __forceinline__ void cull_lookup(const SceneView& s, const Ray& ray, uint32_t obj, CullCache& c)
{
    if (c.obj != obj) {
        const ObjCull o = load_obj(s.objects + obj);
        object_slab(ray, o, c.tmin, c.bfar);
        c.obj = obj;
    }
}

// Exact emulation of :247-265 restricted to the objects that have an accepted hit with
// t <= limit (objects beyond cannot change the outcome).
__device__ __noinline__ Hit trace_bvh_exact(const SceneView& s, Ray& ray, const TriRay& q, const BoxRay& br,
                                            float limit, const Stack& stack, BvhCounters& cnt)
{
    Hit best; best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
    const float m0 = ray.maxDist;
    float m = m0;
    uint32_t prev = kNoHit; // "no object yet"; object ids are < 2^31
    for (;;) {
        // next object id after `prev` that has an accepted hit within the limit
        uint32_t next = kNoHit;
        CullCache cc; cc.obj = kNoHit; cc.tmin = 0.0f; cc.bfar = 0.0f;
        bvh_walk(s, br, limit, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
            const uint32_t obj = __float_as_uint(B.w);
            if ((prev == kNoHit || obj > prev) && obj < next) {
                float t, b0, b1, b2;
                if (tri_test(ray, q, m0, A.x, A.y, A.z, B.x, B.y, B.z, C.x, C.y, C.z, t, b0, b1, b2) && t <= limit) {
                    cull_lookup(s, ray, obj, cc);
                    if (object_cull_pass(cc.tmin, cc.bfar, m0)) next = obj;
                }
            }
            return bound;
        });
        if (next == kNoHit) break;
        prev = next;
        cull_lookup(s, ray, next, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m)) continue;
        // closest hit inside that object with ray.maxDistance = m
        Hit ob; ob.t = FLT_MAX; ob.gid = kNoHit; ob.b0 = ob.b1 = ob.b2 = 0.0f;
        bvh_walk(s, br, m, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
            if (__float_as_uint(B.w) == next) {
                float t, b0, b1, b2;
                if (tri_test(ray, q, m, A.x, A.y, A.z, B.x, B.y, B.z, C.x, C.y, C.z, t, b0, b1, b2)) {
                    const uint32_t gid = __float_as_uint(A.w);
                    if (t < ob.t || (t == ob.t && gid < ob.gid)) {
                        ob.t = t; ob.b0 = b0; ob.b1 = b1; ob.b2 = b2; ob.gid = gid;
                    }
                }
            }
            return __builtin_fminf(bound, ob.t);
        });
        if (ob.gid != kNoHit && ob.t < best.t) {
            best = ob;
            m = ob.t;
        }
    }
    ray.maxDist = m;
    return best;
}

// EXACT = true: rays inside the band take trace_bvh_exact inline (rt_trace_rays, small kernels).
// EXACT = false: the hot render kernel only reports `danger`; the pixel is re-rendered by the
// redo launch, which keeps the rare path (and its registers) out of the hot kernel.
template <bool EXACT>
__device__ __forceinline__ Hit trace_bvh(const SceneView& s, Ray& ray, const Stack& stack, BvhCounters& cnt,
                                         uint32_t& slowPath, bool& danger)
{
    Hit best; best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
    if (!s.rootIsValid) return best;
    const TriRay q = make_tri_ray(ray);
    const BoxRay br = make_box_ray(ray, s.sceneMaxAbs);
    const float rayScale = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.ox), __builtin_fabsf(ray.oy)),
                                           __builtin_fmaxf(__builtin_fabsf(ray.oz), s.sceneMaxAbs));
    const float m0 = ray.maxDist;
    uint32_t bestObj = kNoHit;
    float otherT = FLT_MAX; // closest accepted hit on an object other than best's
    CullCache cc; cc.obj = kNoHit; cc.tmin = 0.0f; cc.bfar = 0.0f;

    bvh_walk(s, br, m0, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
        float t, b0, b1, b2;
        if (!tri_test(ray, q, m0, A.x, A.y, A.z, B.x, B.y, B.z, C.x, C.y, C.z, t, b0, b1, b2)) return bound;
        if (t > bound) return bound;
        const uint32_t gid = __float_as_uint(A.w), obj = __float_as_uint(B.w);
        cull_lookup(s, ray, obj, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m0)) return bound;
        const bool better = t < best.t || (t == best.t && gid < best.gid);
        if (obj == bestObj) {
            if (better) { best.t = t; best.b0 = b0; best.b1 = b1; best.b2 = b2; best.gid = gid; }
        } else if (better) {
            if (best.gid != kNoHit) otherT = best.t;
            best.t = t; best.b0 = b0; best.b1 = b1; best.b2 = b2; best.gid = gid; bestObj = obj;
        } else {
            otherT = __builtin_fminf(otherT, t);
        }
        return __builtin_fminf(m0, best.t + band_of(best.t, rayScale));
    });

    if (best.gid == kNoHit) return best;
    const float limit = best.t + band_of(best.t, rayScale);
    if (otherT <= limit) {
        slowPath++;
        if (EXACT) return trace_bvh_exact(s, ray, q, br, __builtin_fminf(limit, m0), stack, cnt);
        danger = true;
    }
    ray.maxDist = best.t;
    return best;
}

// the reference loop itself with the BVH used for culling only: every object with an accepted hit,
// in index order (redo launch)
__device__ __forceinline__ Hit trace_bvh_exact_all(const SceneView& s, Ray& ray, const Stack& stack, BvhCounters& cnt)
{
    Hit none; none.t = FLT_MAX; none.gid = kNoHit; none.b0 = none.b1 = none.b2 = 0.0f;
    if (!s.rootIsValid) return none;
    const TriRay q = make_tri_ray(ray);
    const BoxRay br = make_box_ray(ray, s.sceneMaxAbs);
    return trace_bvh_exact(s, ray, q, br, ray.maxDist, stack, cnt);
}

} // namespace rt
