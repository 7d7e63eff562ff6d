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
    float pox, poy, poz;   // origin in the rotated frame (kx, ky, kz)
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
    q.pox = sel3(q.kz, r.oy, r.oz, r.ox);
    q.poy = sel3(q.kz, r.oz, r.ox, r.oy);
    q.poz = sel3(q.kz, r.ox, r.oy, r.oz);
    return q;
}

// :551-637.  Returns true and fills t / barycentrics when the triangle is hit within
// (0.001, maxDist]; bit-for-bit the oracle's dist_to_triangle.
// core of the test on vertices already translated to the ray origin and rotated into (kx, ky, kz)
// tsOut / detOut: the operands of the reference's pre-filter on maxDist (tScaled against maxDist * det), for
// callers that must know whether a hit accepted under one maxDist is accepted under a smaller one
__device__ __forceinline__ bool tri_test_core(const TriRay& q, float maxDist, float p0x, float p0y, float p0z, float p1x,
                                              float p1y, float p1z, float p2x, float p2y, float p2z, float& tOut,
                                              float& b0Out, float& b1Out, float& b2Out, float* tsOut = nullptr,
                                              float* detOut = nullptr)
{
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

    // The reference leaves after each of these tests (:600-625).  Here they are evaluated together and the lane leaves once:
    // in a wave some lane nearly always goes on, so the separate exits only cost their own mask bookkeeping (a fifth of
    // the instructions of a triangle step in wf_trace).  Same comparisons on the same operands, so the same decisions.
    const bool mixed = (e0 < 0 || e1 < 0 || e2 < 0) & (e0 > 0 || e1 > 0 || e2 > 0);
    const float det = e0 + e1 + e2;

    p0z *= q.sz; p1z *= q.sz; p2z *= q.sz;
    const float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    const float md = maxDist * det;
    const bool rejNeg = (det < 0) & ((tScaled >= 0) | (tScaled < md));
    const bool rejPos = (det > 0) & ((tScaled <= 0) | (tScaled > md));
    if (mixed | (det == 0) | rejNeg | rejPos) return false;

    const float invDet = 1 / det;
    const float t = tScaled * invDet;
    // t < 0.001 (the reference compares in binary64; 0.001f is the smallest binary32 >= 0.001, so t < 0.001f decides the same
    // for every binary32 t), t > maxDist, and the reference's '<' against FLT_MAX that a FLT_MAX / NaN distance never wins (:421)
    if ((t < 0.001f) | (t > maxDist) | !(t < FLT_MAX)) return false;

    tOut = t; b0Out = e0 * invDet; b1Out = e1 * invDet; b2Out = e2 * invDet;
    if (tsOut) *tsOut = tScaled;
    if (detOut) *detOut = det;
    return true;
}

// world-space vertices (reference order records, brute force)
__device__ __forceinline__ bool tri_test(const Ray& r, const TriRay& q, float maxDist,
                                         float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                         float v2x, float v2y, float v2z, float& tOut, float& b0Out,
                                         float& b1Out, float& b2Out)
{
    const float a0x = v0x - r.ox, a0y = v0y - r.oy, a0z = v0z - r.oz;
    const float a1x = v1x - r.ox, a1y = v1y - r.oy, a1z = v1z - r.oz;
    const float a2x = v2x - r.ox, a2y = v2y - r.oy, a2z = v2z - r.oz;
    return tri_test_core(q, maxDist, sel3(q.kz, a0y, a0z, a0x), sel3(q.kz, a0z, a0x, a0y), sel3(q.kz, a0x, a0y, a0z),
                         sel3(q.kz, a1y, a1z, a1x), sel3(q.kz, a1z, a1x, a1y), sel3(q.kz, a1x, a1y, a1z),
                         sel3(q.kz, a2y, a2z, a2x), sel3(q.kz, a2z, a2x, a2y), sel3(q.kz, a2x, a2y, a2z), tOut, b0Out,
                         b1Out, b2Out);
}

// vertices from the leaf-order copy already rotated for this ray's kz (SceneView::triLeaf): the same
// subtractions on the same operands as above, without the 18 three-way selects per triangle
__device__ __forceinline__ bool tri_test_rot(const TriRay& q, float maxDist, const float4& A, const float4& B,
                                             const float4& C, float& tOut, float& b0Out, float& b1Out, float& b2Out,
                                             float* tsOut = nullptr, float* detOut = nullptr)
{
    return tri_test_core(q, maxDist, A.x - q.pox, A.y - q.poy, A.z - q.poz, B.x - q.pox, B.y - q.poy, B.z - q.poz,
                         C.x - q.pox, C.y - q.poy, C.z - q.poz, tOut, b0Out, b1Out, b2Out, tsOut, detOut);
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
typedef float f2 __attribute__((ext_vector_type(2)));

struct BoxRay {
    float ix, iy, iz;
    float cnx, cny, cnz;   // -(padded origin for near planes) / d
    float cfx, cfy, cfz;   // -(padded origin for far planes) / d
    bool sx, sy, sz;       // direction sign bits
};

__device__ __forceinline__ BoxRay make_box_ray(const Ray& r, float sceneMaxAbs)
{
    BoxRay b;
    const float scale = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.ox), __builtin_fabsf(r.oy)),
                                        __builtin_fmaxf(__builtin_fabsf(r.oz), sceneMaxAbs));
    // 2^-18 relative: 32 ulps of the largest coordinate.  The plane distances below are evaluated
    // as fma(plane, 1/d, -o/d), whose rounding error is half an ulp of |o| in plane position:
    // far inside the pad.
    const float pad = scale * 3.8146973e-06f + 1e-30f;
    // A direction component of (nearly) zero is replaced by +-1e-30: the slab distances stay finite
    // ((plane - o) * 1e30 < FLT_MAX for any scene that fits floats), the ray's true drift along that
    // axis over any distance of interest is far inside the pad, and a ray that runs inside a slab sees
    // it as open-ended.  With 1/0 the distances were inf - inf = NaN, the axis dropped out of the test,
    // and the centre column of an axis-aligned camera visited every box in its plane: single rays with
    // 10 000 node visits that set the length of the trace launches on C4 and C5.
    const float kTiny = 1e-30f;
    const float ddx = __builtin_fabsf(r.dx) < kTiny ? __builtin_copysignf(kTiny, r.dx) : r.dx;
    const float ddy = __builtin_fabsf(r.dy) < kTiny ? __builtin_copysignf(kTiny, r.dy) : r.dy;
    const float ddz = __builtin_fabsf(r.dz) < kTiny ? __builtin_copysignf(kTiny, r.dz) : r.dz;
    // v_rcp_f32 (1 ulp): the box test only has to be conservative, and an ulp of a plane distance is far inside the pad
    b.ix = __builtin_amdgcn_rcpf(ddx); b.iy = __builtin_amdgcn_rcpf(ddy); b.iz = __builtin_amdgcn_rcpf(ddz);
    b.sx = __builtin_signbit(r.dx); b.sy = __builtin_signbit(r.dy); b.sz = __builtin_signbit(r.dz);
    b.cnx = -((b.sx ? r.ox - pad : r.ox + pad) * b.ix);  b.cfx = -((b.sx ? r.ox + pad : r.ox - pad) * b.ix);
    b.cny = -((b.sy ? r.oy - pad : r.oy + pad) * b.iy);  b.cfy = -((b.sy ? r.oy + pad : r.oy - pad) * b.iy);
    b.cnz = -((b.sz ? r.oz - pad : r.oz + pad) * b.iz);  b.cfz = -((b.sz ? r.oz + pad : r.oz - pad) * b.iz);
    return b;
}

// Slab test of the four children of a node.  The test is internal to the product (the oracle has
// no BVH): it only has to be conservative, which the quantisation (outward rounding) and the pad
// ensure.  With plane = origin + q * scale, the distance to a plane is
//   (plane - o) / d = q * (scale / d) + (origin - o) / d = fma(q, a, b)
// with a, b per node and axis, so a child costs one byte-to-float conversion and one FMA per plane.
// fmax/fmin drop NaN operands, which errs on the side of visiting (make_box_ray keeps 1/d finite,
// so none arise from axis-parallel rays).
// key[c] = bits of the entry distance (>= 0, so unsigned order = float order) or ~0 for a miss.
constexpr uint32_t kMissKey = 0xFFFFFFFFu;

__device__ __forceinline__ f2 ubyte2_f(uint32_t v, int c) { return (f2){ (float)((v >> (8 * c)) & 0xFFu), (float)((v >> (8 * c + 8)) & 0xFFu) }; }
__device__ __forceinline__ f2 splat2(float v) { return (f2){ v, v }; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// tlo: distances below it are of no interest (0 in the hot kernels; the band walk of a ray whose closest hit is known
// skips the boxes that end before it).
// hit[c]: the compare behind key[c] != kMissKey, for callers that count or vote on it (a second compare per child otherwise)
template <bool KEYS = true>
__device__ __forceinline__ void node_test(const BoxRay& b, const float4& N0, const uint4& N1, const uint4& N2, float bound,
                                          uint32_t key[4], bool hit[4], const float tlo = 0.0f)
{
    // N0 = ox oy oz sx | N1 = sy sz lox loy | N2 = loz hix hiy hiz
    const f2 ax = splat2(N0.w * b.ix), ay = splat2(__uint_as_float(N1.x) * b.iy), az = splat2(__uint_as_float(N1.y) * b.iz);
    const f2 bnx = splat2(__builtin_fmaf(N0.x, b.ix, b.cnx)), bfx = splat2(__builtin_fmaf(N0.x, b.ix, b.cfx));
    const f2 bny = splat2(__builtin_fmaf(N0.y, b.iy, b.cny)), bfy = splat2(__builtin_fmaf(N0.y, b.iy, b.cfy));
    const f2 bnz = splat2(__builtin_fmaf(N0.z, b.iz, b.cnz)), bfz = splat2(__builtin_fmaf(N0.z, b.iz, b.cfz));
    const uint32_t qnx = b.sx ? N2.y : N1.z, qfx = b.sx ? N1.z : N2.y;
    const uint32_t qny = b.sy ? N2.z : N1.w, qfy = b.sy ? N1.w : N2.z;
    const uint32_t qnz = b.sz ? N2.w : N2.x, qfz = b.sz ? N2.x : N2.w;
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
        const f2 t0x = pk_fma(ubyte2_f(qnx, c), ax, bnx), t1x = pk_fma(ubyte2_f(qfx, c), ax, bfx);
        const f2 t0y = pk_fma(ubyte2_f(qny, c), ay, bny), t1y = pk_fma(ubyte2_f(qfy, c), ay, bfy);
        const f2 t0z = pk_fma(ubyte2_f(qnz, c), az, bnz), t1z = pk_fma(ubyte2_f(qfz, c), az, bfz);
        const float tn0 = __builtin_fmaxf(__builtin_fmaxf(t0x.x, t0y.x), __builtin_fmaxf(t0z.x, tlo));
        const float tn1 = __builtin_fmaxf(__builtin_fmaxf(t0x.y, t0y.y), __builtin_fmaxf(t0z.y, tlo));
        const float tf0 = __builtin_fminf(__builtin_fminf(t1x.x, t1y.x), __builtin_fminf(t1z.x, bound));
        const float tf1 = __builtin_fminf(__builtin_fminf(t1x.y, t1y.y), __builtin_fminf(t1z.y, bound));
        hit[c] = tn0 <= tf0; hit[c + 1] = tn1 <= tf1;
        if (KEYS) {
            key[c] = hit[c] ? __float_as_uint(tn0) : kMissKey;
            key[c + 1] = hit[c + 1] ? __float_as_uint(tn1) : kMissKey;
        } else {   // the entry distances as they are: the packet kernel orders the children by one lane's, read with that lane's verdicts
            key[c] = __float_as_uint(tn0); key[c + 1] = __float_as_uint(tn1);
        }
    }
}

struct BvhCounters { uint32_t nodes, tris; uint32_t idleA = 0, idleB = 0; uint32_t diagA = 0, diagB = 0; };

#ifdef RT_UTIL_COUNTERS
// diagnostic build: lanes idle while the wave runs a traversal-loop body (first active lane counts)
__device__ __forceinline__ void count_idle(uint32_t& acc)
{
    const unsigned long long m = rt_ballot(true);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) acc += 64u - (uint32_t)__popcll(m);
}
#else
__device__ __forceinline__ void count_idle(uint32_t&) {}
#endif

// Traversal stacks.  Entries live in LDS, laid out [entry][lane] so that lanes at the same depth
// hit distinct banks (ds latency ~64 cycles vs a scratch round trip through L1/L2).  Entry 0 holds
// the end-of-traversal sentinel, so a pop needs no emptiness test.
constexpr int32_t kTravEnd = 0x7FFFFFFF; // traversal finished / no leaf parked (internal nodes are < kTravEnd, leaves < 0)

// Full-depth stack of the exact kernels: entries beyond kLdsStack spill to a per-lane scratch array.
struct Stack {
    uint32_t* lds;   // already offset by the lane's slot
    uint32_t* ovf;   // kStackDepth - kLdsStack entries of private memory
    __device__ __forceinline__ void reset(int& sp) { lds[0] = (uint32_t)kTravEnd; sp = 1; }
    __device__ __forceinline__ void push(int& sp, uint32_t v)
    {
        if (sp < kLdsStack) lds[sp * kBlock] = v;
        else if (sp < kStackDepth) ovf[sp - kLdsStack] = v;
        sp++;
    }
    __device__ __forceinline__ uint32_t pop(int& sp)
    {
        --sp;
        if (sp < kLdsStack) return lds[sp * kBlock];
        return sp < kStackDepth ? ovf[sp - kLdsStack] : (uint32_t)kTravEnd;
    }
    // h hit children sorted front to back (c[0] is entered at once): push c[h-1] .. c[1]
    __device__ __forceinline__ void push_far(int& sp, int h, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        if (h >= 4) push(sp, c3);
        if (h >= 3) push(sp, c2);
        if (h >= 2) push(sp, c1);
    }
    __device__ __forceinline__ void push_far(int& sp, bool p1, bool p2, bool p3, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        if (p3) push(sp, c3);
        if (p2) push(sp, c2);
        if (p1) push(sp, c1);
    }
};

// LDS-only stack of the hot kernel: a push beyond kLdsStack entries is dropped and remembered; the
// pixel is then re-rendered by the redo launch (which uses Stack), so the hot loops carry no
// scratch path at all.
template <int CAP>
struct HotStackT {
    uint32_t* lds;
    bool dropped;
    __device__ __forceinline__ void reset(int& sp) { lds[0] = (uint32_t)kTravEnd; sp = 1; }
    __device__ __forceinline__ void push(int& sp, uint32_t v)
    {
        if (sp < CAP) { lds[sp * kBlock] = v; sp++; }
        else dropped = true;
    }
    __device__ __forceinline__ uint32_t pop(int& sp) { --sp; return lds[sp * kBlock]; }
    // three independent predicated stores, no bounds test per entry
    __device__ __forceinline__ void push_far(int& sp, int h, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        if (sp + 3 > CAP) { dropped = dropped || h >= 2; return; }
        uint32_t* top = lds + (sp + h) * kBlock;
        if (h >= 2) top[-2 * kBlock] = c1;
        if (h >= 3) top[-3 * kBlock] = c2;
        if (h >= 4) top[-4 * kBlock] = c3;
        sp += h >= 2 ? h - 1 : 0;
    }
    // the same from the sorted keys' own verdicts (p1: a second child was hit, ...): farthest first, so c3 lowest
    __device__ __forceinline__ void push_far(int& sp, bool p1, bool p2, bool p3, uint32_t c1, uint32_t c2, uint32_t c3)
    {
        if (sp + 3 > CAP) { dropped = dropped || p1; return; }
        const int n = (int)p1 + (int)p2 + (int)p3;
        uint32_t* top = lds + (sp + n) * kBlock;
        if (p1) top[-1 * kBlock] = c1;
        if (p2) top[-2 * kBlock] = c2;
        if (p3) top[-3 * kBlock] = c3;
        sp += n;
    }
};
using HotStack = HotStackT<kLdsStack>;

// Band inside which a hit on a different object triggers the exact in-order emulation.
// rel = SceneView::bandRel: 2e-4 for ordinary scenes, widened by rt_scene_commit to cover the rounding of the
// object-space cull against the world-space triangles when a model matrix is badly conditioned.
__device__ __forceinline__ float band_of(float t, float rayScale, float rel) { return rel * (t + rayScale); }

// One node visit of a lane: test the four children, sort the hit ones front to back (sorting
// network on (distance, child) pairs), enter the nearest, push the others farthest first; pop when
// none is hit.  On return `node` is the next node: internal (>= 0), a leaf (< 0) or kTravEnd.
__device__ __forceinline__ void cmp_swap(uint32_t& ka, uint32_t& kb, int32_t& ia, int32_t& ib)
{
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int32_t i0 = sw ? ib : ia, i1 = sw ? ia : ib;
    ka = k0; kb = k1; ia = i0; ib = i1;
}

template <class STACK>
__device__ __forceinline__ void node_step(const SceneView& s, const BoxRay& br, float bound, STACK& stack, int& sp,
                                          int32_t& node, const float tlo = 0.0f)
{
    const float4* np = reinterpret_cast<const float4*>(s.nodes + node);
    const float4 N0 = np[0];
    const uint4 N1 = reinterpret_cast<const uint4*>(np)[1], N2 = reinterpret_cast<const uint4*>(np)[2];
    const int4 CH = reinterpret_cast<const int4*>(np)[3];
    uint32_t k[4];
    bool hit[4];
    node_test(br, N0, N1, N2, bound, k, hit, tlo);
    int32_t c0 = CH.x, c1 = CH.y, c2 = CH.z, c3 = CH.w;
    (void)hit;
    cmp_swap(k[0], k[1], c0, c1); cmp_swap(k[2], k[3], c2, c3);
    cmp_swap(k[0], k[2], c0, c2); cmp_swap(k[1], k[3], c1, c3);
    cmp_swap(k[1], k[2], c1, c2);
    // sorted, the misses last: how many children were hit reads off the keys (no count to keep)
    stack.push_far(sp, k[1] != kMissKey, k[2] != kMissKey, k[3] != kMissKey, (uint32_t)c1, (uint32_t)c2, (uint32_t)c3);
    node = c0;
    if (k[0] == kMissKey) node = (int32_t)stack.pop(sp);
}

// The triangles no box can stand for (SceneView::thin; rt_commit.hip, thin_kernel).  A ray can only "hit" one where all three
// of its edges subtend less than ~2^-22 rad, i.e. within 2^-22 r^2 / len of the line it lies on, r the distance to it: the
// test below (eight times that, squared) lets nearly every ray skip the triangle; the records are the same for every lane.
__device__ __forceinline__ bool thin_near(const ThinTri& r, const Ray& ray)
{
    const float wx = r.px - ray.ox, wy = r.py - ray.oy, wz = r.pz - ray.oz;
    const float nx = ray.dy * r.uz - ray.dz * r.uy, ny = ray.dz * r.ux - ray.dx * r.uz, nz = ray.dx * r.uy - ray.dy * r.ux;
    const float num = wx * nx + wy * ny + wz * nz;                 // distance between the lines = |num| / |n|
    const float n2 = nx * nx + ny * ny + nz * nz;
    const float rad = r.k * (wx * wx + wy * wy + wz * wz + r.len2) + 1e-30f;   // >= 2^-19 (|w| + len)^2 / len, without the root
    return !(num * num > rad * rad * n2);                          // parallel lines (n = 0) and NaN stay in
}
// ... as a leaf copy would hold it for a ray of dominant axis kz
__device__ __forceinline__ void thin_record(const SceneView& s, uint32_t gid, int kz, float4& A, float4& B, float4& C)
{
    const TriIsect t = s.triRef[gid];
    A = make_float4(sel3(kz, t.v0y, t.v0z, t.v0x), sel3(kz, t.v0z, t.v0x, t.v0y), sel3(kz, t.v0x, t.v0y, t.v0z), __uint_as_float(t.gid));
    B = make_float4(sel3(kz, t.v1y, t.v1z, t.v1x), sel3(kz, t.v1z, t.v1x, t.v1y), sel3(kz, t.v1x, t.v1y, t.v1z), __uint_as_float(t.obj));
    C = make_float4(sel3(kz, t.v2y, t.v2z, t.v2x), sel3(kz, t.v2z, t.v2x, t.v2y), sel3(kz, t.v2x, t.v2y, t.v2z), __uint_as_float(t.mat));
}
template <class STACK, class Visit>
__device__ __forceinline__ void bvh_walk(const SceneView& s, const Ray& ray, const BoxRay& br, const TriRay& q, float bound, STACK& stack,
                                         BvhCounters& cnt, Visit&& visit, const float tlo = 0.0f)
{
    if (!s.rootIsValid) return;
    int sp;
    stack.reset(sp);
    int32_t node = 0; // root
    int32_t leaf = kTravEnd;
    while (node != kTravEnd) {
        while (node >= 0 && node != kTravEnd) {
            cnt.nodes++;
            count_idle(cnt.idleA);
            node_step(s, br, bound, stack, sp, node, tlo);
            if (node < 0 && leaf == kTravEnd) { // park the first leaf, keep searching
                leaf = node;
                node = (int32_t)stack.pop(sp);
            }
            if (!__any(leaf == kTravEnd)) break; // every searching lane holds a leaf
        }
        while (leaf != kTravEnd) {
            const uint32_t code = (uint32_t)(~leaf);
            const uint32_t first = code >> 3, count = code & 7u;
            for (uint32_t k = 0; k < count; k++) {
                const float4* p = reinterpret_cast<const float4*>(s.triLeaf + (size_t)q.kz * s.nTris + first + k);
                const float4 A = p[0], B = p[1], C = p[2];
                cnt.tris++;
                count_idle(cnt.idleB);
                bound = visit(A, B, C, bound);
            }
            leaf = kTravEnd;
            if (node < 0) { // a second leaf was reached while the first was parked
                leaf = node;
                node = (int32_t)stack.pop(sp);
            }
        }
    }
    // the triangles no box can stand for (rt_commit.hip, thin_kernel): a handful at most in a sane scene, none in most
#ifndef RT_DIAG_NO_THIN_ALL   // diagnostic build: the grown boxes of the tree alone (rt_commit.hip, tri_box_walk)
    for (uint32_t k = 0; k < s.nThin; k++) {
        const ThinTri r = s.thin[k];
        if (!thin_near(r, ray)) continue;
        float4 A, B, C;
        thin_record(s, r.gid, q.kz, A, B, C);
        cnt.tris++;
        bound = visit(A, B, C, bound);
    }
#endif
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
__device__ __noinline__ Hit trace_bvh_exact(const SceneView s, Ray& ray, const TriRay& q, const BoxRay& br,
                                            float limit, Stack& stack, BvhCounters& cnt)
{
    Hit best; best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
    const float m0 = ray.maxDist;
    float m = m0;
    uint32_t prev = kNoHit; // "no object yet"; object ids are < 2^31
    bool widen = false;     // an object in the list whose box entry lies past the limit
    for (;;) {
        // next object id after `prev` that has an accepted hit within the limit
        uint32_t next = kNoHit;
        CullCache cc; cc.obj = kNoHit; cc.tmin = 0.0f; cc.bfar = 0.0f;
        bvh_walk(s, ray, br, q, limit, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
            const uint32_t obj = __float_as_uint(B.w);
            if ((prev == kNoHit || obj > prev) && obj < next) {
                float t, b0, b1, b2;
                if (tri_test_rot(q, m0, A, B, C, t, b0, b1, b2) && t <= limit) {
                    cull_lookup(s, ray, obj, cc);
                    if (object_cull_pass(cc.tmin, cc.bfar, m0)) { next = obj; if (cc.tmin > limit) widen = true; }
                }
            }
            return bound;
        });
        if (next == kNoHit) {
            // the list is done.  If one of its objects enters its box past the limit, objects past the limit can decide
            // that object's cull: once more, over the whole list
            if (!(widen && limit < m0)) break;
            limit = m0; widen = false; prev = kNoHit; m = m0;
            best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
            continue;
        }
        prev = next;
        cull_lookup(s, ray, next, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m)) continue;
        // closest hit inside that object with ray.maxDistance = m
        Hit ob; ob.t = FLT_MAX; ob.gid = kNoHit; ob.b0 = ob.b1 = ob.b2 = 0.0f;
        bvh_walk(s, ray, br, q, m, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
            if (__float_as_uint(B.w) == next) {
                float t, b0, b1, b2;
                if (tri_test_rot(q, m, A, B, C, t, b0, b1, b2)) {
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

// The same emulation in one walk, for a ray already known to sit in the tie band (wf_exact); tBest is the
// distance of the closest hit that the band-free traversal found, without any object cull.  The walk,
// bounded by the band limit, keeps per object the closest hit accepted under m0 -- kBandObjects objects fit: the
// floor under a bin, a sphere sunk into the terrain, neighbouring piles leaning into each other -- and the reference's loop over the objects in id
// order (:247-265) is replayed on that table: an object's closest hit under a smaller maxDistance m is
// the same triangle as long as that triangle is still accepted (acceptance is monotone in m), and no hit
// at all when its distance exceeds m.  Returns false, and the caller runs the full emulation above, when
//   - more than kBandObjects objects have a hit inside the limit,
//   - a hit's object fails its cull under m0 (tBest may then not be the closest accepted hit), or
//   - a table hit lies within m but the reference's pre-filter (tScaled against m * det) rejects it:
//     another triangle of that object could then be the one the reference finds.
struct BandEntry {
    uint32_t obj;
    Hit h;
    float ts, det;
};
constexpr int kBandObjects = 4;   // objects the table holds: a floor, the pile standing on it and two neighbours leaning into it

__device__ __forceinline__ bool trace_band_table(const SceneView& s, Ray& ray, float tBest, Stack& stack, BvhCounters& cnt,
                                                 Hit& out, bool* needAll = nullptr)
{
    const TriRay q = make_tri_ray(ray);
    const BoxRay br = make_box_ray(ray, s.sceneMaxAbs);
    const float rayScale = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.ox), __builtin_fabsf(ray.oy)),
                                           __builtin_fmaxf(__builtin_fabsf(ray.oz), s.sceneMaxAbs));
    const float m0 = ray.maxDist;
    const float limit = __builtin_fminf(tBest + band_of(tBest, rayScale, s.bandRel), m0);
    // tBest is the closest accepted hit on any triangle, so the hits of interest lie in [tBest, limit]: boxes that end before
    // tBest (less the band, for the slack of the box test) hold none of them.  Without this the walk visited everything the
    // ray passes on its way to the hit, one dependent fetch after the other: the longest rays set the length of wf_exact.
    const float tLow = __builtin_fmaxf(tBest - band_of(tBest, rayScale, s.bandRel), 0.0f);
    BandEntry e[kBandObjects];
#pragma unroll
    for (int k = 0; k < kBandObjects; k++) {
        e[k].obj = kNoHit;
        e[k].h.t = FLT_MAX; e[k].h.gid = kNoHit; e[k].h.b0 = e[k].h.b1 = e[k].h.b2 = 0.0f;
        e[k].ts = e[k].det = 0.0f;
    }
    bool ok = true, early = false;
    CullCache cc; cc.obj = kNoHit; cc.tmin = 0.0f; cc.bfar = 0.0f;
    bvh_walk(s, ray, br, q, limit, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
        float t, b0, b1, b2, ts, det;
        if (!tri_test_rot(q, m0, A, B, C, t, b0, b1, b2, &ts, &det) || t > limit) return bound;
        const uint32_t gid = __float_as_uint(A.w), obj = __float_as_uint(B.w);
        cull_lookup(s, ray, obj, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m0)) ok = false;
        else if (cc.tmin > limit) { ok = false; early = true; }   // its box entry lies past the band: only the whole object list settles it (trace_bvh)
        else {
            bool placed = false; // the entry of this object, or the first free one
#pragma unroll
            for (int k = 0; k < kBandObjects; k++) {
                if (!placed && (e[k].obj == kNoHit || e[k].obj == obj)) {
                    placed = true;
                    e[k].obj = obj;
                    if (t < e[k].h.t || (t == e[k].h.t && gid < e[k].h.gid)) {
                        e[k].h.t = t; e[k].h.b0 = b0; e[k].h.b1 = b1; e[k].h.b2 = b2; e[k].h.gid = gid; e[k].ts = ts; e[k].det = det;
                    }
                }
            }
            if (!placed) ok = false;
        }
        return ok ? bound : -1.0f; // a bound below every entry distance: the walk only drains its stack
    }, tLow);
    if (needAll) *needAll = early;
    if (!ok) return false;
    // id order, empty entries (kNoHit) last: the reference's loop over the objects (:247-265)
#pragma unroll
    for (int i = 0; i < kBandObjects - 1; i++)
#pragma unroll
        for (int k = 0; k < kBandObjects - 1 - i; k++)
            if (e[k + 1].obj < e[k].obj) { const BandEntry x = e[k]; e[k] = e[k + 1]; e[k + 1] = x; }
    Hit best; best.t = FLT_MAX; best.gid = kNoHit; best.b0 = best.b1 = best.b2 = 0.0f;
    float m = m0;
#pragma unroll
    for (int k = 0; k < kBandObjects; k++) {
        if (e[k].obj == kNoHit) continue;
        cull_lookup(s, ray, e[k].obj, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m)) continue;
        if (e[k].h.t > m) continue;
        const float md = m * e[k].det;
        if (e[k].det < 0 ? e[k].ts < md : e[k].ts > md) return false;
        if (e[k].h.t < best.t) { best = e[k].h; m = e[k].h.t; }
    }
    ray.maxDist = m;
    out = best;
    return true;
}

// Closest hit with the exact in-order emulation inline for rays inside the tie band (rt_trace_rays,
// redo launches).  The hot kernels use the resumable traversal below instead, which only reports
// `danger`: the pixel is then re-rendered by a redo launch, which keeps the rare path (and its
// registers) out of them.
__device__ __forceinline__ Hit trace_bvh(const SceneView& s, Ray& ray, Stack& stack, BvhCounters& cnt,
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
    // In exact arithmetic a hit lies inside its object's box interval [tmin, far].  The reference's slab test runs in object
    // space on a ray pushed through the host's float inverse, and for a ray that runs almost inside one of the box's planes
    // (an axis-parallel ray against an object turned by 90 degrees, whose matrix holds 4e-8 where it means 0) the entry
    // distance can land anywhere: tests/soak_parity.py found hits at t = 7.0 on objects whose box the ray "enters" at 7.42.
    // Such an object passes its cull under the ray's own maxDistance and fails it as soon as ANY earlier object in the list
    // has narrowed maxDistance below tmin (:252-262) -- a dependence on the object order that reaches as far as tmin, not
    // just across the tie band.  So: a winner, or any of the band's candidates, whose box entry lies past the band sends the
    // ray through the emulation of the whole object list (limit = the ray's maxDistance); inside the band the replay below
    // evaluates every cull with the narrowed maxDistance anyway (a triangle that lies in a face of its box has t = tmin up
    // to rounding: common, and harmless).  12 of 9 M soak rays.
    float bestTmin = 0.0f;

    bvh_walk(s, ray, br, q, m0, stack, cnt, [&](const float4& A, const float4& B, const float4& C, float bound) {
        float t, b0, b1, b2;
        if (!tri_test_rot(q, m0, A, B, C, t, b0, b1, b2)) return bound;
        if (t > bound) return bound;
        const uint32_t gid = __float_as_uint(A.w), obj = __float_as_uint(B.w);
        cull_lookup(s, ray, obj, cc);
        if (!object_cull_pass(cc.tmin, cc.bfar, m0)) return bound;
        const bool better = t < best.t || (t == best.t && gid < best.gid);
        if (better) bestTmin = cc.tmin;
        if (obj == bestObj) {
            if (better) { best.t = t; best.b0 = b0; best.b1 = b1; best.b2 = b2; best.gid = gid; }
        } else if (better) {
            if (best.gid != kNoHit) otherT = best.t;
            best.t = t; best.b0 = b0; best.b1 = b1; best.b2 = b2; best.gid = gid; bestObj = obj;
        } else {
            otherT = __builtin_fminf(otherT, t);
        }
        return __builtin_fminf(m0, best.t + band_of(best.t, rayScale, s.bandRel));
    });

    if (best.gid == kNoHit) return best;
    const float limit = best.t + band_of(best.t, rayScale, s.bandRel);
    if (bestTmin > limit) { slowPath++; return trace_bvh_exact(s, ray, q, br, m0, stack, cnt); }
    if (otherT < FLT_MAX && otherT <= limit) { // first the one-walk replay on the object table, then the general emulation
        slowPath++;
        Hit out;
        bool needAll = false;
        if (trace_band_table(s, ray, best.t, stack, cnt, out, &needAll)) return out;
        return trace_bvh_exact(s, ray, q, br, needAll ? m0 : __builtin_fminf(limit, m0), stack, cnt);
    }
    ray.maxDist = best.t;
    return best;
}

// ------------------------------------------------------------------ resumable traversal
// The hot kernel keeps the traversal state of every lane in registers across its main loop, so
// that a wave can leave the traversal as soon as most of its lanes are done, shade and re-arm
// those lanes, and come back.
//
// Scheduling inside a burst: a lane may have a node to visit (node >= 0), triangles to test
// (triCount > 0, the rest of one leaf) or both; a second leaf reached while triangles are pending
// stays in `node` (< 0) and blocks further node visits.  Every iteration the wave runs the kind of
// step that more lanes are waiting for.  Measured with the while-while loops this replaces, a
// triangle-loop iteration ran with 16-25 of 64 lanes and a node-loop iteration with 25-33.
struct Trav {
    int32_t node;
    uint32_t triFirst, triCount, triBase;
    int sp;
    float bound, m0;
    Hit best;
    uint32_t bestObj;
    float otherT;
    CullCache cc;
    TriRay q;
    BoxRay br;
};
// scale of the tie band: the largest |coordinate| of the ray origin or the scene -- from the rotated origin the triangle test
// keeps (the same three values in another order; max is exact), so that no register holds it across the loops
__device__ __forceinline__ float trav_ray_scale(const Trav& T, const SceneView& s)
{
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(T.q.pox), __builtin_fabsf(T.q.poy)),
                           __builtin_fmaxf(__builtin_fabsf(T.q.poz), s.sceneMaxAbs));
}
// per-ray part of the state (everything but the lane's position in the tree)
__device__ __forceinline__ void trav_init(Trav& T, const SceneView& s, const Ray& ray)
{
    T.q = make_tri_ray(ray);
    T.br = make_box_ray(ray, s.sceneMaxAbs);
    T.m0 = ray.maxDist; T.bound = ray.maxDist;
    T.best.t = FLT_MAX; T.best.gid = kNoHit; T.best.b0 = T.best.b1 = T.best.b2 = 0.0f;
    T.bestObj = kNoHit; T.otherT = FLT_MAX;
    T.cc.obj = kNoHit; T.cc.tmin = 0.0f; T.cc.bfar = 0.0f;
    T.triFirst = 0; T.triCount = 0; T.triBase = (uint32_t)T.q.kz * s.nTris;
    T.sp = 0; T.node = kTravEnd;
}

template <class HS>
__device__ __forceinline__ void trav_begin(Trav& T, const SceneView& s, const Ray& ray, HS& stack)
{
    trav_init(T, s, ray);
    stack.reset(T.sp);
    T.node = s.rootIsValid ? 0 : kTravEnd;
}

__device__ __forceinline__ bool trav_running(const Trav& T) { return T.node != kTravEnd || T.triCount != 0u; }

// a leaf in `node` becomes the pending triangle range once the previous range is used up
template <class HS>
__device__ __forceinline__ void trav_take_leaf(Trav& T, const SceneView& s, HS& stack)
{
    if (T.triCount == 0u && T.node < 0) {
        const uint32_t code = (uint32_t)(~T.node);
        T.triFirst = (code >> 3) + T.triBase; T.triCount = code & 7u;
        T.node = (int32_t)stack.pop(T.sp);
    }
}

// One triangle of a reached leaf against the lane's ray: the reference's test on the rotated record,
// then the bookkeeping of the winner and of the closest hit on any other object.
// The object-space cull of the winner's object (GPURaytracer.cu:267-315) is checked once, by the
// shading side: a hit on an object whose cull fails (flat meshes) sends the pixel to the
// redo launch instead of costing every triangle step a 96-byte fetch and a slab test.
__device__ __forceinline__ void trav_test_triangle(Trav& T, const SceneView& s, const float4& A, const float4& B, const float4& C)
{
    float t, b0, b1, b2;
    if (tri_test_rot(T.q, T.m0, A, B, C, t, b0, b1, b2) && t <= T.bound) {
        // selects, not branches: the lanes that get here differ in which of these apply
        const uint32_t gid = __float_as_uint(A.w), obj = __float_as_uint(B.w);
        const bool better = (t < T.best.t) | ((t == T.best.t) & (gid < T.best.gid));
        const bool other = obj != T.bestObj;
        const float displaced = T.best.gid != kNoHit ? T.best.t : T.otherT;   // the hit a better one on another object pushes out
        T.otherT = other ? (better ? displaced : __builtin_fminf(T.otherT, t)) : T.otherT;
        T.bestObj = better ? obj : T.bestObj;
        T.best.t = better ? t : T.best.t; T.best.b0 = better ? b0 : T.best.b0; T.best.b1 = better ? b1 : T.best.b1;
        T.best.b2 = better ? b2 : T.best.b2; T.best.gid = better ? gid : T.best.gid;
        T.bound = better ? __builtin_fminf(T.m0, t + band_of(t, trav_ray_scale(T, s), s.bandRel)) : T.bound;
    }
}

template <class HS>
__device__ __forceinline__ void trav_tri_step(Trav& T, const SceneView& s, const Ray& ray, HS& stack, BvhCounters& cnt)
{
    if (T.triCount != 0u) {
        const float4* p = reinterpret_cast<const float4*>(s.triLeaf + T.triFirst);
        const float4 A = p[0], B = p[1], C = p[2];
        T.triFirst++; T.triCount--;   // (counted by trav_burst, from the vote)
        trav_test_triangle(T, s, A, B, C);
    }
    trav_take_leaf(T, s, stack);
}

#ifndef RT_TRI_NUM
#define RT_TRI_NUM 4
#endif
#define RT_TRI_DEN 4
// Runs traversal steps until fewer than `keep` lanes of the wave are still traversing; the others wait with their result.
// Called by every lane of the wave: a lane without an unfinished query (trav_running false) wants no step and idles, so the
// loop control and the statistics -- node and triangle tests counted per step from the votes, into wave-uniform sums that
// live in scalar registers -- are the same for all lanes.
template <class HS>
__device__ __forceinline__ void trav_burst(Trav& T, const SceneView& s, const Ray& ray, HS& stack,
                                           BvhCounters& cnt, int keepIn, uint32_t& uNodes, uint32_t& uTris)
{
    const int keep = __builtin_amdgcn_readfirstlane(keepIn);   // the same in every lane: says so to the compiler, whose loop is then a scalar one
#ifdef RT_LOOP_TIMERS
    const bool leader = (int)(threadIdx.x & 63) == __builtin_ctzll(rt_ballot(true));
#endif
    for (;;) {
#ifdef RT_LOOP_TIMERS
        const long long tl0 = clock64();
#endif
        const bool wantNode = T.node >= 0 && T.node != kTravEnd;
        const bool wantTri = T.triCount != 0u || T.node < 0;
        // the votes as lane masks of single compares, combined in scalar registers: the vote of a compound condition
        // costs a select and a second compare per lane on top of the compares that make it.  (The vote on a leaf parked in
        // `node` is not redundant with the one on pending triangles: an empty slot's leaf -- reached by a ray with non-finite
        // components, whose box tests all pass -- leaves the pending range empty, and the lane moves on through a triangle step.)
        const unsigned long long mNode = rt_ballot(wantNode), mCount = rt_ballot(T.triCount != 0u), mLeaf = rt_ballot(T.node < 0);
        const unsigned long long mTri = mCount | mLeaf;
        // (counts of the two halves: the compiler knows a 64-bit count and then compares the two as 64-bit values, in the vector unit)
        const uint32_t nNode = (uint32_t)__builtin_popcount((uint32_t)mNode) + (uint32_t)__builtin_popcount((uint32_t)(mNode >> 32));
        const uint32_t nTri = (uint32_t)__builtin_popcount((uint32_t)mTri) + (uint32_t)__builtin_popcount((uint32_t)(mTri >> 32));
        // leave when most of the wave has finished: the finished lanes get shaded and re-armed
        if ((mNode | mTri) == 0ull || (int)__popcll(mNode | mTri) < keep) break;
        if (nTri * (uint32_t)RT_TRI_DEN >= nNode * (uint32_t)RT_TRI_NUM) {
            uTris += (uint32_t)__popcll(mCount);
#ifdef RT_LOOP_TIMERS
            if (wantTri) {
                count_idle(cnt.idleB);
                if ((int)(threadIdx.x & 63) == __builtin_ctzll(rt_ballot(true))) cnt.diagB++;
            }
#endif
            trav_tri_step(T, s, ray, stack, cnt);   // (a lane with no triangle pending does nothing in it)
#ifdef RT_LOOP_TIMERS
            if (leader) cnt.idleB += (uint32_t)((clock64() - tl0) >> 4);
#endif
        } else {
            uNodes += (uint32_t)nNode;
            if (wantNode) {
                count_idle(cnt.idleA);
#ifdef RT_LOOP_TIMERS
                if ((int)(threadIdx.x & 63) == __builtin_ctzll(rt_ballot(true))) cnt.diagA++;
#endif
                node_step(s, T.br, T.bound, stack, T.sp, T.node);
                trav_take_leaf(T, s, stack);
            }
#ifdef RT_LOOP_TIMERS
            if (leader) cnt.idleA += (uint32_t)((clock64() - tl0) >> 4);
#endif
        }
    }
}

// Closest hit of a finished traversal; `danger` = another object has a hit inside the tie band.
__device__ __forceinline__ Hit trav_result(const Trav& T, const SceneView& s, Ray& ray, bool& danger)
{
    if (T.best.gid != kNoHit) {
        const float limit = T.best.t + band_of(T.best.t, trav_ray_scale(T, s), s.bandRel);
        if (T.otherT < FLT_MAX && T.otherT <= limit) danger = true;   // FLT_MAX = no other object hit (the limit may be infinite)
        ray.maxDist = T.best.t;
    }
    return T.best;
}

// ... plus the object-space cull of the winner's object, which the triangle steps skip
__device__ __forceinline__ Hit trav_finish(Trav& T, const SceneView& s, Ray& ray, bool& danger)
{
    if (T.best.gid != kNoHit) {
        cull_lookup(s, ray, T.bestObj, T.cc);
        if (!object_cull_pass(T.cc.tmin, T.cc.bfar, T.m0) || T.cc.tmin > T.best.t + band_of(T.best.t, trav_ray_scale(T, s), s.bandRel)) danger = true;   // see trace_bvh on the second test
    }
    return trav_result(T, s, ray, danger);
}

} // namespace rt
