// rt_render.hip -- the render kernels for gfx950 (wave64).
//
// Replaces raytraceKernel and everything it inlines (processor/GPU/GPURaytracer.cu:78-885) plus
// the GPUMaterial getters (Material.cu:268-323).  Result semantics follow the reference line by
// line (each device function cites it); the execution shape does not.  Three pipelines share the
// device functions of the first half of this file:
//  * wavefront (default): wf_gen / wf_trace + wf_shade per bounce level / wf_resolve / wf_redo over
//    path state kept in HBM, several chunks of the frame in flight on their own HIP streams;
//  * path_kernel: one persistent kernel, lane = pixel, path regeneration (RT_OPT_TRAVERSAL = 2);
//  * render_kernel<LOG2L, MODE>: one thread per (pixel, z lane) with the reference's nested loops --
//    brute force in reference order (MODE 1) and the persistent kernel's redo launch (MODE 2).
// Geometry comes from the flattened world-space records of rt_device.h, through a four-wide BVH
// (rt_trace.h).  gfx950 has no texture unit: bilinear filtering is done in software on RGBA8 texels
// (value = byte/255.0f at fetch, the reference's own conversion, Material.cu:203).
#include "rt_trace.h"
#include "rt_commit.h"

namespace rt {

constexpr float kPi = 3.141592f; // processor/Raytracer.hpp:16

struct V4 { float x, y, z, w; };
__device__ __forceinline__ V4 mk4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ float dot3(const V4& a, const V4& b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // :781
__device__ __forceinline__ V4 normalize4(const V4& v) // :807-811, rsqrtf := 1/sqrtf
{
    const float inv = rtd::rsqrt_f(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk4(v.x * inv, v.y * inv, v.z * inv, v.w);
}
__device__ __forceinline__ V4 cross4(const V4& a, const V4& b) // :771-779
{
    return mk4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f);
}
__device__ __forceinline__ V4 lerp4(const V4& a, const V4& b, float t) // :796-805
{
    return mk4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
__device__ __forceinline__ V4 reflect4(const V4& v, const V4& n) // :861-872
{
    const float d = dot3(v, n);
    return normalize4(mk4(v.x - 2.0f * d * n.x, v.y - 2.0f * d * n.y, v.z - 2.0f * d * n.z, 0.0f));
}
__device__ __forceinline__ V4 rotate_q(const V4& v, const float* q) // :700-717, q = x,y,z,w
{
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float t2 = qw * qx, t3 = qw * qy, t4 = qw * qz;
    const float t5 = -qx * qx, t6 = qx * qy, t7 = qx * qz;
    const float t8 = -qy * qy, t9 = qy * qz, t10 = -qz * qz;
    return mk4(2.0f * ((t8 + t10) * v.x + (t6 - t4) * v.y + (t3 + t7) * v.z) + v.x,
               2.0f * ((t4 + t6) * v.x + (t5 + t10) * v.y + (t9 - t2) * v.z) + v.y,
               2.0f * ((t7 - t3) * v.x + (t2 + t9) * v.y + (t5 + t8) * v.z) + v.z, v.w);
}

// :819-825
__device__ __forceinline__ float random_value(uint32_t& seed)
{
    seed = seed * 747796405u + 2891336453u;
    uint32_t r = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
    r = (r >> 22) ^ r;
    return rtd::u32_unorm(r); // = (float)(r / 4294967295.0)
}
// :827-833
__device__ __forceinline__ float random_normal(uint32_t& seed)
{
    const float theta = (2.0f * kPi) * random_value(seed);
    const float r = random_value(seed);
    const float rho = __builtin_sqrtf(-2.0f * rtd::log_f(r));
    return rho * rtd::cos_f(theta);
}
// :835-841, :853-859
__device__ __forceinline__ V4 random_cosine_hemi(uint32_t& seed, const V4& n)
{
    const float x = random_normal(seed);
    const float y = random_normal(seed);
    const float z = random_normal(seed);
    const V4 u = normalize4(mk4(x, y, z, 0.0f));
    return normalize4(mk4(u.x + n.x, u.y + n.y, u.z + n.z, 0.0f));
}

// ---- software bilinear (pinned tex2D definition) ----
struct Tap { int i0, i1; float a; };
__device__ __forceinline__ Tap make_tap(float xB, int n)
{
    if (!(xB >= -1.0f)) xB = -1.0f;
    if (xB > (float)n) xB = (float)n;
    const float f = __builtin_floorf(xB);
    float a = xB - f;
    a = __builtin_floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i = (int)f, i1 = i + 1;
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    i1 = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
    Tap t; t.i0 = i; t.i1 = i1; t.a = a;
    return t;
}
__device__ __forceinline__ V4 bilerp(const V4& t00, const V4& t10, const V4& t01, const V4& t11, float a, float b)
{
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return mk4(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x,
               ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
               ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z,
               ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w);
}
__device__ __forceinline__ V4 texel8(const DevTexture& t, int x, int y)
{
    const uchar4 p = t.data[(size_t)y * (size_t)t.w + (size_t)x];
    return mk4(rtd::byte_unorm(p.x), rtd::byte_unorm(p.y), rtd::byte_unorm(p.z), rtd::byte_unorm(p.w)); // = p / 255.0f
}
// Material.cu:224-228: normalized coordinates, linear, clamp
__device__ __forceinline__ V4 tex2d_norm(const DevTexture& t, float u, float v)
{
    const Tap tx = make_tap(u * (float)t.w - 0.5f, t.w);
    const Tap ty = make_tap(v * (float)t.h - 0.5f, t.h);
    return bilerp(texel8(t, tx.i0, ty.i0), texel8(t, tx.i1, ty.i0), texel8(t, tx.i0, ty.i1), texel8(t, tx.i1, ty.i1),
                  tx.a, ty.a);
}
// The roughness and metal maps are read in their first channel only (Material.cu:268-323): the material record points at a
// one-byte-per-texel copy of that channel (rt_api.hip, to_dev), a quarter of the footprint for the same texels -- on the
// incoherent hits of the bounce levels the texel fetches are what wf_shade waits for (C4: 70 % of its L2 requests miss).
__device__ __forceinline__ float tex2d_norm_red(const DevTexture& t, float u, float v)
{
    const uint8_t* __restrict__ d = reinterpret_cast<const uint8_t*>(t.data);
    const Tap tx = make_tap(u * (float)t.w - 0.5f, t.w);
    const Tap ty = make_tap(v * (float)t.h - 0.5f, t.h);
    const float t00 = rtd::byte_unorm(d[(size_t)ty.i0 * (size_t)t.w + (size_t)tx.i0]), t10 = rtd::byte_unorm(d[(size_t)ty.i0 * (size_t)t.w + (size_t)tx.i1]);
    const float t01 = rtd::byte_unorm(d[(size_t)ty.i1 * (size_t)t.w + (size_t)tx.i0]), t11 = rtd::byte_unorm(d[(size_t)ty.i1 * (size_t)t.w + (size_t)tx.i1]);
    const float a = tx.a, b = ty.a;
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;   // bilerp().x
}
__device__ __forceinline__ V4 ray_texel(const float4* __restrict__ tab, int w, int x, int y)
{
    const float4 p = tab[(size_t)y * (size_t)w + (size_t)x];
    return mk4(p.x, p.y, p.z, p.w);
}
// Camera.hpp:93-97: unnormalized coordinates, linear, clamp
__device__ __forceinline__ V4 tex2d_rays(const float4* __restrict__ tab, int w, int h, float x, float y)
{
    const Tap tx = make_tap(x - 0.5f, w);
    const Tap ty = make_tap(y - 0.5f, h);
    return bilerp(ray_texel(tab, w, tx.i0, ty.i0), ray_texel(tab, w, tx.i1, ty.i0), ray_texel(tab, w, tx.i0, ty.i1),
                  ray_texel(tab, w, tx.i1, ty.i1), tx.a, ty.a);
}

// :4-20
__device__ __forceinline__ V4 sky_light(const V4& dir, const RenderArgs& a)
{
    const V4 l = mk4(a.skyDir[0], a.skyDir[1], a.skyDir[2], a.skyDir[3]);
    const float sun = rtd::pow1000_f(rtd::fmax_f(0.0f, dot3(dir, l))) * a.lightColor[3];
    return mk4(a.skyColor[0] * a.skyColor[3] + a.lightColor[0] * sun, a.skyColor[1] * a.skyColor[3] + a.lightColor[1] * sun,
               a.skyColor[2] * a.skyColor[3] + a.lightColor[2] * sun, 1.0f);
}

// :640-686
__device__ __forceinline__ float distribution_ggx(const V4& n, const V4& h, float roughness)
{
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float ndh = rtd::fmax_f(0.0f, dot3(n, h));
    const float ndh2 = ndh * ndh;
    float denom = (ndh2 * (a2 - 1.0f) + 1.0f);
    denom = kPi * denom * denom;
    return a2 / (denom + 0.00001f);
}
__device__ __forceinline__ float geometry_schlick(float ndv, float roughness)
{
    const float r = roughness + 1.0f;
    const float k = (r * r) / 8.0f;
    return ndv / (ndv * (1.0f - k) + k);
}

struct Surface {
    V4 albedo, normal;
    float roughness, metal;
    V4 emission;
    V4 position;
};

// winner attributes (:435-548) + material evaluation (:363-377, Material.cu:268-323)
// matHint: the winner's TriShade::matFlags when the caller already has it (the wavefront hit record), so that the
// material fetch does not wait for the triangle record; kNoHit = take it from the record
// full = false: the path ends at this hit (the last iteration of the bounce loop, :172) and the hit is not a camera ray's, so
// only the ids and the emission are used: no tangent frame, no texel (needAlbedo: the ambient term of a camera ray would)
__device__ __forceinline__ Surface shade_surface(const SceneView& s, const Hit& h, uint32_t& objOut, uint32_t& meshOut,
                                                 uint32_t& triOut, uint32_t matHint = kNoHit, bool full = true, bool needAlbedo = true)
{
    const float4* tp = reinterpret_cast<const float4*>(s.triShade + h.gid); // one 128-byte line: positions, then attributes
    const float4 A = tp[0], B = tp[1], C = tp[2];
    const float4* sp = tp + 3;
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    // TriShade: n0 = s0.xyz, n1 = (s0.w, s1.x, s1.y), n2 = (s1.z, s1.w, s2.x),
    //           uv0 = (s2.y, s2.z), uv1 = (s2.w, s3.x), uv2 = (s3.y, s3.z), obj = s3.w, mesh,tri,matFlags = s4.xyz
    const uint32_t matFlags = matHint != kNoHit ? matHint : __float_as_uint(s4.z);
    objOut = __float_as_uint(s3.w); meshOut = __float_as_uint(s4.x); triOut = __float_as_uint(s4.y);
    const bool hasNormals = (matFlags >> 31) != 0;
    // constants now, each map's descriptor where it is sampled: the whole 128-byte record held 32 registers to the end
    const DevMaterial* __restrict__ mp = s.materials + (matFlags & 0x7FFFFFFFu);
    struct { float albedo[4]; float roughness, metal; float emission[3]; float strength; int32_t texNormal, texAlbedo, texRoughness, texMetal; } m;
    {
        const float4* q4 = reinterpret_cast<const float4*>(mp);
        const float4 c0 = q4[0], c1 = q4[1], c2 = q4[2];
        const int2 c3 = *reinterpret_cast<const int2*>(q4 + 3);
        m.albedo[0] = c0.x; m.albedo[1] = c0.y; m.albedo[2] = c0.z; m.albedo[3] = c0.w;
        m.roughness = c1.x; m.metal = c1.y; m.emission[0] = c1.z; m.emission[1] = c1.w; m.emission[2] = c2.x; m.strength = c2.y;
        m.texNormal = __float_as_int(c2.z); m.texAlbedo = __float_as_int(c2.w); m.texRoughness = c3.x; m.texMetal = c3.y;
    }

    const float u0 = s2.y, w0 = s2.z, u1 = s2.w, w1 = s3.x, u2 = s3.y, w2 = s3.z;
    if (!full) {
        Surface o;
        o.normal = o.position = mk4(0.0f, 0.0f, 0.0f, 0.0f);
        o.roughness = o.metal = 0.0f;
        o.albedo = mk4(m.albedo[0], m.albedo[1], m.albedo[2], m.albedo[3]);
        if (needAlbedo && m.texAlbedo >= 0) {
            const float u = u0 * h.b0 + u1 * h.b1 + u2 * h.b2;
            const float v = w0 * h.b0 + w1 * h.b1 + w2 * h.b2;
            o.albedo = tex2d_norm(mp->tex[1], u, v);
        }
        o.emission = mk4(m.emission[0], m.emission[1], m.emission[2], m.strength);
        return o;
    }
    const V4 e1 = mk4(B.x - A.x, B.y - A.y, B.z - A.z, 0.0f);
    const V4 e2 = mk4(C.x - A.x, C.y - A.y, C.z - A.z, 0.0f);
    const float du1 = u1 - u0, dv1 = w1 - w0, du2 = u2 - u0, dv2 = w2 - w0;
    const float f = 1.0f / (du1 * dv2 - du2 * dv1);
    V4 tangent = mk4(f * (dv2 * e1.x - dv1 * e2.x), f * (dv2 * e1.y - dv1 * e2.y), f * (dv2 * e1.z - dv1 * e2.z), 0.0f);
    V4 bitangent = mk4(f * (-du2 * e1.x + du1 * e2.x), f * (-du2 * e1.y + du1 * e2.y), f * (-du2 * e1.z + du1 * e2.z), 0.0f);
    tangent = normalize4(tangent);
    bitangent = normalize4(bitangent);
    V4 normal;
    if (hasNormals) {
        normal = mk4(s0.x * h.b0 + s0.w * h.b1 + s1.z * h.b2, s0.y * h.b0 + s1.x * h.b1 + s1.w * h.b2,
                     s0.z * h.b0 + s1.y * h.b1 + s2.x * h.b2, 0.0f);
        const float corr = dot3(tangent, normal);
        tangent = normalize4(mk4(tangent.x - corr * normal.x, tangent.y - corr * normal.y, tangent.z - corr * normal.z, 0.0f));
        bitangent = normalize4(cross4(normal, tangent));
    } else {
        normal = normalize4(cross4(e2, e1));
    }

    Surface o;
    o.position = mk4(A.x * h.b0 + B.x * h.b1 + C.x * h.b2, A.y * h.b0 + B.y * h.b1 + C.y * h.b2,
                     A.z * h.b0 + B.z * h.b1 + C.z * h.b2, 1.0f);
    const float u = u0 * h.b0 + u1 * h.b1 + u2 * h.b2;
    const float v = w0 * h.b0 + w1 * h.b1 + w2 * h.b2;

    o.albedo = m.texAlbedo < 0 ? mk4(m.albedo[0], m.albedo[1], m.albedo[2], m.albedo[3]) : tex2d_norm(mp->tex[1], u, v);
    V4 nl;
    if (m.texNormal < 0) nl = mk4(0.0f, 0.0f, 1.0f, 0.0f);
    else {
        const V4 n = tex2d_norm(mp->tex[0], u, v);
        nl = mk4(n.x * 2 - 1.0f, n.y * 2 - 1.0f, n.z * 2 - 1.0f, 0.0f);
    }
    // matVecMul(tbn, nl) with tbn columns (T, B, N, (0,0,0,1)), :522-526, :728-743
    const V4 nw = mk4(tangent.x * nl.x + bitangent.x * nl.y + normal.x * nl.z + 0.0f * nl.w,
                      tangent.y * nl.x + bitangent.y * nl.y + normal.y * nl.z + 0.0f * nl.w,
                      tangent.z * nl.x + bitangent.z * nl.y + normal.z * nl.z + 0.0f * nl.w,
                      0.0f * nl.x + 0.0f * nl.y + 0.0f * nl.z + 1.0f * nl.w);
    o.normal = normalize4(nw);
    o.roughness = m.texRoughness < 0 ? m.roughness : tex2d_norm_red(mp->tex[2], u, v);
    o.metal = m.texMetal < 0 ? m.metal : tex2d_norm_red(mp->tex[3], u, v);
    o.emission = mk4(m.emission[0], m.emission[1], m.emission[2], m.strength);
    return o;
}

__device__ __forceinline__ uint32_t seg_hash(uint32_t obj, uint32_t mesh, uint32_t tri, uint32_t sample, uint32_t bounce)
{
    uint32_t h = obj * 0x9E3779B1u;
    h ^= (mesh + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= (tri + 1u) * 0xC2B2AE35u;
    h ^= sample * 0x27D4EB2Fu + bounce * 0x165667B1u + 1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

struct LaneStats {
    uint32_t rays, slow, triBrute, hits;
    BvhCounters bvh;
};

// MODE 1: brute force in reference order
// MODE 2 / 3: BVH with the exact in-order emulation inline for rays inside the tie band
//             (redo launch / rt_trace_rays)
template <int MODE>
__device__ __forceinline__ Hit closest_hit(const SceneView& s, Ray& ray, Stack& stack, LaneStats& st, bool& danger)
{
    st.rays++;
    if (MODE == 1) return trace_brute(s, ray, st.triBrute);
    static_assert(MODE >= 1 && MODE <= 3, "closest_hit: MODE 1 (brute force) or 2 / 3 (BVH, exact)");
    return trace_bvh(s, ray, stack, st.bvh, st.slow, danger);
}

struct PathAcc {
    float ox, oy, oz;   // outgoingLight
    float bx, by, bz;   // betaAccumulation
};

// One iteration of the bounce loop (:172-242) after the closest-hit query.  Returns true when
// the path continues with the updated ray.
// cullObj != kNoHit: the closest-hit query skipped the object-space cull of the winner's object
// (hot kernel); it is evaluated here, its 96-byte fetch riding with the triangle records, and a
// failure is reported through cullFail (the pixel then goes to the redo launch).
__device__ __forceinline__ bool shade_segment(const RenderArgs& a, const Hit& h, Ray& ray, uint32_t& seed, uint32_t sample,
                                              int i, PathAcc& p, LaneStats& st, uint32_t& checksum, uint32_t* primary,
                                              uint32_t cullObj = kNoHit, float cullMax = 0.0f, bool* cullFail = nullptr,
                                              uint32_t matHint = kNoHit)
{
    if (h.gid == kNoHit) {
        checksum += seg_hash(kNoHit, kNoHit, kNoHit, sample, (uint32_t)i);
        const V4 dir = mk4(ray.dx, ray.dy, ray.dz, 0.0f);
        const V4 miss = sky_light(dir, a);
        p.ox += miss.x * miss.w * p.bx;
        p.oy += miss.y * miss.w * p.by;
        p.oz += miss.z * miss.w * p.bz;
        return false;
    }
    uint32_t ho, hm, ht;
#ifndef RT_DIAG_EXACT
    st.hits++;
#endif
    if (cullObj != kNoHit) {
        const ObjCull o = load_obj(a.scene.objects + cullObj);
        float tmin, bfar;
        object_slab(ray, o, tmin, bfar);
        // ... and a hit that lies before its object's box entry by more than the tie band (the object-space slab test of a
        // ray that runs almost inside one of the box's planes can put the entry anywhere): the reference then drops the
        // object as soon as an object earlier in the list has narrowed maxDistance below tmin, however far that hit is from
        // this one (rt_trace.h, trace_bvh): the redo launch replays the whole object list for it.  Within the band the
        // tie-band pass has already settled it (a triangle that lies in a face of its box has t = tmin up to rounding).
        const float rayScale = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.ox), __builtin_fabsf(ray.oy)),
                                               __builtin_fmaxf(__builtin_fabsf(ray.oz), a.scene.sceneMaxAbs));
        if (!object_cull_pass(tmin, bfar, cullMax) || tmin > h.t + band_of(h.t, rayScale, a.scene.bandRel)) *cullFail = true;
    }
    const Surface sf = shade_surface(a.scene, h, ho, hm, ht, matHint, i == 0 || i < a.bounceCount - 1, false);
    checksum += seg_hash(ho, hm, ht, sample, (uint32_t)i);
    if (i == 0 && sample == 0) { primary[0] = ho; primary[1] = hm; primary[2] = ht; }

    if (i == 0) { // ambient term, AO is the constant (1,1,1) on the device path (Material.cu:41-44)
        p.ox += 1.0f * sf.albedo.x * a.aoIntensity;
        p.oy += 1.0f * sf.albedo.y * a.aoIntensity;
        p.oz += 1.0f * sf.albedo.z * a.aoIntensity;
    }
    p.ox += sf.emission.x * sf.emission.w * p.bx;
    p.oy += sf.emission.y * sf.emission.w * p.by;
    p.oz += sf.emission.z * sf.emission.w * p.bz;

    if (!(i < a.bounceCount - 1)) return false;

    const float attenuation = 1.0f;
    const V4 N = sf.normal;
    const V4 D = mk4(ray.dx, ray.dy, ray.dz, 0.0f);
    const V4 diffuseDir = random_cosine_hemi(seed, N);
    const V4 specularDir = reflect4(D, N);
    const V4 L = normalize4(lerp4(specularDir, diffuseDir, sf.roughness));
    const V4 V = mk4(-D.x, -D.y, -D.z, D.w);
    const V4 H = normalize4(mk4(L.x + V.x, L.y + V.y, L.z + V.z, 0.0f));
    const V4 F0 = lerp4(mk4(0.04f, 0.04f, 0.04f, 1.0f), sf.albedo, sf.metal);

    const float NDF = distribution_ggx(N, H, sf.roughness);
    const float ndv = rtd::fmax_f(dot3(N, V), 0.0f);
    const float ndl = rtd::fmax_f(dot3(N, L), 0.0f);
    const float G = geometry_schlick(ndv, sf.roughness) * geometry_schlick(ndl, sf.roughness);
    const float cosT = rtd::fmax_f(dot3(H, V), 0.0f);
    const float pw = rtd::pow5_f(1.0f - rtd::fmin_f(cosT, 1.0f));
    const float Fx = F0.x + (1.0f - F0.x) * pw, Fy = F0.y + (1.0f - F0.y) * pw, Fz = F0.z + (1.0f - F0.z) * pw;

    const float mc = 1.0f - sf.metal;
    const float kDx = (1.0f - Fx) * mc, kDy = (1.0f - Fy) * mc, kDz = (1.0f - Fz) * mc;
    const float denominator = 4.0f * ndv * ndl + 0.0001f;
    const float spx = (NDF * G * Fx) / denominator, spy = (NDF * G * Fy) / denominator, spz = (NDF * G * Fz) / denominator;

    p.bx *= (kDx * sf.albedo.x / kPi + spx) * attenuation * ndl;
    p.by *= (kDy * sf.albedo.y / kPi + spy) * attenuation * ndl;
    p.bz *= (kDz * sf.albedo.z / kPi + spz) * attenuation * ndl;

    const float nudge = 0.0001f;
    ray.ox = sf.position.x + N.x * nudge;
    ray.oy = sf.position.y + N.y * nudge;
    ray.oz = sf.position.z + N.z * nudge;
    ray.dx = L.x; ray.dy = L.y; ray.dz = L.z;
    return true;
}

// setupRay (:140-163)
__device__ __forceinline__ Ray setup_ray(const RenderArgs& a, int x, int y, uint32_t& seed)
{
    Ray ray;
    ray.ox = a.camOrigin[0]; ray.oy = a.camOrigin[1]; ray.oz = a.camOrigin[2];
    ray.maxDist = a.maxDistance;
    V4 d;
    if (a.antiAliasing) {
        const float xf = (float)x + random_value(seed) - 0.5f;
        const float yf = (float)y + random_value(seed) - 0.5f;
        d = tex2d_rays(a.rays, a.width, a.height, xf, yf);
    } else {
        d = tex2d_rays(a.rays, a.width, a.height, (float)x, (float)y);
    }
    d = normalize4(rotate_q(d, a.camRotation));
    ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
    return ray;
}

__device__ __forceinline__ uint32_t sample_seed(const RenderArgs& a, int x, int y, uint32_t sidx)
{
    return (uint32_t)x + (uint32_t)y * (uint32_t)a.width + sidx * 34673804u + a.frameCount * 719393u; // :94
}

// :117-137: divide, clamp, progressive running mean, store
__device__ __forceinline__ void write_pixel(const RenderArgs& a, int x, int y, float r, float g, float b, const uint32_t* primary,
                                            uint32_t checksum)
{
    const float spp = (float)a.sampleCount;
    r /= spp; g /= spp; b /= spp;
    r = rtd::fmin_f(rtd::fmax_f(r, 0.0f), 1.0f);
    g = rtd::fmin_f(rtd::fmax_f(g, 0.0f), 1.0f);
    b = rtd::fmin_f(rtd::fmax_f(b, 0.0f), 1.0f);
    const size_t pix = (size_t)y * (size_t)a.width + (size_t)x;
    if (a.progressiveFrameCount != 0) {
        const float4 prev = a.fb[pix];
        const float n = (float)a.progressiveFrameCount, n1 = (float)(a.progressiveFrameCount + 1u);
        r = (r + prev.x * n) / n1;
        g = (g + prev.y * n) / n1;
        b = (b + prev.z * n) / n1;
    }
    a.fb[pix] = make_float4(r, g, b, 1.0f);
    if (a.ids) {
        if (primary) a.ids[pix] = make_uint4(primary[0], primary[1], primary[2], checksum);
        else reinterpret_cast<uint32_t*>(a.ids + pix)[3] = checksum; // wavefront pipeline: wf_shade wrote the ids
    }
}

// statistics: wave shuffle -> LDS -> one global atomic per workgroup and counter, striped over
// kCounterSlots cache lines (a single hot word saturates at ~88 atomics/us on MI355X and was the
// whole frame time of a sky-only frame before this)
__device__ __forceinline__ void flush_stats(const RenderArgs& a, const LaneStats& st, uint32_t* wgCnt)
{
    const int lane = threadIdx.x & 63;
    uint32_t v0 = st.rays, v1 = st.slow, v2 = st.bvh.nodes, v3 = st.bvh.tris + st.triBrute, v4 = st.hits;
    uint32_t v5 = st.bvh.idleA, v6 = st.bvh.idleB;
#ifdef RT_LOOP_TIMERS
    v1 = st.bvh.diagA; v4 = st.bvh.diagB; // diagnostic build: wave-level loop iterations replace slow-path rays / hits
#endif
#ifdef RT_PHASE_TIMERS
    v3 = st.bvh.diagA; v4 = st.bvh.diagB; // diagnostic build: pool / raygen cycles replace tri tests / hits
#endif
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v5 += __shfl_xor(v5, off); v6 += __shfl_xor(v6, off);
        v0 += __shfl_xor(v0, off);
        v1 += __shfl_xor(v1, off);
        v2 += __shfl_xor(v2, off);
        v3 += __shfl_xor(v3, off);
        v4 += __shfl_xor(v4, off);
    }
    if (threadIdx.x < 8) wgCnt[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) {
        atomicAdd(&wgCnt[0], v0); atomicAdd(&wgCnt[1], v1); atomicAdd(&wgCnt[2], v2);
        atomicAdd(&wgCnt[3], v3); atomicAdd(&wgCnt[4], v4);
        atomicAdd(&wgCnt[5], v5); atomicAdd(&wgCnt[6], v6);
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const uint32_t v = wgCnt[threadIdx.x];
        if (v) atomicAdd(a.counters + (size_t)(blockIdx.x % kCounterSlots) * kCounterStride + threadIdx.x, (unsigned long long)v);
    }
}

// bijective XCD-aware remap: workgroups b, b+8, b+16, ... share an XCD; give each XCD a
// contiguous run of tiles (cdna_hip_programming.md section 5, "XCD swizzle must be bijective")
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, xcd = b & 7u, i = b >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + i;
}

// ------------------------------------------------------------------------------------------------
// Hot kernel: persistent wavefronts, path regeneration, resumable traversal.
//
// A lane owns one PIXEL at a time and walks all of its samples in the reference's summation order
// (z-lane major: s = z, z+4, z+8, ... for z = 0..3; the per-z partial sums are then added in z
// order, GPURaytracer.cu:88-116), so no cross-lane reduction is needed.  The wave runs one loop:
//   (1) lanes whose closest-hit query has finished are shaded (miss -> sky, hit -> material + BRDF
//       + bounce ray); lanes whose path ended start their next sample, or finish their pixel and
//       take a new one from the wave's pool (an 8x8 tile fetched with ONE atomic per 64 pixels
//       from 8 striped counters; the workgroups that share an XCD drain the same band of tiles);
//   (2) every lane that holds a ray (re)enters the traversal loops, which the wave leaves again
//       as soon as fewer than kKeepNum/kKeepDen of the entering lanes are still traversing.
// Traversal therefore always runs with most lanes busy, whatever the path lengths and however
// uneven the rays; in the reference (and in the tile kernel below) a wave pays for its longest
// path and its longest ray while the other lanes sit idle.
#ifndef RT_KEEP_NUM
#define RT_KEEP_NUM 3
#endif
constexpr int kKeepNum = RT_KEEP_NUM, kKeepDen = 8;

__global__ __launch_bounds__(256, 3) void path_kernel(const RenderArgs a)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    __shared__ uint32_t poolSlot[kBlock];
    __shared__ uint32_t wgCnt[8];
    HotStack stack; stack.lds = ldsStack + threadIdx.x; stack.dropped = false;
    const int lane = threadIdx.x & 63;
    uint32_t* slot = poolSlot + (threadIdx.x & ~63);

    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0; st.bvh.idleA = 0; st.bvh.idleB = 0;

    // wave-uniform work pool
    const uint32_t nTiles = (uint32_t)a.tiles8X * (uint32_t)a.tiles8Y;
    const uint32_t part0 = blockIdx.x & 7u;
    uint32_t exhausted = 0;       // bit p: partition p has no tile left
    unsigned long long poolMask = 0;
    int poolX = 0, poolY = 0;
    bool noMore = false;

    // lane state
    bool hasPixel = false, hasPath = false, flagged = false;
    int x = 0, y = 0;
    uint32_t z = 0, iter = 0, seed = 0, checksum = 0;
    uint32_t rays0 = 0, hits0 = 0;  // statistics at pixel start: a flagged pixel is counted by the redo launch instead
    int bounce = 0;
    uint32_t primary[3] = { kNoHit, kNoHit, kNoHit };
    float pr = 0.0f, pg = 0.0f, pb = 0.0f;   // partial sum of the current z lane
    float tr = 0.0f, tg = 0.0f, tb = 0.0f;   // sum over z lanes
    PathAcc acc; acc.ox = acc.oy = acc.oz = 0.0f; acc.bx = acc.by = acc.bz = 1.0f;
    Ray ray; ray.ox = ray.oy = ray.oz = ray.dx = ray.dy = ray.dz = 0.0f; ray.maxDist = 0.0f;
    Trav T; trav_begin(T, a.scene, ray, stack); T.node = kTravEnd;
    uint32_t uNodes = 0, uTris = 0;   // wave-uniform statistics of the traversal bursts
    const uint32_t spp = a.sampleCount;

    for (;;) {
#ifdef RT_PHASE_TIMERS
        const long long tp0 = clock64();
#endif
        // ---- (1) shade the lanes whose closest-hit query is complete
        if (hasPath && !trav_running(T)) {
            bool danger = false;
            const Hit h = trav_finish(T, a.scene, ray, danger);
            if (stack.dropped) { flagged = true; stack.dropped = false; } // stack overflow: the redo launch has the deep stack
            const uint32_t sidx = iter * 4u + z;
            if (danger) { flagged = true; st.slow++; }
            const bool more = shade_segment(a, h, ray, seed, sidx, bounce, acc, st, checksum, primary);
            bounce++;
            if (more) {
                st.rays++;
                trav_begin(T, a.scene, ray, stack);
            } else {
                hasPath = false;
                // exposureCorrection (:690-698), per sample, then :102-104
                pr += 1.0f - rtd::exp_f(-acc.ox * a.exposure);
                pg += 1.0f - rtd::exp_f(-acc.oy * a.exposure);
                pb += 1.0f - rtd::exp_f(-acc.oz * a.exposure);
            }
        }
#ifdef RT_PHASE_TIMERS
        const long long tpa = clock64();
        const int nShade = __popcll(rt_ballot(true));
#endif
        // ---- (1a) lanes whose path ended: next sample of the pixel, or the pixel is finished
        if (hasPixel && !hasPath) {
            iter++;
            if (iter * 4u + z >= spp) {             // this z lane has no sample left (:90)
                tr += pr; tg += pg; tb += pb;       // :112-116, z order
                pr = pg = pb = 0.0f;
                z++; iter = 0;
                if (z >= 4u || z >= spp) {
                    for (; z < 4u; z++) { tr += 0.0f; tg += 0.0f; tb += 0.0f; } // slots never written read as 0
                    if (flagged) {
                        st.rays = rays0; st.hits = hits0;
                        const uint32_t k = atomicAdd(a.redoCount, 1u);
                        a.redoList[k] = (uint32_t)y * (uint32_t)a.width + (uint32_t)x;
                    } else {
                        write_pixel(a, x, y, tr, tg, tb, primary, checksum);
                    }
                    hasPixel = false;
                }
            }
        }
        // ---- (1b) lanes without a pixel take one from the wave's pool
        unsigned long long need = rt_ballot(!hasPixel && !noMore);
        while (need != 0ull) {
            if (poolMask == 0ull) {
                int t = -1;
                if (lane == 0) {
                    for (uint32_t k = 0; k < 8u && t < 0; k++) {
                        const uint32_t p = (part0 + k) & 7u;
                        if (exhausted & (1u << p)) continue;
                        const uint32_t lo = (uint32_t)(((unsigned long long)nTiles * p) >> 3);
                        const uint32_t hi = (uint32_t)(((unsigned long long)nTiles * (p + 1u)) >> 3);
                        const uint32_t i = atomicAdd(a.tileCounters + p * 32u, 1u);
                        if (i < hi - lo) t = (int)(lo + i); else exhausted |= 1u << p;
                    }
                }
                t = __shfl(t, 0);
                if (t < 0) { noMore = true; break; }
                poolX = a.x0 + (t % a.tiles8X) * 8;
                poolY = a.y0 + (t / a.tiles8X) * 8;
                const bool valid = poolX + (lane & 7) < a.x1 && poolY + (lane >> 3) < a.y1;
                poolMask = rt_ballot(valid);
                if (poolMask == 0ull) continue;
            }
            // the k-th needing lane takes the k-th free pixel of the tile
            const unsigned long long below = (1ull << lane) - 1ull;
            const bool isFree = (poolMask >> lane) & 1ull;
            if (isFree) slot[__popcll(poolMask & below)] = (uint32_t)lane;
            const int rank = __popcll(need & below);
            const int avail = __popcll(poolMask);
            const bool want = (need >> lane) & 1ull;
            __builtin_amdgcn_wave_barrier();
            if (want && rank < avail) {
                const uint32_t bit = slot[rank];
                x = poolX + (int)(bit & 7u); y = poolY + (int)(bit >> 3);
                hasPixel = true; flagged = false;
                rays0 = st.rays; hits0 = st.hits;
                z = 0; iter = 0; checksum = 0;
                primary[0] = primary[1] = primary[2] = kNoHit;
                pr = pg = pb = 0.0f; tr = tg = tb = 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned long long taken = rt_ballot(want && rank < avail);
            int k = __popcll(taken); // the bits handed out are the lowest popc(taken) set bits
            unsigned long long m = poolMask;
            while (k-- > 0) m &= m - 1ull;
            poolMask = m;
            need = rt_ballot(!hasPixel && !noMore);
        }
#ifdef RT_PHASE_TIMERS
        const long long tpc = clock64();
#endif
        // ---- (1c) start the next sample
        if (hasPixel && !hasPath) {
            const uint32_t sidx = iter * 4u + z;
            seed = sample_seed(a, x, y, sidx);
            ray = setup_ray(a, x, y, seed);
            // a camera ray next to a listed thin triangle's line (wf_gen says why): the pixel goes to the redo launch
#ifndef RT_DIAG_NO_THIN_GEN
            for (uint32_t k = 0; k < a.scene.nThin; k++)
                if (thin_near(a.scene.thin[k], ray)) flagged = true;
#endif
            acc.ox = acc.oy = acc.oz = 0.0f; acc.bx = acc.by = acc.bz = 1.0f;
            bounce = 0;
            hasPath = a.bounceCount > 0;
            if (hasPath) {
                st.rays++;
                trav_begin(T, a.scene, ray, stack);
            } else { // bounceCount <= 0: the bounce loop does not run, radiance 0
                pr += 1.0f - rtd::exp_f(-0.0f * a.exposure);
                pg += 1.0f - rtd::exp_f(-0.0f * a.exposure);
                pb += 1.0f - rtd::exp_f(-0.0f * a.exposure);
            }
        }
        if (rt_ballot(hasPixel) == 0ull) break; // every lane is idle and no tile is left

        // ---- (2) traversal burst for every lane with an unfinished query
        const bool running = hasPath && trav_running(T);
        const int nRun = __popcll(rt_ballot(running));
#ifdef RT_PHASE_TIMERS
        const long long tp1 = clock64();
#endif
        (void)running;
        trav_burst(T, a.scene, ray, stack, st.bvh, (nRun * kKeepNum + kKeepDen - 1) / kKeepDen, uNodes, uTris);
#ifdef RT_PHASE_TIMERS
        // diagnostic build: wave cycles outside / inside the traversal burst (reported through the idle-lane slots, /64)
        // slots: idleA burst, idleB shade, diagA pool, diagB raygen (all /64); slow = main-loop iterations
        if (lane == 0) {
            st.bvh.idleA += (uint32_t)((clock64() - tp1) >> 6); st.bvh.idleB += (uint32_t)((tpa - tp0) >> 6);
            st.bvh.diagA += (uint32_t)((tpc - tpa) >> 6); st.bvh.diagB += (uint32_t)((tp1 - tpc) >> 6);
            st.slow++;
        }
#endif
    }
    if (lane == 0) { st.bvh.nodes += uNodes; st.bvh.tris += uTris; }
    flush_stats(a, st, wgCnt);
}

// ------------------------------------------------------------------------------------------------
// Tile / list kernel: one thread per (pixel, z lane) as in the reference, nested sample and bounce
// loops.  Used for the brute-force mode (MODE 1, whole tile) and for the redo launch (MODE 2,
// pixel list written by path_kernel).  LOG2L: log2 of the sample lanes per pixel (0, 1, 2).
template <int MODE>
__device__ __forceinline__ V4 trace_path(const RenderArgs& a, Ray& ray, uint32_t& seed, uint32_t sample, Stack& stack,
                                         LaneStats& st, uint32_t& checksum, uint32_t* primary)
{
    PathAcc p; p.ox = p.oy = p.oz = 0.0f; p.bx = p.by = p.bz = 1.0f;
    for (int i = 0; i < a.bounceCount; i++) {
        bool danger = false;
        const Hit h = closest_hit<MODE>(a.scene, ray, stack, st, danger);
        if (!shade_segment(a, h, ray, seed, sample, i, p, st, checksum, primary)) break;
    }
    return mk4(p.ox, p.oy, p.oz, 1.0f);
}

template <int LOG2L, int MODE>
__global__ __launch_bounds__(256) void render_kernel(const RenderArgs a)
{
    constexpr int L = 1 << LOG2L;
    constexpr int TW = (L == 4) ? 4 : 8;   // wave tile in pixels
    constexpr int TH = (L == 1) ? 8 : 4;
    constexpr int BW = 2 * TW, BH = 2 * TH; // workgroup = 2x2 waves
    constexpr int PIX_PER_WG = 256 / L;

    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    __shared__ uint32_t wgCnt[8];
    uint32_t ovf[kStackDepth - kLdsStack];
    Stack stack; stack.lds = ldsStack + threadIdx.x; stack.ovf = ovf;
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0; st.bvh.idleA = 0; st.bvh.idleB = 0;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t z = (uint32_t)(lane & (L - 1));
    const int p = lane >> LOG2L;
    const uint32_t spp = a.sampleCount;
    const bool listMode = a.pixelList != nullptr;
    const uint32_t listCount = listMode ? *a.pixelCount : 0u;

    for (uint32_t grp = blockIdx.x;; grp += gridDim.x) {
        int x, y;
        bool inTile;
        if (listMode) {
            if ((unsigned long long)grp * PIX_PER_WG >= listCount) break;
            const uint32_t idx = grp * PIX_PER_WG + (uint32_t)(threadIdx.x >> LOG2L);
            inTile = idx < listCount;
            const uint32_t pix = inTile ? a.pixelList[idx] : 0u;
            x = (int)(pix % (uint32_t)a.width); y = (int)(pix / (uint32_t)a.width);
        } else {
            const uint32_t nBlocks = (uint32_t)a.tilesX * (uint32_t)a.tilesY;
            const uint32_t tile = xcd_remap(grp, nBlocks);
            const int ty = (int)(tile / (uint32_t)a.tilesX), tx = (int)(tile % (uint32_t)a.tilesX);
            x = a.x0 + tx * BW + (wave & 1) * TW + (p % TW);
            y = a.y0 + ty * BH + (wave >> 1) * TH + (p / TW);
            inTile = x < a.x1 && y < a.y1;
        }

        uint32_t checksum = 0;
        uint32_t primary[3] = { kNoHit, kNoHit, kNoHit };
        float cr = 0.0f, cg = 0.0f, cb = 0.0f; // this lane's partial sum over its samples (:87-105)
        if (inTile && z < spp) {
            const uint32_t iters = (spp + 3u) / 4u; // BLOCK_DEPTH = 4 (GPURaytracer.hpp:19)
            for (uint32_t iter = 0; iter < iters; iter++) {
                const uint32_t sidx = iter * 4u + z;
                if (sidx >= spp) break;
                uint32_t seed = sample_seed(a, x, y, sidx);
                Ray ray = setup_ray(a, x, y, seed);
                const V4 c = trace_path<MODE>(a, ray, seed, sidx, stack, st, checksum, primary);
                // exposureCorrection (:690-698), per sample
                cr += 1.0f - rtd::exp_f(-c.x * a.exposure);
                cg += 1.0f - rtd::exp_f(-c.y * a.exposure);
                cb += 1.0f - rtd::exp_f(-c.z * a.exposure);
            }
        }

        // :107-116 -- lane z == 0 adds the partial sums of z = 0..3 in that order.  Lanes that the
        // reference never writes (z >= spp) contribute 0.
        float r = 0.0f, g = 0.0f, b = 0.0f;
        uint32_t sum = 0;
        const int base = lane & ~(L - 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < L) {
                r += __shfl(cr, base + i);
                g += __shfl(cg, base + i);
                b += __shfl(cb, base + i);
                sum += (uint32_t)__shfl((int)checksum, base + i);
            } else {
                r += 0.0f; g += 0.0f; b += 0.0f;
            }
        }
        if (inTile && z == 0) write_pixel(a, x, y, r, g, b, primary, sum);
        if (!listMode) break;
    }
    flush_stats(a, st, wgCnt);
}

// closest-hit queries for caller-supplied rays (rt_trace_rays)
template <int MODE>
__global__ __launch_bounds__(256) void trace_rays_kernel(SceneView s, const float4* __restrict__ rays, int n,
                                                         uint4* __restrict__ out, unsigned long long* counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 r0 = rays[2 * i], r1 = rays[2 * i + 1];
    Ray ray; ray.ox = r0.x; ray.oy = r0.y; ray.oz = r0.z; ray.maxDist = r0.w; ray.dx = r1.x; ray.dy = r1.y; ray.dz = r1.z;
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    uint32_t ovf[kStackDepth - kLdsStack];
    Stack stack; stack.lds = ldsStack + threadIdx.x; stack.ovf = ovf;
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0; st.bvh.idleA = 0; st.bvh.idleB = 0;
    bool danger = false;
    const Hit h = closest_hit<MODE>(s, ray, stack, st, danger);
    uint4 o = make_uint4(kNoHit, kNoHit, kNoHit, kNoHit);
    if (h.gid != kNoHit) {
        const TriShade* ts = s.triShade + h.gid;
        o = make_uint4(ts->obj, ts->mesh, ts->tri, __float_as_uint(h.t));
    }
    out[i] = o;
    if (st.slow) atomicAdd(counters + 1, (unsigned long long)st.slow);
}

// ------------------------------------------------------------------------------------------------
// Wavefront pipeline.  The persistent path kernel above keeps a lane's whole path in registers, so
// the traversal loops run at the occupancy the shading code's registers allow (168 VGPRs, 3 waves
// per SIMD) and the shading code runs behind the traversal's live state; measured, both halves were
// bound by the latency of dependent fetches at that occupancy.  Here the path state lives in HBM
// (288 GB: ~190 bytes per sample in flight is nothing) and each bounce level is two launches:
//   wf_trace: persistent, closest-hit queries only, lanes refilled from the level's queue;
//   wf_shade: one lane per live path, full waves, material fetches batched.
// The per-sample results are summed per pixel in the reference's order by wf_resolve.
__device__ __forceinline__ uint32_t* wf_count(const WfBufs& w, int level, int stripe)
{
    return w.ctrl + ((size_t)(level * 2 + 0) * kWfStripes + stripe) * kWfCtrlStride;
}
__device__ __forceinline__ uint32_t* wf_cursor(const WfBufs& w, int level, int stripe)
{
    return w.ctrl + ((size_t)(level * 2 + 1) * kWfStripes + stripe) * kWfCtrlStride;
}
// position of entry `off` of a stripe at a level inside the ping-pong arrays (queue, rayO, rayD, accO, accB)
__device__ __forceinline__ size_t wf_pos(const WfBufs& w, int level, uint32_t stripe, uint32_t off)
{
    return ((size_t)(level & 1) * kWfStripes + stripe) * (size_t)w.subCap + off;
}
// ... and inside the single-buffered hit arrays
__device__ __forceinline__ size_t wf_hit_pos(const WfBufs& w, uint32_t stripe, uint32_t off)
{
    return (size_t)stripe * (size_t)w.subCap + off;
}
// origin + maxDistance of a level's ray: the camera rays all start at the camera with the renderer's
// maxDistance (setup_ray), so level 0 keeps no copy (16 bytes per sample less to write, and to read twice)
__device__ __forceinline__ float4 wf_ray_origin(const RenderArgs& a, const WfBufs& w, int level, size_t p)
{
    if (level == 0) return make_float4(a.camOrigin[0], a.camOrigin[1], a.camOrigin[2], a.maxDistance);
    return w.rayO[p];
}
// finished-sample results are stored sample-major inside each 64-pixel tile, so that wf_resolve's
// lanes (one pixel each) read consecutive addresses
__device__ __forceinline__ size_t wf_out_index(uint32_t pix, uint32_t sidx, uint32_t spp)
{
    return ((size_t)(pix >> 6) * spp + sidx) * 64u + (pix & 63u);
}
// slot -> pixel; false when the pixel lies outside the render rectangle (edge tiles)
__device__ __forceinline__ bool wf_pixel(const RenderArgs& a, const WfBufs& w, uint32_t pix, int& x, int& y)
{
    const uint32_t tile = w.tileBegin + (pix >> 6) * w.tileStride, in = pix & 63u;
    const uint32_t row0 = fast_div(tile, a.divTilesX), col = tile - row0 * (uint32_t)a.tiles8X;
    uint32_t row = row0;
    if (a.bandTileRows > 0) { // row counts through this call's bands only: band k of the call is band bandFirst + k * bandStride of the rectangle
        const uint32_t k = fast_div(row, a.divBandRows), in_band = row - k * (uint32_t)a.bandTileRows;
        row = ((uint32_t)a.bandFirst + k * (uint32_t)a.bandStride) * (uint32_t)a.bandTileRows + in_band;
    }
#ifndef RT_TILE_ORDER
#define RT_TILE_ORDER 1
#endif
#if RT_TILE_ORDER
    // Z order inside the tile: consecutive pixels form 2x2, 4x4 ... blocks, so the 64 camera rays that walk the tree together
    // as a packet (16 pixels at 4 spp, 4 at 16) cover a square of the image instead of a strip of a row or two
    const uint32_t px = (in & 1u) | ((in >> 1) & 2u) | ((in >> 2) & 4u), py = ((in >> 1) & 1u) | ((in >> 2) & 2u) | ((in >> 3) & 4u);
#else
    const uint32_t px = in & 7u, py = in >> 3;
#endif
    x = a.x0 + (int)col * 8 + (int)px;
    y = a.y0 + (int)row * 8 + (int)py;
    return x < a.x1 && y < a.y1;
}

// slot of entry `off` of a stripe at level 0: wf_gen lays the slots out piece by piece (piece = 256 consecutive slots,
// piece p in stripe p % kWfStripes), so no queue array is kept for the camera rays; at the levels past them the slot
// travels in the spare word of the beta record
__device__ __forceinline__ uint32_t wf_level0_slot(uint32_t stripe, uint32_t off) { return ((off >> 8) * kWfStripes + stripe) * 256u + (off & 255u); }
__device__ __forceinline__ bool wf_is_hole(const float4& d) { return d.x == 0.0f && d.y == 0.0f && d.z == 0.0f; }
__device__ __forceinline__ int wf_levels(const RenderArgs& a) { return a.bounceCount > 0 ? a.bounceCount : 0; }
// The sample's path is traced again from the camera by the redo launch, with the exact in-order emulation inline
// (tie bands between objects, culled winners, stack overflows, camera rays next to a listed thin triangle); its path in
// the pipeline ends where this is called, so a sample is listed once, under the stripe it lives in (a list cannot overflow).
__device__ __forceinline__ void wf_redo_sample(const RenderArgs& a, const WfBufs& w, uint32_t stripe, uint32_t slot)
{
    const uint32_t k = atomicAdd(w.ctrl + wf_ctrl_redo(wf_levels(a), (int)stripe), 1u);
    w.redoSlots[(size_t)stripe * w.subCap + k] = slot;
}

// level 0: camera rays (:140-163) in slot order; holes for pixels outside the rectangle
__global__ __launch_bounds__(256) void wf_gen_kernel(const RenderArgs a, const WfBufs w)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nPieces = (w.nSlots + 255u) / 256u;
    if (blockIdx.x == 0 && threadIdx.x < kWfStripes) {
        const uint32_t s = threadIdx.x;
        const uint32_t pieces = s < nPieces ? (nPieces - s + kWfStripes - 1) / kWfStripes : 0u;
        *wf_count(w, 0, (int)s) = pieces * 256u;
    }
    const uint32_t piece = blockIdx.x, stripe = piece % kWfStripes, off = (piece / kWfStripes) * 256u + threadIdx.x;
    float4 entry = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // a hole (pixel outside the rectangle, or no bounce loop): the zero direction
    bool traced = false;
    if (i < w.nSlots) {
        const uint32_t spp = a.sampleCount;
        const uint32_t pix = fast_div(i, a.divSpp), sidx = i - pix * spp;
        int x, y;
        if (wf_pixel(a, w, pix, x, y)) {
            uint32_t seed = sample_seed(a, x, y, sidx);
            const Ray ray = setup_ray(a, x, y, seed);
            if (a.bounceCount > 0) { // outgoing light 0, beta 1, checksum 0 and the slot (wf_level0_slot): wf_shade(0) supplies them
                entry = make_float4(ray.dx, ray.dy, ray.dz, __uint_as_float(seed));
                traced = true;
            } else { // the bounce loop does not run: radiance 0 (:172)
                if (sidx == 0 && a.ids) {
                    uint32_t* idp = reinterpret_cast<uint32_t*>(a.ids + ((size_t)y * (size_t)a.width + (size_t)x));
                    idp[0] = kNoHit; idp[1] = kNoHit; idp[2] = kNoHit;
                }
                w.out[wf_out_index(pix, sidx, spp)] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
                w.outRays[wf_out_index(pix, sidx, spp)] = 0;
            }
        }
    }
    // The triangles no box can stand for (rt_commit.hip, thin_kernel; rt_trace.h, thin_near): a ray that passes within rounding
    // of the LINE such a triangle lies on can "hit" it at a distance outside what the sliver's box holds, so a walk of the tree can miss it.  Camera
    // rays are the rays that can be aimed at such a line by construction (an axis-aligned camera without anti-aliasing over
    // gridded geometry puts pixel centres on vertices), so every camera ray is tested against every listed line here, and one
    // that comes near goes to the redo launch, whose walks test the listed triangles (bvh_walk).  With the origin common to all
    // camera rays the test of thin_near, |w . (d x u)| <= rad |d x u|, becomes |d . (u x w)| <= rad (as |d x u| <= 1: a superset,
    // the plane through the camera and the line instead of the line; 1.25 covers the rounding of the other association), with
    // u x w and rad worked out once per workgroup and line: seven operations per ray and line.
#ifndef RT_DIAG_NO_THIN_GEN   // diagnostic build: shows that tests/thin_scene.py fails without this test
    if (a.scene.nThin != 0u && a.bounceCount > 0) {
        __shared__ float4 thinPlane[64];
        bool near = false;
        for (uint32_t base = 0; base < a.scene.nThin; base += 64u) {
            __syncthreads();
            if (threadIdx.x < 64u && base + threadIdx.x < a.scene.nThin) {
                const ThinTri r = a.scene.thin[base + threadIdx.x];
                const float wx = r.px - a.camOrigin[0], wy = r.py - a.camOrigin[1], wz = r.pz - a.camOrigin[2];
                const float rad = 1.25f * (r.k * (wx * wx + wy * wy + wz * wz + r.len2) + 1e-30f);
                thinPlane[threadIdx.x] = make_float4(r.uy * wz - r.uz * wy, r.uz * wx - r.ux * wz, r.ux * wy - r.uy * wx, rad);
            }
            __syncthreads();
            const uint32_t cnt = a.scene.nThin - base < 64u ? a.scene.nThin - base : 64u;
            if (traced)
                for (uint32_t k = 0; k < cnt; k++) {
                    const float4 pl = thinPlane[k];
                    const float num = entry.x * pl.x + entry.y * pl.y + entry.z * pl.z;
                    near = near || !(num * num > pl.w * pl.w);   // NaN stays in
                }
        }
        if (near) {
            wf_redo_sample(a, w, stripe, i);
            entry = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // a hole for the pipeline
        }
    }
#endif
    w.rayD[wf_pos(w, 0, stripe, off)] = entry;
}

#ifndef RT_WF_OCC
#define RT_WF_OCC 4
#endif
// closest-hit queries of one level: persistent waves, each lane refilled from the queue as soon as
// the wave's share of finished lanes makes it worth leaving the traversal loops
__global__ __launch_bounds__(256, RT_WF_OCC) void wf_trace_kernel(const RenderArgs a, const WfBufs w, const int level, const int levels)
{
    __builtin_assume(level >= 1);   // the camera rays go through the packet kernel (launch_render_wf)
    __shared__ uint32_t ldsStack[kWfStack * kBlock];
    __shared__ float4 poolO[kBlock], poolD[kBlock];   // the block of rays being handed out (`cur` below), one entry per lane
    __shared__ uint32_t wgCnt[8];
    HotStackT<kWfStack> stack; stack.lds = ldsStack + threadIdx.x; stack.dropped = false;
    const int lane = threadIdx.x & 63;
    float4* curO = poolO + (threadIdx.x & ~63);
    float4* curD = poolD + (threadIdx.x & ~63);
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;

    // Work arrives in blocks of up to 64 consecutive queue entries of one stripe, claimed with one
    // atomic.  Three blocks are in flight so that a refill never waits on memory: `cur` (in LDS: eight
    // registers less across the traversal loops, and a lane fetches its ray with two reads where it took seven
    // cross-lane moves), `nxt` (in registers, its loads issued at the previous promotion) and the pending
    // claim (its atomic issued then too).  Measured before this: a third of the wave time of the
    // level's launch went into the dependent chain atomic -> queue entry -> ray at refill time.
    const uint32_t waveId = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t stripe = waveId % kWfStripes, stripeCnt = 0;   // wave-uniform
    unsigned long long exhausted = 0ull;
    uint32_t poolStart = 0, poolEnd = 0;   // the block's entries not yet handed out: they go in order (no holes past the camera rays), so a range says it all
    bool pendIssued = false, nxtHas = false, noMore = false, have = false;
    uint32_t pendC = 0;                 // lane 0: cursor value returned by the pending atomic
    uint32_t cBase = 0, nBase = 0, nCount = 0;   // wave-uniform: hit-array position of a block's first entry, entries in the block (no holes past the camera rays)
    float4 nO = make_float4(0, 0, 0, 0), nD = nO;
    uint32_t hitPos = 0;
    Ray ray; ray.ox = ray.oy = ray.oz = ray.dx = ray.dy = ray.dz = 0.0f; ray.maxDist = 0.0f;
    Trav T; trav_begin(T, a.scene, ray, stack); T.node = kTravEnd;
    uint32_t uNodes = 0, uTris = 0, uSlow = 0;   // wave-uniform statistics (scalar registers)

    // picks the next stripe that still has entries (wave-uniform); false when none is left.  The count
    // of the stripe in use stays cached: re-reading it would put a dependent fetch into every refill.
    bool stripeOk = false;
    auto pick_stripe = [&]() -> bool {
        if (stripeOk && !(exhausted & (1ull << stripe))) return true;
        for (int k = 0; k < kWfStripes; k++) {
            const uint32_t sidx = (stripe + (uint32_t)k) % kWfStripes;
            if (exhausted & (1ull << sidx)) continue;
            const uint32_t cnt = *wf_count(w, level, (int)sidx);
            if (cnt == 0u) { exhausted |= 1ull << sidx; continue; }
            stripe = sidx; stripeCnt = cnt; stripeOk = true;
            return true;
        }
        stripeOk = false;
        return false;
    };
    auto issue_claim = [&]() {
        pendIssued = pick_stripe();
        if (pendIssued && lane == 0) pendC = atomicAdd(wf_cursor(w, level, (int)stripe), 64u);
    };
    // resolves the pending claim into (offset within the stripe, entries); claims again elsewhere when
    // the stripe ran out meanwhile
    auto resolve_claim = [&](uint32_t& off, uint32_t& n) -> bool {
        while (pendIssued) {
            const uint32_t c = (uint32_t)__shfl((int)pendC, 0);
            if (c < stripeCnt) { off = c; n = stripeCnt - c < 64u ? stripeCnt - c : 64u; return true; }
            exhausted |= 1ull << stripe;
            issue_claim();
        }
        return false;
    };
    auto load_next = [&]() {
        uint32_t off = 0, n = 0;
        nxtHas = resolve_claim(off, n);
        if (nxtHas) {
            const size_t p = wf_pos(w, level, stripe, off + (uint32_t)lane);
            const bool in = (uint32_t)lane < n;
            nCount = n;
            if (in) { nO = wf_ray_origin(a, w, level, p); nD = w.rayD[p]; }
            nBase = (uint32_t)wf_hit_pos(w, stripe, off);
            issue_claim();
        }
    };
    issue_claim();
    load_next();

    for (;;) {
#ifdef RT_PHASE_TIMERS
        const long long tq0 = clock64();
#endif
        // ---- results of the lanes that finished
        bool inBand = false;
        if (have && !trav_running(T)) {
            // A hit of another object inside the tie band (1e-5 of the rays on C2, 2e-3 on C4, 7e-4 on C5): the ray is listed
            // for wf_exact, which traces it again with the reference's in-order semantics between this launch and the level's
            // shade launch (RT_EXACT_PASS = 1, the default).  A dropped stack entry -- and, in a build with RT_EXACT_PASS = 0, the
            // tie band too -- marks the hit record with kWfDanger: wf_shade then sends the SAMPLE to the redo launch, which
            // traces its whole path again with the exact emulation inline and a full-depth stack.
            bool band = false;
            const Hit h = trav_result(T, a.scene, ray, band);
            const bool over = stack.dropped;
            stack.dropped = false;
            inBand = band;
#if RT_EXACT_PASS
            if (band) { // listed under the stripe the ray sits in: that list cannot overflow
                const uint32_t bs = hitPos / w.subCap;
                w.exactList[(size_t)bs * w.subCap + atomicAdd(w.ctrl + wf_ctrl_exact(levels, level, (int)bs), 1u)] = hitPos;
                band = false;
            }
#endif
            if (h.gid != kNoHit) w.hitA[hitPos] = make_float4(h.t, h.b0, h.b1, h.b2); // a miss has nothing to say here: 16 bytes less per sky ray
            w.hitM[hitPos] = make_uint2((h.gid == kNoHit ? kWfMiss : h.gid) | (over || band ? kWfDanger : 0u), T.bestObj);
            have = false;
        }
        uSlow += (uint32_t)__popcll(rt_ballot(inBand));
#ifdef RT_PHASE_TIMERS
        const long long tq1 = clock64();
#endif
        // ---- refill from the pool
        unsigned long long need = rt_ballot(!have && !noMore);
        bool fresh = false;   // the lane took a ray: its traversal is set up once, behind the loop (inside it the whole traversal
                              // state would have to be merged at every turn: ninety register copies per refill)
        while (need != 0ull) {
            if (poolStart == poolEnd) {
                if (!nxtHas) { noMore = true; break; }
                curO[lane] = nO; curD[lane] = nD; cBase = nBase;
                poolStart = 0u; poolEnd = nCount;
                load_next();
                if (poolEnd == 0u) continue;
            }
            // the k-th needing lane takes the k-th entry of what is left of the block
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t avail = poolEnd - poolStart;
            const bool take = ((need >> lane) & 1ull) && rank < avail;
            __builtin_amdgcn_wave_barrier();
            if (take) {
                const uint32_t src = poolStart + rank;
                const float4 o = curO[src], d = curD[src];
                ray.ox = o.x; ray.oy = o.y; ray.oz = o.z; ray.maxDist = o.w; ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
                hitPos = cBase + src;
                have = true; fresh = true;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t wanted = (uint32_t)__popcll(need);
            poolStart += wanted < avail ? wanted : avail;
            need = rt_ballot(!have && !noMore);
        }
        if (fresh) trav_begin(T, a.scene, ray, stack);
        if (rt_ballot(have) == 0ull) break;
        // ---- traversal burst
        const bool running = have && trav_running(T);
        const int nRun = __popcll(rt_ballot(running));
#ifdef RT_PHASE_TIMERS
        const long long tq2 = clock64();
#endif
        trav_burst(T, a.scene, ray, stack, st.bvh, noMore ? 1 : (nRun * kKeepNum + kKeepDen - 1) / kKeepDen, uNodes, uTris);
#ifdef RT_PHASE_TIMERS
        if (lane == 0) {
            st.bvh.idleA += (uint32_t)((clock64() - tq2) >> 6); st.bvh.idleB += (uint32_t)((tq1 - tq0) >> 6);
            st.bvh.diagA += (uint32_t)((tq2 - tq1) >> 6); st.slow++; st.bvh.diagB += (uint32_t)nRun;
        }
#endif
    }
    if (lane == 0) { st.bvh.nodes += uNodes; st.bvh.tris += uTris; st.slow += uSlow; }
    flush_stats(a, st, wgCnt);
}

// Closest-hit queries of level 0 as packets: the 64 camera rays of a block (neighbouring pixels, all their
// samples) walk the tree together.  Node, stack and child order are wave-uniform -- the node arrives
// through scalar loads instead of 64 divergent fetches, the stack is one VGPR indexed by lane
// (v_writelane / v_readlane), there is no per-lane scheduling -- while every lane tests the boxes and
// triangles against its own ray and keeps its own closest hit, so the result per ray is what wf_trace
// gives: a node is entered when any lane's ray enters it, and a lane that does not merely tests
// triangles its ray cannot hit.  Only worth it for coherent rays: after the first bounce the union of
// 64 rays' nodes is most of the tree.
#ifndef RT_PK_OCC
#define RT_PK_OCC 6
#endif
__global__ __launch_bounds__(256, RT_PK_OCC) void wf_trace_packet_kernel(const RenderArgs a, const WfBufs w, const int level, const int levels)
{
    __builtin_assume(level == 0);   // launched for the camera rays only (launch_render_wf)
    __shared__ uint32_t wgCnt[8];
    const int lane = threadIdx.x & 63;
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;
    const uint32_t waveId = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t stripe = waveId % kWfStripes, stripeCnt = 0;
    unsigned long long exhausted = 0ull;
    bool stripeOk = false, pendIssued = false;
    uint32_t pendC = 0;
    uint32_t uNodeTests = 0, uNodeFetches = 0, uTris = 0;   // wave-uniform statistics (scalar registers): no vector work per step
    auto pick_stripe = [&]() -> bool {
        if (stripeOk && !(exhausted & (1ull << stripe))) return true;
        for (int k = 0; k < kWfStripes; k++) {
            const uint32_t sidx = (stripe + (uint32_t)k) % kWfStripes;
            if (exhausted & (1ull << sidx)) continue;
            const uint32_t cnt = *wf_count(w, level, (int)sidx);
            if (cnt == 0u) { exhausted |= 1ull << sidx; continue; }
            stripe = sidx; stripeCnt = cnt; stripeOk = true;
            return true;
        }
        stripeOk = false;
        return false;
    };
    auto issue_claim = [&]() {
        pendIssued = pick_stripe();
        if (pendIssued && lane == 0) pendC = atomicAdd(wf_cursor(w, level, (int)stripe), 64u);
    };
    issue_claim();
    for (;;) {
        // ---- the block claimed last time round; the next claim's atomic goes out before the walk
        uint32_t off = 0, n = 0, blockStripe = 0;
        bool got = false;
        while (pendIssued) {
            const uint32_t c = (uint32_t)__shfl((int)pendC, 0);
            if (c < stripeCnt) { off = c; n = stripeCnt - c < 64u ? stripeCnt - c : 64u; blockStripe = stripe; got = true; break; }
            exhausted |= 1ull << stripe;
            issue_claim();
        }
        if (!got) break;
#ifdef RT_PHASE_TIMERS
        const long long tk0 = clock64();
#endif
        issue_claim();
        const size_t p = wf_pos(w, level, blockStripe, off + (uint32_t)lane);
        const float4 d = (uint32_t)lane < n ? w.rayD[p] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const bool active = !wf_is_hole(d);
        Ray ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = 1.0f; ray.dy = ray.dz = 0.0f; ray.maxDist = -1.0f;
        if (active) {
            const float4 o = wf_ray_origin(a, w, level, p);
            ray.ox = o.x; ray.oy = o.y; ray.oz = o.z; ray.maxDist = o.w; ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
        }
        Trav T; trav_init(T, a.scene, ray); // a lane without a ray has bound -1: it enters no box and accepts no hit
        const unsigned long long activeMask = rt_ballot(active);
        if (activeMask == 0ull) continue;
        const uint32_t nActive = (uint32_t)__popcll(activeMask);
        bool overflow = false;
#ifdef RT_PHASE_TIMERS
        const long long tk1 = clock64();
        int dbgNodes = 0, dbgLeaves = 0, dbgSp = 0;
#endif
        if (a.scene.rootIsValid) {
            uint32_t stk = 0;      // the wave's stack: entry i lives in lane i
            int sp = 0;            // wave-uniform
            int32_t node = 0;      // wave-uniform
            for (;;) {
                if (node >= 0) {
                    const float4* np = reinterpret_cast<const float4*>(a.scene.nodes + node);
                    const float4 N0 = np[0];
                    const uint4 N1 = reinterpret_cast<const uint4*>(np)[1], N2 = reinterpret_cast<const uint4*>(np)[2];
                    const int4 CH = reinterpret_cast<const int4*>(np)[3];
                    uint32_t k[4];
                    bool kHit[4];
                    node_test<false>(T.br, N0, N1, N2, T.bound, k, kHit);   // k: entry distances, valid where kHit
                    uNodeTests += nActive; uNodeFetches++;
#ifdef RT_PHASE_TIMERS
                    dbgNodes++; dbgSp = sp > dbgSp ? sp : dbgSp;
#endif
                    // children some lane enters, ordered by the entry distances of the first lane that enters any
                    const unsigned long long h0 = rt_ballot(kHit[0]), h1 = rt_ballot(kHit[1]), h2 = rt_ballot(kHit[2]), h3 = rt_ballot(kHit[3]);
                    const unsigned long long any = h0 | h1 | h2 | h3;
                    int h = 0;
                    if (any != 0ull) {
                        const int rep = __builtin_ctzll(any);
                        uint32_t ks[4]; int32_t cs[4] = { CH.x, CH.y, CH.z, CH.w };
                        const unsigned long long hm[4] = { h0, h1, h2, h3 };
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const uint32_t kr = (uint32_t)__builtin_amdgcn_readlane((int)k[c], rep);
                            ks[c] = hm[c] == 0ull ? kMissKey : (((hm[c] >> rep) & 1ull) == 0ull ? 0xFFFFFFFEu : kr);   // some lane enters, the ordering lane does not: last of the entered
                            h += hm[c] != 0ull;
                            cs[c] = __builtin_amdgcn_readfirstlane(cs[c]);
                        }
                        cmp_swap(ks[0], ks[1], cs[0], cs[1]); cmp_swap(ks[2], ks[3], cs[2], cs[3]);
                        cmp_swap(ks[0], ks[2], cs[0], cs[2]); cmp_swap(ks[1], ks[3], cs[1], cs[3]);
                        cmp_swap(ks[1], ks[2], cs[1], cs[2]);
                        if (sp + 3 > 64) overflow = true;
                        else {
                            if (h >= 4) { stk = lane == sp ? (uint32_t)cs[3] : stk; sp++; }
                            if (h >= 3) { stk = lane == sp ? (uint32_t)cs[2] : stk; sp++; }
                            if (h >= 2) { stk = lane == sp ? (uint32_t)cs[1] : stk; sp++; }
                        }
                        node = cs[0];
                    }
                    if (h == 0) {
                        if (sp == 0) break;
                        node = __builtin_amdgcn_readlane((int)stk, --sp);
                    }
                } else {
                    const uint32_t code = (uint32_t)(~node);
                    const uint32_t first = code >> 3, count = code & 7u;
#ifdef RT_PHASE_TIMERS
                    dbgLeaves++;
#endif
                    for (uint32_t q = 0; q < count; q++) {
                        const float4* tp = reinterpret_cast<const float4*>(a.scene.triLeaf + T.triBase + first + q);
                        const float4 A = tp[0], B = tp[1], C = tp[2];
                        uTris += nActive;
                        trav_test_triangle(T, a.scene, A, B, C);
                    }
                    if (sp == 0) break;
                    node = __builtin_amdgcn_readlane((int)stk, --sp);
                }
            }
        }
#ifdef RT_PHASE_TIMERS
        const long long tk2 = clock64();
#endif
        if (active) {
            bool band = false;
            const Hit hres = trav_result(T, a.scene, ray, band);
            const uint32_t hitPos = (uint32_t)wf_hit_pos(w, blockStripe, off + (uint32_t)lane);
            if (band) st.slow++;
#if RT_EXACT_PASS
            if (band) {
                w.exactList[(size_t)blockStripe * w.subCap + atomicAdd(w.ctrl + wf_ctrl_exact(levels, level, (int)blockStripe), 1u)] = hitPos;
                band = false;
            }
#endif
            if (hres.gid != kNoHit) w.hitA[hitPos] = make_float4(hres.t, hres.b0, hres.b1, hres.b2);
            w.hitM[hitPos] = make_uint2((hres.gid == kNoHit ? kWfMiss : hres.gid) | (overflow || band ? kWfDanger : 0u), T.bestObj);
        }
#ifdef RT_PHASE_TIMERS
        if (lane == 0) { // slots as in wf_trace: burst = the walk, write = results, refill = claim + ray fetch + set-up
            st.bvh.idleA += (uint32_t)((tk2 - tk1) >> 6); st.bvh.idleB += (uint32_t)((clock64() - tk2) >> 6);
            st.bvh.diagA += (uint32_t)((tk1 - tk0) >> 6); st.slow++;
            if (tk2 - tk1 > (1ll << 22)) {
                int px, py; wf_pixel(a, w, wf_level0_slot(blockStripe, off) / a.sampleCount, px, py);
                printf("long block: pixel %d %d  cycles %lld  node steps %d leaf steps %d  max sp %d\n", px, py, tk2 - tk1, dbgNodes, dbgLeaves, dbgSp);
            }
            if (tk2 - tk1 > (1ll << 20)) st.bvh.diagB += tk2 - tk1 > (1ll << 24) ? (1u << 20) : (tk2 - tk1 > (1ll << 22) ? (1u << 10) : 1u);
        }
#endif
    }
    if (lane == 0) {
        st.bvh.nodes += uNodeTests; st.bvh.tris += uTris;
#if !defined(RT_PHASE_TIMERS) && !defined(RT_LOOP_TIMERS) && !defined(RT_UTIL_COUNTERS)
        st.bvh.idleA += uNodeTests; st.bvh.idleB += uNodeFetches;   // rt_stats.packet_node_tests / packet_node_fetches
#endif
    }
    flush_stats(a, st, wgCnt);
}

#if RT_EXACT_PASS
// The rays of a level whose closest hit depends on the reference's object order (another object's hit
// inside the tie band): traced again with the exact in-order emulation inline, one
// lane per ray, between the level's trace and shade launches.  Exits at once when the list is empty.
__global__ __launch_bounds__(256) void wf_exact_kernel(const RenderArgs a, const WfBufs w, const int level, const int levels)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    __shared__ uint32_t wgCnt[8];
    uint32_t ovf[kStackDepth - kLdsStack];
    Stack stack; stack.lds = ldsStack + threadIdx.x; stack.ovf = ovf;
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;
    // the stripes' lists read as one: entry i belongs to the last stripe whose running total is <= i
    __shared__ uint32_t listBase[kWfStripes + 1];
    if (threadIdx.x < 64) {
        const uint32_t cnt = w.ctrl[wf_ctrl_exact(levels, level, (int)threadIdx.x)];
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        listBase[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) listBase[0] = 0u;
    }
    __syncthreads();
    const uint32_t n = listBase[kWfStripes];
    const size_t half = (size_t)(level & 1) * kWfStripes * (size_t)w.subCap;
    // The rays are few (1e-3 of a level on C4) and each walks on its own, so a wave runs as long as its slowest lane and
    // mixes node and triangle phases on the way: they are spread over all the waves of the launch, `per` lanes of each.
    const uint32_t waves = gridDim.x * 4u, wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    uint32_t per = (n + waves - 1u) / waves;
    per = per < 1u ? 1u : (per > 64u ? 64u : per);
    for (uint32_t base = wave * per; base < n; base += waves * per) {
        const uint32_t i = base + lane;
        if (lane >= per || i >= n) continue;
        uint32_t ls = 0;
#pragma unroll
        for (uint32_t step = kWfStripes / 2; step > 0; step >>= 1)
            if (listBase[ls + step] <= i) ls += step;
        const uint32_t hp = w.exactList[(size_t)ls * w.subCap + (i - listBase[ls])];
        const float4 o = wf_ray_origin(a, w, level, half + hp), d = w.rayD[half + hp];
        Ray ray; ray.ox = o.x; ray.oy = o.y; ray.oz = o.z; ray.maxDist = o.w; ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
        Hit h;
        if (!trace_band_table(a.scene, ray, w.hitA[hp].x, stack, st.bvh, h)) {
            bool unused = false;
            h = closest_hit<3>(a.scene, ray, stack, st, unused);
#ifdef RT_DIAG_EXACT
            st.hits++; // diagnostic build: table fallbacks are reported as shaded hits
#endif
        }
        w.hitA[hp] = make_float4(h.t, h.b0, h.b1, h.b2);
        const uint32_t hobj = h.gid != kNoHit ? a.scene.triRef[h.gid].obj : kNoHit;
        w.hitM[hp] = make_uint2((h.gid == kNoHit ? kWfMiss : h.gid) | (w.hitM[hp].x & kWfDanger), hobj);
    }
    st.rays = 0; st.slow = 0; // queries are counted per finished sample, tie-band rays by wf_trace
#ifdef RT_DIAG_EXACT
    st.bvh.nodes = 0; st.bvh.tris = 0;
#endif
    flush_stats(a, st, wgCnt);
}
#endif

// material evaluation, BRDF and bounce ray of one level (:172-242): one lane per live path.
//
// Levels >= 1 (DEAL): a quarter of a bounce level's entries are hits and the surface code is several times the sky
// code, so waves of entries in queue order ran it with 18 of 64 lanes.  A workgroup therefore takes a BATCH of
// kShadeBatch x 256 consecutive entries of a stripe, classifies them in one pass and deals them out -- hits first,
// in queue order -- into a list in LDS; its waves then work through that list 64 entries at a time, each at its own
// pace: all but one of the 64-entry runs are all-hit or all-miss, and no barrier ties a sky wave to a surface wave's
// chain of dependent fetches (with one barrier pair per 256 entries the sky waves spent their time waiting for the
// surface wave of their workgroup, and dealing bought nothing).
// The hit record names the object (WfBufs::hitM), so a hit lane's triangle and object-cull fetches go out together.
#ifndef RT_SHADE_OCC
#define RT_SHADE_OCC 4
#endif
#ifndef RT_SHADE_BATCH
#define RT_SHADE_BATCH 16
#endif
constexpr uint32_t kShadeBatch = RT_SHADE_BATCH;   // pieces of 256 entries per batch

// one entry of a level: reads its path state, shades, writes the finished sample or returns the survivor
struct ShadeOut {
    bool more;
    uint32_t slot, seed, chk;
    Ray ray;
    PathAcc acc;
};
__device__ __forceinline__ void shade_entry(const RenderArgs& a, const WfBufs& w, const int level, const uint32_t stripe,
                                            const uint32_t off, const bool mine, LaneStats& st, ShadeOut& o)
{
    const size_t pin = wf_pos(w, level, stripe, off), ph = wf_hit_pos(w, stripe, off);
    o.more = false;
    o.seed = o.chk = 0u;
    o.ray.ox = o.ray.oy = o.ray.oz = o.ray.dx = o.ray.dy = o.ray.dz = o.ray.maxDist = 0.0f;
    o.acc.ox = o.acc.oy = o.acc.oz = o.acc.bx = o.acc.by = o.acc.bz = 0.0f;
    o.slot = kWfHole;
    if (!mine) return;
    const float4 d = w.rayD[pin];
    if (level == 0 && wf_is_hole(d)) return;
    const uint2 hM = w.hitM[ph];
    const float4 aO = level == 0 ? make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u)) : w.accO[pin];
    const float4 aB = level == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : w.accB[pin];
    o.slot = level == 0 ? wf_level0_slot(stripe, off) : __float_as_uint(aB.w);
    const uint32_t spp = a.sampleCount;
    const uint32_t pix = fast_div(o.slot, a.divSpp), sidx = o.slot - pix * spp;
    Ray ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.maxDist = 0.0f; ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
    uint32_t seed = __float_as_uint(d.w), checksum = __float_as_uint(aO.w);
    PathAcc acc; acc.ox = aO.x; acc.oy = aO.y; acc.oz = aO.z; acc.bx = aB.x; acc.by = aB.y; acc.bz = aB.z;
    Hit h; h.t = 0.0f; h.b0 = h.b1 = h.b2 = 0.0f;
    h.gid = (hM.x & kWfMiss) == kWfMiss ? kNoHit : (hM.x & kWfMiss);
    bool danger = (hM.x & kWfDanger) != 0u;
    if (danger) { // tie band / stack overflow: the redo launch traces the whole path again, nothing of this one is kept
#ifdef RT_DEBUG_FLAGS
        printf("redo: level %d slot %u (trace flag) gid %u\n", level, o.slot, hM.x & kWfMiss);
#endif
        wf_redo_sample(a, w, stripe, o.slot);
        return;
    }
    uint32_t cullObj = kNoHit;
    float m0 = 0.0f;
    if (h.gid != kNoHit) { // the ray's origin and the hit's distance / barycentrics matter to a hit only
        const float4 ro = wf_ray_origin(a, w, level, pin), hA = w.hitA[ph];
        ray.ox = ro.x; ray.oy = ro.y; ray.oz = ro.z; m0 = ro.w;
        h.t = hA.x; h.b0 = hA.y; h.b1 = hA.z; h.b2 = hA.w;
        ray.maxDist = h.t;
        cullObj = hM.y;
    }
    uint32_t primary[3] = { kNoHit, kNoHit, kNoHit };
    o.more = shade_segment(a, h, ray, seed, sidx, level, acc, st, checksum, primary, cullObj, m0, &danger);
    if (danger) { // the winner's object fails its cull, or enters its box past the tie band (shade_segment)
#ifdef RT_DEBUG_FLAGS
        printf("redo: level %d slot %u (cull) gid %u\n", level, o.slot, h.gid);
#endif
        wf_redo_sample(a, w, stripe, o.slot);
        o.more = false;
        return;
    }
    if (level == 0 && sidx == 0 && a.ids) {
        int x, y;
        wf_pixel(a, w, pix, x, y);
        uint32_t* idp = reinterpret_cast<uint32_t*>(a.ids + ((size_t)y * (size_t)a.width + (size_t)x));
        idp[0] = primary[0]; idp[1] = primary[1]; idp[2] = primary[2];
    }
    if (o.more) { // kept in registers until the survivors' positions are known
        o.ray = ray; o.seed = seed; o.chk = checksum; o.acc = acc;
        // ... and what the sample is worth if the bounce ray misses everything (the next iteration's :176-183, its inputs all
        // here) is stored now, as the sample's result so far: three quarters of a bounce level's rays do miss, and a miss then
        // has nothing left to do -- the level's shade launch never reads its path state (48 bytes a ray) and never runs a
        // wave for it.  A ray that hits overwrites the result when its own path ends (or with its own next what-if).
        const V4 miss = sky_light(mk4(ray.dx, ray.dy, ray.dz, 0.0f), a);
        float sx = acc.ox, sy = acc.oy, sz = acc.oz;
        sx += miss.x * miss.w * acc.bx;
        sy += miss.y * miss.w * acc.by;
        sz += miss.z * miss.w * acc.bz;
        const size_t oi = wf_out_index(pix, sidx, spp);
        w.out[oi] = make_float4(sx, sy, sz, __uint_as_float(checksum + seg_hash(kNoHit, kNoHit, kNoHit, sidx, (uint32_t)(level + 1))));
        w.outRays[oi] = (uint8_t)(level + 2 > 255 ? 255 : level + 2);
    } else
    { // exposureCorrection (:690-698), per sample
        const size_t oi = wf_out_index(pix, sidx, spp);
        w.out[oi] = make_float4(acc.ox, acc.oy, acc.oz, __uint_as_float(checksum));
        w.outRays[oi] = (uint8_t)(level + 1 > 255 ? 255 : level + 1);
    }
}

// survivors go to the same stripe of the next level's queue: one atomic per wave
__device__ __forceinline__ void shade_append(const WfBufs& w, const int level, const uint32_t stripe, const ShadeOut& o)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long mm = rt_ballot(o.more);
    if (mm == 0ull) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(wf_count(w, level + 1, (int)stripe), (uint32_t)__popcll(mm));
    base = __shfl(base, 0);
    if (o.more) {
        const size_t po = wf_pos(w, level + 1, stripe, base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull)));
        w.rayO[po] = make_float4(o.ray.ox, o.ray.oy, o.ray.oz, o.ray.maxDist);
        w.rayD[po] = make_float4(o.ray.dx, o.ray.dy, o.ray.dz, __uint_as_float(o.seed));
        w.accO[po] = make_float4(o.acc.ox, o.acc.oy, o.acc.oz, __uint_as_float(o.chk));
        w.accB[po] = make_float4(o.acc.bx, o.acc.by, o.acc.bz, __uint_as_float(o.slot));
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a release fence for global memory: it
// waits (s_waitcnt vmcnt(0)) until every store the wave has issued -- here the survivors' path state on its way to
// HBM -- is acknowledged.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool DEAL> // queue order: 5 waves per SIMD (96 VGPRs, 5 spilled: 1.26 -> 1.23 ms on C2); dealt: 4 (at 5 it spills 39 and loses)
__global__ __launch_bounds__(256, DEAL ? RT_SHADE_OCC : RT_SHADE_OCC + 1) void wf_shade_kernel(const RenderArgs a, const WfBufs w, const int level)
{
    __shared__ uint32_t wgCnt[8];
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;
    if (DEAL) __builtin_assume(level >= 1);   // launch_render_wf: <false> shades level 0, <true> the bounce levels
    if (!DEAL) { // queue order, one piece of 256 entries at a time
        const uint32_t piecesPerStripe = (w.subCap + 255u) / 256u;
        const uint32_t total = piecesPerStripe * kWfStripes;
        for (uint32_t piece = blockIdx.x; piece < total; piece += gridDim.x) {
            const uint32_t stripe = piece % kWfStripes, off = (piece / kWfStripes) * 256u + threadIdx.x;
            const uint32_t cnt = *wf_count(w, level, (int)stripe);
            if (off - threadIdx.x >= cnt) continue; // uniform per workgroup
            ShadeOut o; // per entry: nothing of it lives across the loop's back edge
            shade_entry(a, w, 0, stripe, off, off < cnt, st, o);   // (this kernel is launched for level 0 only: a constant to the compiler)
            shade_append(w, 0, stripe, o);
        }
    } else {
        constexpr uint32_t kBatchEntries = kShadeBatch * 256u;
        __shared__ uint16_t perm[kBatchEntries];            // dealt list: hits [0, nHit), misses [nHit, nLive), entries of the batch
        __shared__ uint32_t runCnt[2][kShadeBatch * 4 + 1]; // hits / misses per (piece, wave) of the classification pass, then their prefix sums
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long below = (1ull << lane) - 1ull;
        // pieces per batch: kShadeBatch for a full level; fewer for a thin one, so that every workgroup still gets a batch
        // (a deep level of a small frame holds a few thousand entries per stripe)
        const uint32_t piecesPerStripe = (w.subCap + 255u) / 256u;
        const uint32_t total = piecesPerStripe * kWfStripes;     // as many batches as there would be of one piece each
        const uint32_t wantPerStripe = (gridDim.x * 2u + kWfStripes - 1u) / kWfStripes;
        for (uint32_t batch = blockIdx.x; batch < total; batch += gridDim.x) {
            const uint32_t stripe = batch % kWfStripes;
            const uint32_t cnt = *wf_count(w, level, (int)stripe);
            uint32_t nbp = ((cnt + 255u) / 256u + wantPerStripe - 1u) / wantPerStripe;
            nbp = nbp < 1u ? 1u : (nbp > kShadeBatch ? kShadeBatch : nbp);
            const uint32_t base = (batch / kWfStripes) * nbp * 256u;
            if (base >= cnt) continue; // uniform per workgroup
            // ---- classification pass: every thread looks at kShadeBatch entries, 256 apart
            uint32_t cls = 0u; // 2 bits per piece: 1 = hit, 2 = miss
#pragma unroll
            for (uint32_t j = 0; j < kShadeBatch; j++) {
                const uint32_t off = base + j * 256u + threadIdx.x;
                const bool in = j < nbp && off < cnt;
                const uint32_t g0 = in ? w.hitM[wf_hit_pos(w, stripe, off)].x : kWfMiss;
                // a miss has nothing left to do: its result was stored when the ray was made (shade_entry).  One that carries the
                // flag of a dropped stack entry goes the way of the hits, which sends its sample to the redo launch
                const bool isHit = in && ((g0 & kWfMiss) != kWfMiss || (g0 & kWfDanger) != 0u);   // no holes past the camera rays
                const bool isMiss = false;
                cls |= (isHit ? 1u : (isMiss ? 2u : 0u)) << (2u * j);
                const unsigned long long hb = rt_ballot(isHit), mb = rt_ballot(isMiss);
                if (lane == 0) { runCnt[0][j * 4u + wave] = (uint32_t)__popcll(hb); runCnt[1][j * 4u + wave] = (uint32_t)__popcll(mb); }
            }
            lds_barrier();
            // ---- exclusive prefix sums over the (piece, wave) runs: wave 0 scans the hits, wave 1 the misses
            if (wave < 2) {
                static_assert(kShadeBatch * 4u <= 64u, "one wave scans the runs of a batch");
                const uint32_t v = (uint32_t)lane < kShadeBatch * 4u ? runCnt[wave][lane] : 0u;
                uint32_t incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                    if (lane >= d) incl += up;
                }
                if ((uint32_t)lane < kShadeBatch * 4u) runCnt[wave][lane] = incl - v;
                if (lane == 63) runCnt[wave][kShadeBatch * 4u] = incl;
            }
            lds_barrier();
            const uint32_t nHit = runCnt[0][kShadeBatch * 4u], nLive = nHit + runCnt[1][kShadeBatch * 4u];
#pragma unroll
            for (uint32_t j = 0; j < kShadeBatch; j++) {
                const uint32_t c = (cls >> (2u * j)) & 3u;
                const unsigned long long hb = rt_ballot(c == 1u), mb = rt_ballot(c == 2u);
                if (c == 1u) perm[runCnt[0][j * 4u + wave] + (uint32_t)__popcll(hb & below)] = (uint16_t)(j * 256u + threadIdx.x);
                if (c == 2u) perm[nHit + runCnt[1][j * 4u + wave] + (uint32_t)__popcll(mb & below)] = (uint16_t)(j * 256u + threadIdx.x);
            }
            lds_barrier();
            // ---- the waves take runs of 64 dealt entries in turn; nothing ties them together until the batch is done
            for (uint32_t r = (uint32_t)wave * 64u; r < nLive; r += 256u) {
                const bool mine = r + (uint32_t)lane < nLive;
                const uint32_t off = base + (mine ? (uint32_t)perm[r + (uint32_t)lane] : 0u);
                ShadeOut o;
                shade_entry(a, w, level, stripe, off, mine, st, o);
                shade_append(w, level, stripe, o);
            }
            lds_barrier(); // the list is rewritten by the next batch
        }
    }
    flush_stats(a, st, wgCnt);
}

// :88-137: per pixel, the sample results in the reference's summation order (after wf_redo has supplied the
// samples that left the pipeline).
__device__ __forceinline__ void wf_resolve_pixel(const RenderArgs& a, const WfBufs& w, uint32_t pix, int x, int y, LaneStats& st)
{
    const uint32_t spp = a.sampleCount;
    float tr = 0.0f, tg = 0.0f, tb = 0.0f;
    uint32_t checksum = 0, rays = 0;
    for (uint32_t z = 0; z < 4u; z++) {
        float pr = 0.0f, pg = 0.0f, pb = 0.0f; // a z lane without samples contributes 0
        for (uint32_t sidx = z; sidx < spp; sidx += 4u) {
            const size_t oi = wf_out_index(pix, sidx, spp);
            const float4 v = w.out[oi]; // the sample's outgoing light; the exposure (:185-187) is applied here, once per sample
            pr += 1.0f - rtd::exp_f(-v.x * a.exposure);
            pg += 1.0f - rtd::exp_f(-v.y * a.exposure);
            pb += 1.0f - rtd::exp_f(-v.z * a.exposure);
            checksum += __float_as_uint(v.w);
            rays += w.outRays[oi];
        }
        tr += pr; tg += pg; tb += pb;
    }
    st.rays += rays;
    write_pixel(a, x, y, tr, tg, tb, nullptr, checksum);
}

__global__ __launch_bounds__(256) void wf_resolve_kernel(const RenderArgs a, const WfBufs w)
{
    __shared__ uint32_t wgCnt[8];
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;
    int x, y;
    const uint32_t pix = blockIdx.x * 256u + threadIdx.x;
    if (pix < w.nPix && wf_pixel(a, w, pix, x, y)) wf_resolve_pixel(a, w, pix, x, y, st);
    flush_stats(a, st, wgCnt);
}

// The samples that left the pipeline (wf_redo_sample) again from the camera, one lane per path, with the exact
// in-order emulation inline (closest_hit<2>): tie bands between objects, culled winners, stack overflows, camera
// rays next to a listed thin triangle.  Exits at once when the lists are empty.
__global__ __launch_bounds__(256) void wf_redo_kernel(const RenderArgs a, const WfBufs w)
{
    __shared__ uint32_t ldsStack[kLdsStack * kBlock];
    __shared__ uint32_t wgCnt[8];
    // the stripes' lists read as one: entry i belongs to the last stripe whose running total is <= i
    __shared__ uint32_t listBase[kWfStripes + 1];
    if (threadIdx.x < 64) {
        const uint32_t cnt = w.ctrl[wf_ctrl_redo(wf_levels(a), (int)threadIdx.x)];
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        listBase[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) listBase[0] = 0u;
    }
    __syncthreads();
    const uint32_t n = listBase[kWfStripes];
    uint32_t ovf[kStackDepth - kLdsStack];
    Stack stack; stack.lds = ldsStack + threadIdx.x; stack.ovf = ovf;
    LaneStats st; st.rays = 0; st.slow = 0; st.triBrute = 0; st.hits = 0; st.bvh.nodes = 0; st.bvh.tris = 0;
    const uint32_t spp = a.sampleCount;
    // A whole path per lane, every lane on a walk of its own, one dependent fetch after the other: the items are spread over
    // all the waves of the launch, `per` lanes of each, so that a short list costs one path's latency and not 64 of them in turn.
    const uint32_t waves = gridDim.x * 4u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t per = (n + waves - 1u) / waves;
    per = per < 1u ? 1u : (per > 64u ? 64u : per);
    for (uint32_t base = wave * per; base < n; base += waves * per) {
        const uint32_t item = base + lane;
        if (lane >= per || item >= n) continue;
        uint32_t ls = 0;
#pragma unroll
        for (uint32_t step = kWfStripes / 2; step > 0; step >>= 1)
            if (listBase[ls + step] <= item) ls += step;
        const uint32_t slot = w.redoSlots[(size_t)ls * w.subCap + (item - listBase[ls])];
        const uint32_t pix = fast_div(slot, a.divSpp), sidx = slot - pix * spp;
        int x, y;
        wf_pixel(a, w, pix, x, y);
        uint32_t seed = sample_seed(a, x, y, sidx), checksum = 0;
        uint32_t primary[3] = { kNoHit, kNoHit, kNoHit };
        Ray ray = setup_ray(a, x, y, seed);
        const uint32_t rays0 = st.rays;
        const V4 c = trace_path<2>(a, ray, seed, sidx, stack, st, checksum, primary);
        const size_t oi = wf_out_index(pix, sidx, spp);
        w.out[oi] = make_float4(c.x, c.y, c.z, __uint_as_float(checksum));
        w.outRays[oi] = (uint8_t)(st.rays - rays0);
        if (sidx == 0 && a.ids) {
            uint32_t* idp = reinterpret_cast<uint32_t*>(a.ids + ((size_t)y * (size_t)a.width + (size_t)x));
            idp[0] = primary[0]; idp[1] = primary[1]; idp[2] = primary[2];
        }
    }
    st.rays = 0; // counted by wf_resolve from outRays
    flush_stats(a, st, wgCnt);
}

// mode 0: persistent path kernel + redo launch for the pixels it flagged; mode 1: brute force
void launch_render(const RenderArgs& a, int mode, int numCUs, hipStream_t stream)
{
    if (a.x1 <= a.x0 || a.y1 <= a.y0) return;
    RenderArgs b = a;
    if (mode == 1) {
        const uint32_t spp = a.sampleCount;
        const int log2l = spp >= 3 ? 2 : (spp == 2 ? 1 : 0);
        const int L = 1 << log2l;
        const int BW = 2 * ((L == 4) ? 4 : 8), BH = 2 * ((L == 1) ? 8 : 4);
        b.tilesX = (a.x1 - a.x0 + BW - 1) / BW;
        b.tilesY = (a.y1 - a.y0 + BH - 1) / BH;
        b.pixelList = nullptr; b.pixelCount = nullptr;
        const uint32_t grid = (uint32_t)b.tilesX * (uint32_t)b.tilesY;
        if (log2l == 0) hipLaunchKernelGGL((render_kernel<0, 1>), dim3(grid), dim3(256), 0, stream, b);
        else if (log2l == 1) hipLaunchKernelGGL((render_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((render_kernel<2, 1>), dim3(grid), dim3(256), 0, stream, b);
        return;
    }
    b.tiles8X = (a.x1 - a.x0 + 7) / 8;
    b.tiles8Y = (a.y1 - a.y0 + 7) / 8;
    const uint32_t nTiles = (uint32_t)b.tiles8X * (uint32_t)b.tiles8Y;
    // persistent grid: 4 workgroups (16 waves) per CU is what 128 VGPRs admit; never more waves than tiles
    uint32_t grid = (uint32_t)numCUs * 3u; // persistent grid = what the kernel's 168 VGPRs admit per CU (3 workgroups)
    const uint32_t need = (nTiles + 3u) / 4u;
    if (grid > need) grid = need;
    if (grid < 8u) grid = 8u;
    hipLaunchKernelGGL(path_kernel, dim3(grid), dim3(256), 0, stream, b);
    // redo launch: exits at once when nothing was flagged
    RenderArgs c = b;
    c.pixelList = b.redoList; c.pixelCount = b.redoCount;
    hipLaunchKernelGGL((render_kernel<2, 2>), dim3(256), dim3(256), 0, stream, c);
}

// rows of 8x8 tiles the wavefront pipeline walks for these arguments: the rectangle's, or those of the bands
// this call owns (rt_render_bands)
int wf_tile_rows(const RenderArgs& a)
{
    const int rows = (a.y1 - a.y0 + 7) / 8;
    if (a.bandTileRows <= 0) return rows;
    const int nBands = (rows + a.bandTileRows - 1) / a.bandTileRows;
    const int owned = a.bandFirst < nBands ? (nBands - a.bandFirst + a.bandStride - 1) / a.bandStride : 0;
    return owned * a.bandTileRows;
}

// Wavefront pipeline + redo launches.  The render rectangle is cut into chunks of whole 8x8 tiles;
// chunk k runs on lane k % n (a lane = a HIP stream with its own workspace).  The lanes exist for
// the tails: a level's trace launch ends with its slowest ray, traversed by one lane of one wave
// at memory latency (measured on the 1 M-triangle scene: waves alive 53 % of the launch, the SQs
// busy 57 %), and with several lanes in flight the other chunks' launches fill the machine meanwhile.
void launch_render_wf(const RenderArgs& a, const WfLanes& L, int numCUs, hipStream_t mainStream)
{
    if (a.x1 <= a.x0 || a.y1 <= a.y0) return;
    RenderArgs b = a;
    b.tiles8X = (a.x1 - a.x0 + 7) / 8;
    b.tiles8Y = wf_tile_rows(a);
    b.divSpp = make_fast_div(a.sampleCount); b.divTilesX = make_fast_div((uint32_t)b.tiles8X);
    b.divBandRows = make_fast_div(a.bandTileRows > 0 ? (uint32_t)a.bandTileRows : 1u);
    b.pixelList = nullptr; b.pixelCount = nullptr;
    const uint32_t nTiles = (uint32_t)b.tiles8X * (uint32_t)b.tiles8Y;
    if (nTiles == 0) return;
    const uint32_t spp = a.sampleCount;
    const uint32_t tilesPerChunk = L.tilesPerChunk;
    if (tilesPerChunk == 0 || L.n <= 0) return;
    const int levels = a.bounceCount > 0 ? a.bounceCount : 0;
    const size_t ctrlWords = wf_ctrl_words(levels);
    const uint32_t nChunks = (nTiles + tilesPerChunk - 1) / tilesPerChunk;
    const int used = (int)(nChunks < (uint32_t)L.n ? nChunks : (uint32_t)L.n);
    if (used > 1) {
        hipEventRecord(L.fork, mainStream);
        for (int j = 1; j < used; j++) hipStreamWaitEvent(L.streams[L.first + j], L.fork, 0);
    }
    for (uint32_t k = 0; k < nChunks; k++) {
        const int j = (int)(k % (uint32_t)used);
        hipStream_t stream = j == 0 ? mainStream : L.streams[L.first + j];
        WfBufs w = L.bufs[L.first + j];
        const uint32_t nt = (nTiles - k + nChunks - 1) / nChunks; // tiles k, k + nChunks, ...
        w.tileBegin = k; w.tileStride = nChunks; w.nPix = nt * 64u; w.nSlots = w.nPix * spp;
        const uint32_t nPieces = (w.nSlots + 255u) / 256u;
        RenderArgs c = b;
        hipMemsetAsync(w.ctrl, 0, sizeof(uint32_t) * ctrlWords, stream);
        hipLaunchKernelGGL(wf_gen_kernel, dim3(nPieces), dim3(256), 0, stream, c, w);
        uint32_t gridT = (uint32_t)numCUs * RT_WF_OCC;
        const uint32_t needT = (nPieces + 3u) / 4u; // never more waves than 64-entry grabs
        if (gridT > needT) gridT = needT;
#ifndef RT_SHADE_GRID
#define RT_SHADE_GRID 12u // shade workgroups per CU (grid-stride over pieces / batches): 6 / 8 / 12 / 16 / 24 / 32 measured; 12 is 0.6 % ahead of 8 on C2, 1.3 % on C3
#endif
        uint32_t gridS = (uint32_t)numCUs * RT_SHADE_GRID;
        if (gridS > nPieces) gridS = nPieces;
        for (int level = 0; level < levels; level++) {
            if (level == 0) {
                uint32_t gridP = (uint32_t)numCUs * RT_PK_OCC;
                if (gridP > needT) gridP = needT;
                hipLaunchKernelGGL(wf_trace_packet_kernel, dim3(gridP), dim3(256), 0, stream, c, w, level, levels);
            }
            else hipLaunchKernelGGL(wf_trace_kernel, dim3(gridT), dim3(256), 0, stream, c, w, level, levels);
#if RT_EXACT_PASS
            hipLaunchKernelGGL(wf_exact_kernel, dim3(gridT < RT_EXACT_GRID ? gridT : RT_EXACT_GRID), dim3(256), 0, stream, c, w, level, levels);
#endif
            if (level == 0) hipLaunchKernelGGL((wf_shade_kernel<false>), dim3(gridS), dim3(256), 0, stream, c, w, level);
            else hipLaunchKernelGGL((wf_shade_kernel<true>), dim3(gridS), dim3(256), 0, stream, c, w, level);
        }
        // the samples that left the pipeline, one lane per path (exits at once when there is none); then every pixel's sum
        uint32_t gridR = (w.nSlots + 255u) / 256u;
        if (gridR > (uint32_t)numCUs * 2u) gridR = (uint32_t)numCUs * 2u; // 2 workgroups per CU is what its 222 VGPRs admit; the loop strides
        hipLaunchKernelGGL(wf_redo_kernel, dim3(gridR), dim3(256), 0, stream, c, w);
        hipLaunchKernelGGL(wf_resolve_kernel, dim3((w.nPix + 255u) / 256u), dim3(256), 0, stream, c, w);
    }
    for (int j = 1; j < used; j++) {
        hipEventRecord(L.join[L.first + j], L.streams[L.first + j]);
        hipStreamWaitEvent(mainStream, L.join[L.first + j], 0);
    }
}

void launch_trace_rays(const SceneView& s, const float4* rays, int n, uint4* out, unsigned long long* counters, int mode,
                       hipStream_t stream)
{
    if (n <= 0) return;
    const int grid = (n + 255) / 256;
    if (mode == 1) hipLaunchKernelGGL((trace_rays_kernel<1>), dim3(grid), dim3(256), 0, stream, s, rays, n, out, counters);
    else hipLaunchKernelGGL((trace_rays_kernel<3>), dim3(grid), dim3(256), 0, stream, s, rays, n, out, counters);
}

} // namespace rt
