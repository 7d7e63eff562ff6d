/*
 * rt_oracle.c -- TEST INFRASTRUCTURE (see rt_oracle.h).  Plain-C restatement of the
 * reference kernel logic; brute force over every triangle, exactly as the reference.
 * Every function cites the reference lines (relative to /root/reference/src) it follows.
 * Build: gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile).  No fast-math, no FMA.
 *
 * Pinned definitions where the reference relies on NVIDIA hardware:
 *  - transcendentals: oracle_math.h ("detmath v1");  rsqrtf(x) := 1.0f/sqrtf(x)
 *  - tex2D bilinear: CUDA programming-guide formula, weights rounded to 8 fractional bits
 *  - shared-memory slots never written when spp < 4 (GPURaytracer.cu:84,112-116) read as 0
 *  - object iteration order := array order (reference: unordered_map order, Scene.cpp:7)
 */
#include "rt_oracle.h"
#include "oracle_math.h"

#include <float.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_PI 3.141592f /* processor/Raytracer.hpp:16 */
#define OR_BLOCK_DEPTH 4 /* processor/GPU/GPURaytracer.hpp:19 */

typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c0, c1, c2, c3; } m4; /* linal.hpp:29-35, columns */

static inline v4 mk4(float x, float y, float z, float w) { v4 r = { x, y, z, w }; return r; }

static inline m4 load_m4(const float* f)
{
    m4 m;
    m.c0 = mk4(f[0], f[1], f[2], f[3]);
    m.c1 = mk4(f[4], f[5], f[6], f[7]);
    m.c2 = mk4(f[8], f[9], f[10], f[11]);
    m.c3 = mk4(f[12], f[13], f[14], f[15]);
    return m;
}

/* GPURaytracer.cu:728-743 */
static inline v4 mat_vec(const m4* m, v4 v)
{
    return mk4(m->c0.x * v.x + m->c1.x * v.y + m->c2.x * v.z + m->c3.x * v.w,
               m->c0.y * v.x + m->c1.y * v.y + m->c2.y * v.z + m->c3.y * v.w,
               m->c0.z * v.x + m->c1.z * v.y + m->c2.z * v.z + m->c3.z * v.w,
               m->c0.w * v.x + m->c1.w * v.y + m->c2.w * v.z + m->c3.w * v.w);
}
static inline v4 mat_vec_t(const m4* m, v4 v)
{
    return mk4(m->c0.x * v.x + m->c0.y * v.y + m->c0.z * v.z + m->c0.w * v.w,
               m->c1.x * v.x + m->c1.y * v.y + m->c1.z * v.z + m->c1.w * v.w,
               m->c2.x * v.x + m->c2.y * v.y + m->c2.z * v.z + m->c2.w * v.w,
               m->c3.x * v.x + m->c3.y * v.y + m->c3.z * v.z + m->c3.w * v.w);
}

/* :771-817 */
static inline v4 cross4(v4 a, v4 b)
{
    return mk4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f);
}
static inline float dot3(v4 a, v4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v4 negate4(v4 v) { return mk4(-v.x, -v.y, -v.z, v.w); }
static inline v4 lerp4(v4 a, v4 b, float t)
{
    return mk4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
static inline v4 normalize4(v4 v)
{
    float inv = om_rsqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk4(v.x * inv, v.y * inv, v.z * inv, v.w);
}
/* :861-872 */
static inline v4 reflect4(v4 v, v4 n)
{
    float d = dot3(v, n);
    v4 o = mk4(v.x - 2.0f * d * n.x, v.y - 2.0f * d * n.y, v.z - 2.0f * d * n.z, 0.0f);
    return normalize4(o);
}
/* :700-717 */
static inline v4 rotate_q(v4 v, v4 q)
{
    float t2 = q.w * q.x, t3 = q.w * q.y, t4 = q.w * q.z;
    float t5 = -q.x * q.x, t6 = q.x * q.y, t7 = q.x * q.z;
    float t8 = -q.y * q.y, t9 = q.y * q.z, t10 = -q.z * q.z;
    return mk4(2.0f * ((t8 + t10) * v.x + (t6 - t4) * v.y + (t3 + t7) * v.z) + v.x,
               2.0f * ((t4 + t6) * v.x + (t5 + t10) * v.y + (t9 - t2) * v.z) + v.y,
               2.0f * ((t7 - t3) * v.x + (t2 + t9) * v.y + (t5 + t8) * v.z) + v.z,
               v.w);
}

/* :819-825 -- PCG-style hash; the division is in double, the return narrows to float */
static inline float random_value(uint32_t* seed)
{
    uint32_t s = *seed * 747796405u + 2891336453u;
    *seed = s;
    uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    r = (r >> 22) ^ r;
    return (float)((double)r / 4294967295.0);
}
/* :827-833 -- theta is drawn first, then r */
static inline float random_normal(uint32_t* seed)
{
    float theta = (2.0f * OR_PI) * random_value(seed);
    float r = random_value(seed);
    float rho = sqrtf(-2.0f * om_logf(r));
    return rho * om_cosf(theta);
}
/* :835-841 */
static inline v4 random_unit(uint32_t* seed)
{
    float x = random_normal(seed);
    float y = random_normal(seed);
    float z = random_normal(seed);
    return normalize4(mk4(x, y, z, 0.0f));
}
/* :853-859 */
static inline v4 random_cosine_hemi(uint32_t* seed, v4 n)
{
    v4 u = random_unit(seed);
    return normalize4(mk4(u.x + n.x, u.y + n.y, u.z + n.z, 0.0f));
}

/* ---------------- texture filtering (pinned definition of tex2D) ---------------- */

typedef struct { int i0, i1; float a; } tap;

/* xB = coordinate already scaled to texel units minus 0.5; n = extent */
static inline tap make_tap(float xB, int n)
{
    tap t;
    if (!(xB >= -1.0f)) xB = -1.0f;          /* also catches NaN */
    if (xB > (float)n) xB = (float)n;
    float f = floorf(xB);
    float a = xB - f;
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f); /* 1.8 fixed point */
    int i = (int)f;
    int i1 = i + 1;
    if (i < 0) i = 0;
    if (i > n - 1) i = n - 1;
    if (i1 < 0) i1 = 0;
    if (i1 > n - 1) i1 = n - 1;
    t.i0 = i; t.i1 = i1; t.a = a;
    return t;
}

static inline v4 bilerp(v4 t00, v4 t10, v4 t01, v4 t11, float a, float b)
{
    float w00 = (1.0f - a) * (1.0f - b);
    float w10 = a * (1.0f - b);
    float w01 = (1.0f - a) * b;
    float w11 = a * b;
    return mk4(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x,
               ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
               ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z,
               ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w);
}

static inline v4 texel8(const or_texture* t, int x, int y)
{
    const uint8_t* p = t->rgba + 4 * ((size_t)y * (size_t)t->w + (size_t)x);
    /* Material.cu:203: (float)byte / 255.0f */
    return mk4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
}

/* Material.cu:224-228: normalized coordinates, linear, clamp */
static v4 tex2d_norm(const or_texture* t, float u, float v)
{
    tap tx = make_tap(u * (float)t->w - 0.5f, t->w);
    tap ty = make_tap(v * (float)t->h - 0.5f, t->h);
    return bilerp(texel8(t, tx.i0, ty.i0), texel8(t, tx.i1, ty.i0),
                  texel8(t, tx.i0, ty.i1), texel8(t, tx.i1, ty.i1), tx.a, ty.a);
}

static inline v4 texel_ray(const float* table, int w, int x, int y)
{
    const float* p = table + 4 * ((size_t)y * (size_t)w + (size_t)x);
    return mk4(p[0], p[1], p[2], p[3]);
}

/* Camera.hpp:93-97: unnormalized coordinates, linear, clamp */
static v4 tex2d_rays(const float* table, int w, int h, float x, float y)
{
    tap tx = make_tap(x - 0.5f, w);
    tap ty = make_tap(y - 0.5f, h);
    return bilerp(texel_ray(table, w, tx.i0, ty.i0), texel_ray(table, w, tx.i1, ty.i0),
                  texel_ray(table, w, tx.i0, ty.i1), texel_ray(table, w, tx.i1, ty.i1), tx.a, ty.a);
}

/* ---------------- ray, hits ---------------- */

typedef struct { v4 origin, direction; int bounceCount; float maxDistance; } ray_t; /* GPURaytracer.hpp:52-58 */
typedef struct { float distance; float b0, b1, b2; } trihit_t;                       /* :46-50 */
typedef struct {
    int didHit;
    uint32_t objectIndex, meshIndex, triIndex;
    float distance;
    v4 hitPosition;
    float u, v;
    m4 tbn;
    int materialId;
} rayhit_t; /* :22-34 (+ mesh/triangle ids for the hit-id AOV) */

/* :551-637 -- PBRT-v3 watertight ray/triangle with the double-precision edge fallback */
static trihit_t dist_to_triangle(const ray_t* ray, v4 v0, v4 v1, v4 v2)
{
    trihit_t out; out.distance = FLT_MAX; out.b0 = out.b1 = out.b2 = 0.0f;

    float p0[3] = { v0.x - ray->origin.x, v0.y - ray->origin.y, v0.z - ray->origin.z };
    float p1[3] = { v1.x - ray->origin.x, v1.y - ray->origin.y, v1.z - ray->origin.z };
    float p2[3] = { v2.x - ray->origin.x, v2.y - ray->origin.y, v2.z - ray->origin.z };
    float d[3] = { ray->direction.x, ray->direction.y, ray->direction.z };

    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    int kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;

    float dx = d[kx], dy = d[ky], dz = d[kz];
    float p0x = p0[kx], p0y = p0[ky], p0z = p0[kz];
    float p1x = p1[kx], p1y = p1[ky], p1z = p1[kz];
    float p2x = p2[kx], p2y = p2[ky], p2z = p2[kz];

    float sz = 1.f / dz;
    float sx = -dx * sz;
    float sy = -dy * sz;

    p0x += sx * p0z; p0y += sy * p0z;
    p1x += sx * p1z; p1y += sy * p1z;
    p2x += sx * p2z; p2y += sy * p2z;

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

    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return out;
    float det = e0 + e1 + e2;
    if (det == 0) return out;

    p0z *= sz; p1z *= sz; p2z *= sz;
    float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0 && (tScaled >= 0 || tScaled < ray->maxDistance * det)) return out;
    else if (det > 0 && (tScaled <= 0 || tScaled > ray->maxDistance * det)) return out;

    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;

    if ((double)t < 0.001 || t > ray->maxDistance) return out;

    out.distance = t; out.b0 = b0; out.b1 = b1; out.b2 = b2;
    return out;
}

/* :267-315 -- slab test in object space; strict tmin < tmax */
static int intersects_object_box(const ray_t* ray, const or_object* obj, const m4* inv)
{
    float tmin = 0.0f;
    float tmax = ray->maxDistance;

    v4 dirBB = mat_vec(inv, ray->direction);
    v4 orgBB = mat_vec(inv, ray->origin);
    float ix = 1.0f / dirBB.x, iy = 1.0f / dirBB.y, iz = 1.0f / dirBB.z;

    int sg = signbit(dirBB.x) != 0;
    float bmin = sg ? obj->bmax[0] : obj->bmin[0];
    float bmax = sg ? obj->bmin[0] : obj->bmax[0];
    float dmin = (bmin - orgBB.x) * ix;
    float dmax = (bmax - orgBB.x) * ix;
    tmin = om_fmaxf(dmin, tmin);
    tmax = om_fminf(dmax, tmax);

    sg = signbit(dirBB.y) != 0;
    bmin = sg ? obj->bmax[1] : obj->bmin[1];
    bmax = sg ? obj->bmin[1] : obj->bmax[1];
    dmin = (bmin - orgBB.y) * iy;
    dmax = (bmax - orgBB.y) * iy;
    tmin = om_fmaxf(dmin, tmin);
    tmax = om_fminf(dmax, tmax);

    sg = signbit(dirBB.z) != 0;
    bmin = sg ? obj->bmax[2] : obj->bmin[2];
    bmax = sg ? obj->bmin[2] : obj->bmax[2];
    dmin = (bmin - orgBB.z) * iz;
    dmax = (bmax - orgBB.z) * iz;
    tmin = om_fmaxf(dmin, tmin);
    tmax = om_fminf(dmax, tmax);

    return tmin < tmax;
}

static inline v4 load_pos(const float* a, int i3) { return mk4(a[i3], a[i3 + 1], a[i3 + 2], 1.0f); }

/* :379-549 -- brute force over every mesh and triangle, then the winner's attributes */
static rayhit_t intersect_object(const ray_t* ray, const or_scene* sc, const or_object* obj,
                                 const m4* model, const m4* inv)
{
    int bestMesh = -1, bestTri = -1;
    trihit_t best; best.distance = FLT_MAX; best.b0 = best.b1 = best.b2 = 0.0f;

    for (int mi = 0; mi < obj->meshCount; mi++) {
        const or_mesh* me = &sc->meshes[obj->meshIds[mi]];
        for (int i = 0; i < me->nTris; i++) {
            v4 v0 = mat_vec(model, load_pos(me->pos, me->idx[i * 3 + 0] * 3));
            v4 v1 = mat_vec(model, load_pos(me->pos, me->idx[i * 3 + 1] * 3));
            v4 v2 = mat_vec(model, load_pos(me->pos, me->idx[i * 3 + 2] * 3));
            trihit_t h = dist_to_triangle(ray, v0, v1, v2);
            if (h.distance < best.distance) { best = h; bestTri = i; bestMesh = mi; }
        }
    }

    rayhit_t out;
    out.didHit = 0; out.objectIndex = out.meshIndex = out.triIndex = 0xFFFFFFFFu;
    out.distance = FLT_MAX; out.hitPosition = mk4(0, 0, 0, 0); out.u = out.v = 0.0f;
    out.tbn.c0 = out.tbn.c1 = out.tbn.c2 = out.tbn.c3 = mk4(0, 0, 0, 0); out.materialId = -1;
    if (bestMesh == -1) return out;

    const or_mesh* me = &sc->meshes[obj->meshIds[bestMesh]];
    int i0 = me->idx[bestTri * 3 + 0], i1 = me->idx[bestTri * 3 + 1], i2 = me->idx[bestTri * 3 + 2];

    v4 v0 = load_pos(me->pos, i0 * 3), v1 = load_pos(me->pos, i1 * 3), v2 = load_pos(me->pos, i2 * 3);
    v4 n0 = mk4(0, 0, 0, 0), n1 = n0, n2 = n0;
    if (me->nrm) {
        n0 = mat_vec_t(inv, mk4(me->nrm[i0 * 3], me->nrm[i0 * 3 + 1], me->nrm[i0 * 3 + 2], 0.0f));
        n1 = mat_vec_t(inv, mk4(me->nrm[i1 * 3], me->nrm[i1 * 3 + 1], me->nrm[i1 * 3 + 2], 0.0f));
        n2 = mat_vec_t(inv, mk4(me->nrm[i2 * 3], me->nrm[i2 * 3 + 1], me->nrm[i2 * 3 + 2], 0.0f));
    }
    v0 = mat_vec(model, v0); v1 = mat_vec(model, v1); v2 = mat_vec(model, v2);

    float u0 = 0, w0 = 0, u1 = 0, w1 = 0, u2 = 0, w2 = 0; /* reference would fault on a NULL uv array */
    if (me->uv) {
        u0 = me->uv[i0 * 2]; w0 = me->uv[i0 * 2 + 1];
        u1 = me->uv[i1 * 2]; w1 = me->uv[i1 * 2 + 1];
        u2 = me->uv[i2 * 2]; w2 = me->uv[i2 * 2 + 1];
    }

    v4 e1 = mk4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, 0.0f);
    v4 e2 = mk4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, 0.0f);
    float du1 = u1 - u0, dv1 = w1 - w0, du2 = u2 - u0, dv2 = w2 - w0;

    float f = 1.0f / (du1 * dv2 - du2 * dv1);
    v4 tangent = mk4(f * (dv2 * e1.x - dv1 * e2.x), f * (dv2 * e1.y - dv1 * e2.y), f * (dv2 * e1.z - dv1 * e2.z), 0.0f);
    v4 bitangent = mk4(f * (-du2 * e1.x + du1 * e2.x), f * (-du2 * e1.y + du1 * e2.y), f * (-du2 * e1.z + du1 * e2.z), 0.0f);
    tangent = normalize4(tangent);
    bitangent = normalize4(bitangent);
    v4 normal;
    if (me->nrm) {
        normal = mk4(n0.x * best.b0 + n1.x * best.b1 + n2.x * best.b2,
                     n0.y * best.b0 + n1.y * best.b1 + n2.y * best.b2,
                     n0.z * best.b0 + n1.z * best.b1 + n2.z * best.b2, 0.0f);
        float corr = dot3(tangent, normal);
        tangent = normalize4(mk4(tangent.x - corr * normal.x, tangent.y - corr * normal.y, tangent.z - corr * normal.z, 0.0f));
        bitangent = normalize4(cross4(normal, tangent));
    } else {
        normal = normalize4(cross4(e2, e1));
    }

    out.tbn.c0 = mk4(tangent.x, tangent.y, tangent.z, 0.0f);
    out.tbn.c1 = mk4(bitangent.x, bitangent.y, bitangent.z, 0.0f);
    out.tbn.c2 = mk4(normal.x, normal.y, normal.z, 0.0f);
    out.tbn.c3 = mk4(0.0f, 0.0f, 0.0f, 1.0f);

    out.hitPosition = mk4(v0.x * best.b0 + v1.x * best.b1 + v2.x * best.b2,
                          v0.y * best.b0 + v1.y * best.b1 + v2.y * best.b2,
                          v0.z * best.b0 + v1.z * best.b1 + v2.z * best.b2, 1.0f);
    out.u = u0 * best.b0 + u1 * best.b1 + u2 * best.b2;
    out.v = w0 * best.b0 + w1 * best.b1 + w2 * best.b2;

    out.didHit = 1;
    out.distance = best.distance;
    out.meshIndex = (uint32_t)bestMesh;
    out.triIndex = (uint32_t)bestTri;
    out.materialId = obj->materialIds[obj->materialIndices[bestMesh]];
    return out;
}

/* :247-265 -- objects in array order; ray.maxDistance is narrowed by every accepted hit */
static rayhit_t closest_hit(ray_t* ray, const or_scene* sc)
{
    rayhit_t closest;
    closest.didHit = 0; closest.objectIndex = closest.meshIndex = closest.triIndex = 0xFFFFFFFFu;
    closest.distance = FLT_MAX; closest.materialId = -1;
    closest.hitPosition = mk4(0, 0, 0, 0); closest.u = closest.v = 0.0f;
    closest.tbn.c0 = closest.tbn.c1 = closest.tbn.c2 = closest.tbn.c3 = mk4(0, 0, 0, 0);
    for (int i = 0; i < sc->nObjects; i++) {
        const or_object* obj = &sc->objects[i];
        m4 model = load_m4(obj->model), inv = load_m4(obj->invModel);
        if (!intersects_object_box(ray, obj, &inv)) continue;
        rayhit_t hp = intersect_object(ray, sc, obj, &model, &inv);
        if (hp.didHit && hp.distance < closest.distance) {
            closest = hp;
            closest.objectIndex = (uint32_t)i;
            ray->maxDistance = hp.distance;
        }
    }
    return closest;
}

/* :4-20 */
static v4 sky_light(v4 dir, const or_sky* sky)
{
    v4 l = mk4(sky->direction[0], sky->direction[1], sky->direction[2], sky->direction[3]);
    float sun = om_powf(om_fmaxf(0.0f, dot3(dir, l)), 1000.0f) * sky->lightColor[3];
    return mk4(sky->skyColor[0] * sky->skyColor[3] + sky->lightColor[0] * sun,
               sky->skyColor[1] * sky->skyColor[3] + sky->lightColor[1] * sun,
               sky->skyColor[2] * sky->skyColor[3] + sky->lightColor[2] * sun, 1.0f);
}

/* :640-686 */
static inline float fresnel_term(float cosTheta, float f0)
{
    return f0 + (1.0f - f0) * om_powf(1.0f - om_fminf(cosTheta, 1.0f), 5.0f);
}
static inline float distribution_ggx(v4 n, v4 h, float roughness)
{
    float a = roughness * roughness;
    float a2 = a * a;
    float ndh = om_fmaxf(0.0f, dot3(n, h));
    float ndh2 = ndh * ndh;
    float denom = (ndh2 * (a2 - 1.0f) + 1.0f);
    denom = OR_PI * denom * denom;
    return a2 / (denom + 0.00001f);
}
static inline float geometry_schlick(float ndv, float roughness)
{
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return ndv / (ndv * (1.0f - k) + k);
}
static inline float geometry_smith(v4 n, v4 v, v4 l, float roughness)
{
    float ndv = om_fmaxf(dot3(n, v), 0.0f);
    float ndl = om_fmaxf(dot3(n, l), 0.0f);
    return geometry_schlick(ndv, roughness) * geometry_schlick(ndl, roughness);
}

typedef struct { v4 albedo, normal; float ao[3]; float roughness, metal; v4 emission; } matdata_t;

/* :363-377 + Material.cu:268-323.  AO/emission textures never reach the device
 * (Material.cu:41-44), so AO is the constant (1,1,1) and emission the constant colour. */
static matdata_t material_data(const rayhit_t* hit, const or_scene* sc)
{
    matdata_t o;
    const or_material* m = &sc->materials[hit->materialId];
    float u = hit->u, v = hit->v;

    if (m->texAlbedo < 0) o.albedo = mk4(m->albedo[0], m->albedo[1], m->albedo[2], m->albedo[3]);
    else o.albedo = tex2d_norm(&sc->textures[m->texAlbedo], u, v);

    v4 nrm;
    if (m->texNormal < 0) nrm = mk4(0.0f, 0.0f, 1.0f, 0.0f);
    else {
        v4 n = tex2d_norm(&sc->textures[m->texNormal], u, v);
        nrm = mk4(n.x * 2 - 1.0f, n.y * 2 - 1.0f, n.z * 2 - 1.0f, 0.0f);
    }
    o.normal = normalize4(mat_vec(&hit->tbn, nrm));

    o.roughness = m->texRoughness < 0 ? m->roughness : tex2d_norm(&sc->textures[m->texRoughness], u, v).x;
    o.metal = m->texMetal < 0 ? m->metal : tex2d_norm(&sc->textures[m->texMetal], u, v).x;
    o.ao[0] = o.ao[1] = o.ao[2] = 1.0f;
    o.emission = mk4(m->emissionColor[0], m->emissionColor[1], m->emissionColor[2], m->emissionStrength);
    return o;
}

static inline uint32_t seg_hash(uint32_t obj, uint32_t mesh, uint32_t tri, uint32_t sample, uint32_t bounce)
{
    uint32_t h = obj * 0x9E3779B1u;
    h ^= (mesh + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= (tri + 1u) * 0xC2B2AE35u;
    h ^= sample * 0x27D4EB2Fu + bounce * 0x165667B1u + 1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

typedef struct { uint32_t primary[3]; uint32_t checksum; uint64_t queries; } path_aov;

/* :165-245 */
static v4 trace(ray_t* ray, const or_scene* sc, const or_sky* sky, float aoIntensity, uint32_t* seed,
                uint32_t sample, path_aov* aov)
{
    v4 out = mk4(0.0f, 0.0f, 0.0f, 1.0f);
    v4 beta = mk4(1.0f, 1.0f, 1.0f, 1.0f);

    for (int i = 0; i < ray->bounceCount; i++) {
        rayhit_t hit = closest_hit(ray, sc);
        aov->queries++;
        aov->checksum += seg_hash(hit.objectIndex, hit.meshIndex, hit.triIndex, sample, (uint32_t)i);
        if (i == 0 && sample == 0) {
            aov->primary[0] = hit.objectIndex; aov->primary[1] = hit.meshIndex; aov->primary[2] = hit.triIndex;
        }
        if (!hit.didHit) {
            v4 miss = sky_light(ray->direction, sky);
            out.x += miss.x * miss.w * beta.x;
            out.y += miss.y * miss.w * beta.y;
            out.z += miss.z * miss.w * beta.z;
            break;
        }
        matdata_t md = material_data(&hit, sc);

        if (i == 0) {
            out.x += md.ao[0] * md.albedo.x * aoIntensity;
            out.y += md.ao[1] * md.albedo.y * aoIntensity;
            out.z += md.ao[2] * md.albedo.z * aoIntensity;
        }

        v4 em = md.emission;
        out.x += em.x * em.w * beta.x;
        out.y += em.y * em.w * beta.y;
        out.z += em.z * em.w * beta.z;

        if (i < ray->bounceCount - 1) {
            float attenuation = 1.0f;
            v4 N = md.normal;
            v4 diffuseDir = random_cosine_hemi(seed, N);
            v4 specularDir = reflect4(ray->direction, N);
            v4 L = normalize4(lerp4(specularDir, diffuseDir, md.roughness));
            v4 V = negate4(ray->direction);
            v4 H = normalize4(mk4(L.x + V.x, L.y + V.y, L.z + V.z, 0.0f));

            v4 F0 = lerp4(mk4(0.04f, 0.04f, 0.04f, 1.0f), md.albedo, md.metal);

            float NDF = distribution_ggx(N, H, md.roughness);
            float G = geometry_smith(N, V, L, md.roughness);
            float cosT = om_fmaxf(dot3(H, V), 0.0f);
            float Fx = fresnel_term(cosT, F0.x), Fy = fresnel_term(cosT, F0.y), Fz = fresnel_term(cosT, F0.z);

            float mc = 1.0f - md.metal;
            float kDx = (1.0f - Fx) * mc, kDy = (1.0f - Fy) * mc, kDz = (1.0f - Fz) * mc;

            float numx = NDF * G * Fx, numy = NDF * G * Fy, numz = NDF * G * Fz;
            float denominator = 4.0f * om_fmaxf(dot3(N, V), 0.0f) * om_fmaxf(dot3(N, L), 0.0f) + 0.0001f;
            float spx = numx / denominator, spy = numy / denominator, spz = numz / denominator;

            float NdotL = om_fmaxf(dot3(N, L), 0.0f);
            beta.x *= (kDx * md.albedo.x / OR_PI + spx) * attenuation * NdotL;
            beta.y *= (kDy * md.albedo.y / OR_PI + spy) * attenuation * NdotL;
            beta.z *= (kDz * md.albedo.z / OR_PI + spz) * attenuation * NdotL;

            float nudge = 0.0001f;
            ray->origin = mk4(hit.hitPosition.x + N.x * nudge, hit.hitPosition.y + N.y * nudge,
                              hit.hitPosition.z + N.z * nudge, 1.0f);
            ray->direction = L;
        }
    }
    return out;
}

/* :140-163 */
static ray_t setup_ray(const or_camera* cam, int x, int y, int bounceCount, float maxDistance, int aa, uint32_t* seed)
{
    ray_t ray;
    ray.origin = mk4(cam->origin[0], cam->origin[1], cam->origin[2], cam->origin[3]);
    ray.bounceCount = bounceCount;
    ray.maxDistance = maxDistance;
    if (aa) {
        float xf = (float)x + random_value(seed) - 0.5f;
        float yf = (float)y + random_value(seed) - 0.5f;
        ray.direction = tex2d_rays(cam->rays, cam->width, cam->height, xf, yf);
    } else {
        ray.direction = tex2d_rays(cam->rays, cam->width, cam->height, (float)x, (float)y);
    }
    ray.direction = rotate_q(ray.direction, mk4(cam->rotation[0], cam->rotation[1], cam->rotation[2], cam->rotation[3]));
    ray.direction = normalize4(ray.direction);
    return ray;
}

/* :78-138 for one pixel: z lanes 0..3 each sum their samples s = iter*4+z in iter order,
 * then lane 0 adds the four partial sums in z order. */
static void render_pixel(const or_scene* sc, const or_camera* cam, const or_renderer* rnd, const or_sky* sky,
                         uint32_t x, uint32_t y, float* px, path_aov* aov)
{
    float part[OR_BLOCK_DEPTH][3];
    uint32_t spp = rnd->sampleCount;
    for (uint32_t z = 0; z < OR_BLOCK_DEPTH; z++) {
        part[z][0] = part[z][1] = part[z][2] = 0.0f; /* unwritten slots := 0 */
        if (z >= spp) continue;
        int iters = (int)((spp + OR_BLOCK_DEPTH - 1) / OR_BLOCK_DEPTH);
        for (int iter = 0; iter < iters; iter++) {
            uint32_t s = (uint32_t)iter * OR_BLOCK_DEPTH + z;
            if (s >= spp) break;
            uint32_t seed = x + y * (uint32_t)cam->width + s * 34673804u + rnd->frameCount * 719393u;
            ray_t ray = setup_ray(cam, (int)x, (int)y, rnd->bounceCount, rnd->maxDistance, rnd->antiAliasingEnabled, &seed);
            v4 c = trace(&ray, sc, sky, rnd->aoIntensity, &seed, s, aov);
            /* :690-698 exposure per sample */
            c = mk4(1.0f - om_expf(-c.x * cam->exposure), 1.0f - om_expf(-c.y * cam->exposure),
                    1.0f - om_expf(-c.z * cam->exposure), 1.0f);
            part[z][0] += c.x; part[z][1] += c.y; part[z][2] += c.z;
        }
    }
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (int i = 0; i < OR_BLOCK_DEPTH; i++) { r += part[i][0]; g += part[i][1]; b += part[i][2]; }
    r /= (float)spp; g /= (float)spp; b /= (float)spp;
    r = om_fminf(om_fmaxf(r, 0.0f), 1.0f); /* :125-127 */
    g = om_fminf(om_fmaxf(g, 0.0f), 1.0f);
    b = om_fminf(om_fmaxf(b, 0.0f), 1.0f);
    if (rnd->progressiveFrameCount != 0) {
        float n = (float)rnd->progressiveFrameCount;
        float n1 = (float)(rnd->progressiveFrameCount + 1u);
        r = (r + px[0] * n) / n1;
        g = (g + px[1] * n) / n1;
        b = (b + px[2] * n) / n1;
    }
    px[0] = r; px[1] = g; px[2] = b; px[3] = 1.0f;
}

int or_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

uint64_t or_render(const or_scene* scene, const or_camera* cam, const or_renderer* rnd, const or_sky* sky,
                   int x0, int y0, int x1, int y1, float* fb, uint32_t* hitIds, int nThreads)
{
    uint64_t total = 0;
#ifdef _OPENMP
    if (nThreads <= 0) nThreads = omp_get_max_threads();
#else
    nThreads = 1;
#endif
    (void)nThreads;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total) num_threads(nThreads)
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            size_t p = (size_t)y * (size_t)cam->width + (size_t)x;
            path_aov aov;
            aov.primary[0] = aov.primary[1] = aov.primary[2] = 0xFFFFFFFFu;
            aov.checksum = 0; aov.queries = 0;
            render_pixel(scene, cam, rnd, sky, (uint32_t)x, (uint32_t)y, fb + 4 * p, &aov);
            if (hitIds) {
                hitIds[4 * p + 0] = aov.primary[0]; hitIds[4 * p + 1] = aov.primary[1];
                hitIds[4 * p + 2] = aov.primary[2]; hitIds[4 * p + 3] = aov.checksum;
            }
            total += aov.queries;
        }
    }
    return total;
}

void or_trace_rays(const or_scene* scene, const float* rays, int nRays, uint32_t* out, int nThreads)
{
#ifdef _OPENMP
    if (nThreads <= 0) nThreads = omp_get_max_threads();
#else
    nThreads = 1;
#endif
    (void)nThreads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nThreads)
    for (int i = 0; i < nRays; i++) {
        const float* r = rays + 8 * (size_t)i;
        ray_t ray;
        ray.origin = mk4(r[0], r[1], r[2], 1.0f);
        ray.maxDistance = r[3];
        ray.direction = mk4(r[4], r[5], r[6], 0.0f);
        ray.bounceCount = 1;
        rayhit_t h = closest_hit(&ray, scene);
        uint32_t tb; float t = h.distance; memcpy(&tb, &t, 4);
        out[4 * (size_t)i + 0] = h.objectIndex; out[4 * (size_t)i + 1] = h.meshIndex;
        out[4 * (size_t)i + 2] = h.triIndex;    out[4 * (size_t)i + 3] = h.didHit ? tb : 0xFFFFFFFFu;
    }
}

/* ---- unit-test taps ---- */
float or_random_value(uint32_t* seed) { return random_value(seed); }

int or_dist_to_triangle(const float* origin, const float* dir, float maxDistance,
                        const float* v0, const float* v1, const float* v2, float* tOut, float* baryOut)
{
    ray_t ray;
    ray.origin = mk4(origin[0], origin[1], origin[2], 1.0f);
    ray.direction = mk4(dir[0], dir[1], dir[2], 0.0f);
    ray.maxDistance = maxDistance; ray.bounceCount = 1;
    trihit_t h = dist_to_triangle(&ray, mk4(v0[0], v0[1], v0[2], 1.0f), mk4(v1[0], v1[1], v1[2], 1.0f),
                                  mk4(v2[0], v2[1], v2[2], 1.0f));
    *tOut = h.distance; baryOut[0] = h.b0; baryOut[1] = h.b1; baryOut[2] = h.b2;
    return h.distance < FLT_MAX;
}

int or_slab(const float* origin, const float* dir, float maxDistance, const float* invModel,
            const float* bmin, const float* bmax)
{
    ray_t ray;
    ray.origin = mk4(origin[0], origin[1], origin[2], 1.0f);
    ray.direction = mk4(dir[0], dir[1], dir[2], 0.0f);
    ray.maxDistance = maxDistance; ray.bounceCount = 1;
    or_object o;
    memset(&o, 0, sizeof(o));
    for (int i = 0; i < 3; i++) { o.bmin[i] = bmin[i]; o.bmax[i] = bmax[i]; }
    m4 inv = load_m4(invModel);
    return intersects_object_box(&ray, &o, &inv);
}

void or_tex2d(const or_texture* t, float u, float v, float* out4)
{
    v4 r = tex2d_norm(t, u, v);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void or_tex2d_rays(const float* table, int w, int h, float x, float y, float* out4)
{
    v4 r = tex2d_rays(table, w, h, x, y);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void or_sky_light(const float* dir, const or_sky* sky, float* out4)
{
    v4 r = sky_light(mk4(dir[0], dir[1], dir[2], 0.0f), sky);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void or_exposure(const float* rgb, float exposure, float* out3)
{
    out3[0] = 1.0f - om_expf(-rgb[0] * exposure);
    out3[1] = 1.0f - om_expf(-rgb[1] * exposure);
    out3[2] = 1.0f - om_expf(-rgb[2] * exposure);
}
float or_logf(float x) { return om_logf(x); }
float or_cosf(float x) { return om_cosf(x); }
float or_expf(float x) { return om_expf(x); }
float or_powf(float x, float y) { return om_powf(x, y); }
float or_rsqrtf(float x) { return om_rsqrtf(x); }
