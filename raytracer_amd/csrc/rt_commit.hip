// rt_commit.hip -- scene upload for gfx950: flatten every instance into world-space records
// and build the acceleration structure on the GPU.
//
// Replaces the per-frame instance table of Scene::getGPUObjectDataVector (Scene.cpp:3-29) and the
// pointer tables of DisplayObject::sendToGPU (DisplayObject.cpp:30-81).  The reference builds no
// acceleration structure; this file adds one (63-bit Morton codes -> radix sort -> binary tree by PLOC,
// or a Karras radix tree + frontier box fit for fast rebuilds -> top-down collapse into 64-byte four-wide
// nodes with up to kLeafMax triangles per leaf -> leaf-order triangle copies).  World-space positions and normals are produced with the reference's own expressions
// (matVecMul, GPURaytracer.cu:728-743; normals through the transposed inverse, :463-469) so the
// render kernel tests exactly the triangles the reference would.
#include "rt_device.h"
#include "rt_commit.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <rocprim/rocprim.hpp>
#include <float.h>

namespace rt {

// ---- order-preserving float <-> uint (for atomic min / max of bounds) ----
__device__ __forceinline__ uint32_t f2o(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float o2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off));
    return v;
}

// One thread per instance triangle (gid).  Writes triRef[gid], triShade[gid] and folds the
// triangle into the scene bounds (bounds[0..2] = min, [3..5] = max, order-preserving uints).
__global__ __launch_bounds__(256) void flatten_kernel(const Segment* __restrict__ segs, int nSegs,
                                                      const ObjXform* __restrict__ xf, uint32_t nTris,
                                                      TriIsect* __restrict__ triRef, TriShade* __restrict__ triShade,
                                                      uint32_t* __restrict__ bounds)
{
    // grid-stride: a wave folds all its triangles into one set of bounds before it touches the six global
    // words (one atomic per wave and 64 triangles made same-address atomics the whole cost of this kernel
    // on the 13.5 M-triangle scene: 14.4 ms)
    float mnx = FLT_MAX, mny = FLT_MAX, mnz = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX, mxz = -FLT_MAX;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nTris; g += gridDim.x * blockDim.x) {
        // segment = last one with gidStart <= g
        int lo = 0, hi = nSegs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].gidStart <= g) lo = mid; else hi = mid - 1;
        }
        const Segment sg = segs[lo];
        const uint32_t t = g - sg.gidStart;
        const int i0 = sg.idx[3 * t + 0], i1 = sg.idx[3 * t + 1], i2 = sg.idx[3 * t + 2];
        const float* m = xf[sg.obj].model;
        const float* inv = xf[sg.obj].inv;

        float v[3][3];
        const int ids[3] = { i0, i1, i2 };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float x = sg.pos[3 * ids[k] + 0], y = sg.pos[3 * ids[k] + 1], z = sg.pos[3 * ids[k] + 2];
            // matVecMul(model, (x,y,z,1)), :737-742
            v[k][0] = m[0] * x + m[4] * y + m[8] * z + m[12] * 1.0f;
            v[k][1] = m[1] * x + m[5] * y + m[9] * z + m[13] * 1.0f;
            v[k][2] = m[2] * x + m[6] * y + m[10] * z + m[14] * 1.0f;
        }
        TriIsect ti;
        ti.v0x = v[0][0]; ti.v0y = v[0][1]; ti.v0z = v[0][2]; ti.gid = g;
        ti.v1x = v[1][0]; ti.v1y = v[1][1]; ti.v1z = v[1][2]; ti.obj = sg.obj;
        ti.v2x = v[2][0]; ti.v2y = v[2][1]; ti.v2z = v[2][2]; ti.mat = sg.materialId | (sg.nrm ? 0x80000000u : 0u);
        triRef[g] = ti;

        TriShade ts;
        ts.pos = ti;
        float n[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
        if (sg.nrm) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float x = sg.nrm[3 * ids[k] + 0], y = sg.nrm[3 * ids[k] + 1], z = sg.nrm[3 * ids[k] + 2];
                // matVecMul(inverse, (x,y,z,0), transpose = true), :731-735
                n[k][0] = inv[0] * x + inv[1] * y + inv[2] * z + inv[3] * 0.0f;
                n[k][1] = inv[4] * x + inv[5] * y + inv[6] * z + inv[7] * 0.0f;
                n[k][2] = inv[8] * x + inv[9] * y + inv[10] * z + inv[11] * 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { ts.n0[c] = n[0][c]; ts.n1[c] = n[1][c]; ts.n2[c] = n[2][c]; }
        if (sg.uv) {
            ts.uv0[0] = sg.uv[2 * i0]; ts.uv0[1] = sg.uv[2 * i0 + 1];
            ts.uv1[0] = sg.uv[2 * i1]; ts.uv1[1] = sg.uv[2 * i1 + 1];
            ts.uv2[0] = sg.uv[2 * i2]; ts.uv2[1] = sg.uv[2 * i2 + 1];
        } else {
            ts.uv0[0] = ts.uv0[1] = ts.uv1[0] = ts.uv1[1] = ts.uv2[0] = ts.uv2[1] = 0.0f;
        }
        ts.obj = sg.obj; ts.mesh = sg.meshInObj; ts.tri = t;
        ts.matFlags = sg.materialId | (sg.nrm ? 0x80000000u : 0u);
        ts.pad = 0;
        triShade[g] = ts;

#pragma unroll
        for (int k = 0; k < 3; k++) {
            // non-finite vertices (singular transforms) stay out of the bounds
            if (v[k][0] - v[k][0] == 0.0f && v[k][1] - v[k][1] == 0.0f && v[k][2] - v[k][2] == 0.0f) {
                mnx = __builtin_fminf(mnx, v[k][0]); mxx = __builtin_fmaxf(mxx, v[k][0]);
                mny = __builtin_fminf(mny, v[k][1]); mxy = __builtin_fmaxf(mxy, v[k][1]);
                mnz = __builtin_fminf(mnz, v[k][2]); mxz = __builtin_fmaxf(mxz, v[k][2]);
            }
        }
    }
    mnx = wave_min(mnx); mny = wave_min(mny); mnz = wave_min(mnz);
    mxx = wave_max(mxx); mxy = wave_max(mxy); mxz = wave_max(mxz);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(bounds + 0, f2o(mnx)); atomicMin(bounds + 1, f2o(mny)); atomicMin(bounds + 2, f2o(mnz));
        atomicMax(bounds + 3, f2o(mxx)); atomicMax(bounds + 4, f2o(mxy)); atomicMax(bounds + 5, f2o(mxz));
    }
}

// Extended Morton code (after Vaidyanathan et al. 2016): the 63 key bits are not handed to x, y, z
// strictly in turn; an axis whose cell is already thin compared with the longest one sits its turn
// out (build_scene).  A flat scene (a terrain is 400 x 13 x 400 units) otherwise spends every third
// split on slabs that are already thin: C5 32.6 -> 18.0 node visits per ray, C4 19.0 -> 15.6.
// `plan` holds the axis of every bit, most significant first, two bits each.
#ifndef RT_MORTON_THIN
#define RT_MORTON_THIN 0.25f
#endif
struct MortonPlan { unsigned long long lo, hi; };   // bits 0..31 in lo, 32..62 in hi

__device__ __forceinline__ void tri_box(const TriIsect& t, float* mn, float* mx)
{
    mn[0] = __builtin_fminf(t.v0x, __builtin_fminf(t.v1x, t.v2x)); mx[0] = __builtin_fmaxf(t.v0x, __builtin_fmaxf(t.v1x, t.v2x));
    mn[1] = __builtin_fminf(t.v0y, __builtin_fminf(t.v1y, t.v2y)); mx[1] = __builtin_fmaxf(t.v0y, __builtin_fmaxf(t.v1y, t.v2y));
    mn[2] = __builtin_fminf(t.v0z, __builtin_fminf(t.v1z, t.v2z)); mx[2] = __builtin_fmaxf(t.v0z, __builtin_fmaxf(t.v1z, t.v2z));
}

// The box a walk of the tree may prune a triangle by.  For a well-shaped triangle that is its bounding box.  For a sliver it
// is not.  The reference's test (GPURaytracer.cu:551-637) reports t = sum b_i z_i with b_i = e_i / det >= 0 (it rejects mixed
// signs) and z_i = the ray parameter at which the ray reaches vertex i's coordinate along its dominant axis, so t always lies
// in the box's slab along THAT axis.  But the b_i carry the rounding noise of the edge functions: the sheared vertex
// offsets are good to ~6 * 2^-24 D (D = the largest coordinate difference between the ray's origin and the vertices), an
// edge function over a triangle of length L to ~2^-20.4 D L, against det = w L for a triangle of altitude w (both seen in
// projection along the ray, which cancels).  The point of the ray at the reported t is therefore displaced ALONG the ray from
// where it crosses the triangle, by up to that share of the slab's length, and leaves the box sideways by up to
//     L min(1, 2^-20.4 D / w)
// (for a collinear triangle, w = 0, the b_i are all noise and t is anywhere in the slab: the whole length).  A walk that has
// found something nearer in between would prune the box and never run the test whose answer the reference reports
// (thin_kernel below tells how this was found).  So the box a walk sees grows on every side by that amount with D = 2 S
// (S = the scene's largest |coordinate|: a bounce ray starts on a surface of the scene), less the three quarters of the
// ray-side pad of the box test (2^-18 S on every side of every box, make_box_ray) that the box test's own rounding leaves
// unused: nothing for a triangle up to ~4 times as long as it is high, the whole length for a collinear one.  With that
// every traversal finds every hit the reference reports for any ray that starts inside the scene's bounds, by construction
// (tests/thin_scene.py passes with the camera-ray test of wf_gen compiled out, which it did not before); camera rays, which
// may start far outside the bounds, are in addition tested against the listed lines where they are made (wf_gen).
__device__ __forceinline__ void tri_box_walk(const TriIsect& t, float S, float* mn, float* mx)
{
    tri_box(t, mn, mx);
#if defined(RT_DIAG_NO_THIN_ALL) && RT_DIAG_NO_THIN_ALL == 2   // diagnostic build: neither the listed-line tests nor this (what the soak then finds)
    return;
#endif
    const float ax = t.v1x - t.v0x, ay = t.v1y - t.v0y, az = t.v1z - t.v0z;
    const float bx = t.v2x - t.v0x, by = t.v2y - t.v0y, bz = t.v2z - t.v0z;
    const float cx = t.v2x - t.v1x, cy = t.v2y - t.v1y, cz = t.v2z - t.v1z;
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float area2 = sqrtf(nx * nx + ny * ny + nz * nz);
    const float L = sqrtf(fmaxf(fmaxf(ax * ax + ay * ay + az * az, bx * bx + by * by + bz * bz), cx * cx + cy * cy + cz * cz));
    if (!(L > 0.0f) || !(L < FLT_MAX) || !(S > 0.0f)) return;   // no extent, or not finite: never hit
    const float w = area2 / L;                                     // altitude on the longest edge
    const float noise = 1.4453e-06f * S;                           // 2^-20.4 * 2 S
    float r = !(w > noise) ? L : L * (noise / w);                  // NaN area -> L
    r -= 2.86102294921875e-06f * S;                                // 3/4 of the ray-side pad
    if (!(r > 0.0f)) return;
    for (int a = 0; a < 3; a++) { mn[a] -= r; mx[a] += r; }
}

__global__ __launch_bounds__(256) void morton_kernel(const TriIsect* __restrict__ triRef, uint32_t nTris,
                                                     const uint32_t* __restrict__ bounds, MortonPlan plan,
                                                     unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nTris) return;
    const float bx = o2f(bounds[0]), by = o2f(bounds[1]), bz = o2f(bounds[2]);
    const float ex = o2f(bounds[3]) - bx, ey = o2f(bounds[4]) - by, ez = o2f(bounds[5]) - bz;
    float mn[3], mx[3];
    tri_box(triRef[g], mn, mx);
    const float cx = 0.5f * (mn[0] + mx[0]), cy = 0.5f * (mn[1] + mx[1]), cz = 0.5f * (mn[2] + mx[2]);
    float fx = ex > 0.0f ? (cx - bx) / ex : 0.0f, fy = ey > 0.0f ? (cy - by) / ey : 0.0f, fz = ez > 0.0f ? (cz - bz) / ez : 0.0f;
    // 30 bits per axis (with 10 the small triangles of a detailed object in a large scene share one
    // cell and the bottom of the tree degenerates into index-order splits)
    const float kCells = 1073741824.0f;
    uint32_t q[3];
    q[0] = (uint32_t)__builtin_fminf(__builtin_fmaxf(fx * kCells, 0.0f), kCells - 64.0f); // NaN -> 0
    q[1] = (uint32_t)__builtin_fminf(__builtin_fmaxf(fy * kCells, 0.0f), kCells - 64.0f);
    q[2] = (uint32_t)__builtin_fminf(__builtin_fmaxf(fz * kCells, 0.0f), kCells - 64.0f);
    int used[3] = { 0, 0, 0 };
    unsigned long long key = 0;
#pragma unroll 1
    for (int b = 0; b < 63; b++) {
        const int ax = (int)(((b < 32 ? plan.lo >> (2 * b) : plan.hi >> (2 * (b - 32)))) & 3ull);
        const uint32_t qa = ax == 0 ? q[0] : (ax == 1 ? q[1] : q[2]);
        const int u = ax == 0 ? used[0] : (ax == 1 ? used[1] : used[2]);
        key = (key << 1) | ((qa >> (29 - u)) & 1u);
        used[0] += ax == 0; used[1] += ax == 1; used[2] += ax == 2;
    }
    keys[g] = key;
    vals[g] = g;
}

// Triangles of (nearly) no area.  The reference's watertight test decides by the signs of three edge functions computed in
// binary32; when the ray passes within rounding of the LINE through a collinear triangle, all three are rounding noise and a
// hit is reported a third of the time when the point where the ray crosses the line lies inside the triangle's span (never
// when it does not: 0 of 54 000 such rays, measured with the oracle).  The distance reported is a mean of the three vertex
// depths under weights of rounding noise: anywhere along the sliver, up to 1e-2 outside the stretch of the ray that the
// sliver's bounding box holds, so a walk that has found something nearer or an overlapping sliver first prunes the box and
// never tests the triangle (tests/soak_parity.py: objects with two axes scaled to zero, rays aimed at their vertices: 6 of
// 18 M ray queries returned a farther hit; the reference's own sword model holds ten such triangles).  An edge function is p_a x p_b with an error of
// 2^-22 |p_a||p_b|, so a violated edge decides correctly unless it subtends less than 2^-22 rad at the crossing point; for a
// triangle of altitude w seen from up to 2 S away (S = the scene's largest |coordinate|) that needs w < 2^-20 S.  Triangles
// with altitude below 2^-17 S are listed here, as the line they lie on, and the walks of the exact paths (bvh_walk: every
// rt_trace_rays query, the redo launches, wf_exact) test them after the tree, behind a cheap test of the ray's distance to that
// line (rt_trace.h, thin_near).  Camera rays -- the rays a scene can aim at such a line by construction, from any distance --
// are tested against every listed line where they are made (wf_gen, path_kernel) and sent to the exact walks when they come
// near (tests/thin_scene.py aims them on purpose).  Since round 3 the tree itself also stands for these triangles: the box a
// walk sees for a sliver is grown by what the noise can displace the reported hit (tri_box_walk above: the whole length for
// a collinear triangle), so that every walk -- the hot kernels' too, for camera and bounce rays alike -- reaches the test
// whose answer the reference reports; the listed-line tests stay as the cover for rays that start far outside the scene's
// bounds.  The triangles stay in the tree as well (testing one twice changes nothing).  A triangle of no extent at all, or
// with a vertex that is not finite, is never hit (all three edge functions are exactly 0, or NaN) and is not listed.
__global__ __launch_bounds__(256) void thin_kernel(const TriIsect* __restrict__ triRef, uint32_t nTris, float tau,
                                                   uint32_t* __restrict__ count, ThinTri* __restrict__ list)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nTris) return;
    const TriIsect t = triRef[g];
    const float ax = t.v1x - t.v0x, ay = t.v1y - t.v0y, az = t.v1z - t.v0z;   // v0 -> v1
    const float bx = t.v2x - t.v0x, by = t.v2y - t.v0y, bz = t.v2z - t.v0z;   // v0 -> v2
    const float cx = t.v2x - t.v1x, cy = t.v2y - t.v1y, cz = t.v2z - t.v1z;   // v1 -> v2
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float area2 = sqrtf(nx * nx + ny * ny + nz * nz);
    const float la = sqrtf(ax * ax + ay * ay + az * az), lb = sqrtf(bx * bx + by * by + bz * bz), lc = sqrtf(cx * cx + cy * cy + cz * cz);
    const float longest = fmaxf(fmaxf(la, lb), lc);
    if (!(longest > 0.0f) || !(longest < FLT_MAX)) return;
    if (!(area2 / longest < tau)) return;
    ThinTri r;
    r.gid = g; r.k = 3.814697265625e-06f / longest; r.len2 = longest * longest; r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    if (lc >= la && lc >= lb) { r.px = t.v1x; r.py = t.v1y; r.pz = t.v1z; r.ux = cx / lc; r.uy = cy / lc; r.uz = cz / lc; }
    else if (lb >= la) { r.px = t.v0x; r.py = t.v0y; r.pz = t.v0z; r.ux = bx / lb; r.uy = by / lb; r.uz = bz / lb; }
    else { r.px = t.v0x; r.py = t.v0y; r.pz = t.v0z; r.ux = ax / la; r.uy = ay / la; r.uz = az / la; }
    list[atomicAdd(count, 1u)] = r;
}

// boxes of the triangles in sorted (Morton) order: "node" N-1+j of the binary tree is sorted triangle j
__global__ __launch_bounds__(256) void leafbox_kernel(const TriIsect* __restrict__ triRef, const uint32_t* __restrict__ order,
                                                      uint32_t nTris, float sceneMaxAbs, float* __restrict__ boxes)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nTris) return;
    const TriIsect t = triRef[order[j]];
    float mn[3], mx[3];
    tri_box_walk(t, sceneMaxAbs, mn, mx);
    float* b = boxes + 6 * (size_t)(nTris - 1 + j);
    b[0] = mn[0]; b[1] = mn[1]; b[2] = mn[2]; b[3] = mx[0]; b[4] = mx[1]; b[5] = mx[2];
}

// leaf-order copies of the intersection records: sorted triangle j goes to slot pos[j], the place the
// collapse gave it (the leaves of the finished tree, left to right, so that a leaf is a run of slots)
__global__ __launch_bounds__(256) void gather_kernel(const TriIsect* __restrict__ triRef, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ pos, uint32_t nTris,
                                                     TriIsect* __restrict__ triLeaf)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nTris) return;
    const TriIsect t = triRef[order[j]];
    const size_t d = pos[j];
    // three copies, rotated for rays whose dominant axis kz is 0, 1, 2: (x,y,z) -> (kz+1, kz+2, kz) mod 3
    TriIsect r = t;
    r.v0x = t.v0y; r.v0y = t.v0z; r.v0z = t.v0x; r.v1x = t.v1y; r.v1y = t.v1z; r.v1z = t.v1x; r.v2x = t.v2y; r.v2y = t.v2z; r.v2z = t.v2x;
    triLeaf[d] = r;
    r.v0x = t.v0z; r.v0y = t.v0x; r.v0z = t.v0y; r.v1x = t.v1z; r.v1y = t.v1x; r.v1z = t.v1y; r.v2x = t.v2z; r.v2y = t.v2x; r.v2z = t.v2y;
    triLeaf[(size_t)nTris + d] = r;
    triLeaf[2 * (size_t)nTris + d] = t;
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees":
// internal node i of the binary radix tree over the sorted keys (ties broken by position).
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ __launch_bounds__(256) void karras_kernel(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ children,
                                                     int* __restrict__ count, int* __restrict__ parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    // child encoding here: >= 0 internal, < 0 leaf (~leaf position)
    const int left = (first == gamma) ? ~gamma : gamma;
    const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    children[i] = make_int2(left, right);
    count[i] = last - first + 1;
    // parent of internal node c: parent[c]; of leaf j: parent[n - 1 + j] (the numbering of `boxes`)
    parent[left < 0 ? n - 1 + (~left) : left] = i;
    parent[right < 0 ? n - 1 + (~right) : right] = i;
}

// Bottom-up box fit over frontiers.  A node enters the next frontier when its second child has been
// fitted (an arrival counter per node); its own box is computed by the next launch, so every box a
// launch reads was written by an earlier launch (kernel boundaries give the cross-XCD visibility, no
// in-kernel hand-off; MI355X_MICROARCH.md: the per-XCD L2s are not coherent).  Total work is O(n);
// the level-synchronous version this replaces rescanned every node once per tree level (37 passes
// over 13.5 M nodes: 26 ms of a 38 ms commit).
__device__ __forceinline__ void frontier_arrive(int node, const int* __restrict__ parent, uint32_t* __restrict__ arrive,
                                                int* __restrict__ next, uint32_t* __restrict__ nextCount, bool active)
{
    int p = -1;
    if (active) {
        p = parent[node];
        if (atomicAdd(&arrive[p], 1u) != 1u) p = -1; // the first child to arrive leaves the parent to the second
    }
    // one append atomic per wave
    const unsigned long long m = __ballot(p >= 0);
    if (m != 0ull) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(nextCount, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __builtin_ctzll(m));
        if (p >= 0) next[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
    }
}

__global__ __launch_bounds__(256) void fit_leaves_kernel(const int* __restrict__ parent, int nLeaves, uint32_t* __restrict__ arrive,
                                                         int* __restrict__ next, uint32_t* __restrict__ nextCount)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    frontier_arrive(nLeaves - 1 + j, parent, arrive, next, nextCount, j < nLeaves);
}

__global__ __launch_bounds__(256) void fit_frontier_kernel(const int2* __restrict__ children, const int* __restrict__ parent, int nLeaves,
                                                           float* __restrict__ boxes, uint32_t* __restrict__ arrive,
                                                           const int* __restrict__ cur, uint32_t curCount, int* __restrict__ next,
                                                           uint32_t* __restrict__ nextCount)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    int i = -1;
    if (t < curCount) {
        i = cur[t];
        const int2 c = children[i];
        const float* a = boxes + 6 * (size_t)(c.x < 0 ? nLeaves - 1 + (~c.x) : c.x);
        const float* b = boxes + 6 * (size_t)(c.y < 0 ? nLeaves - 1 + (~c.y) : c.y);
        float* o = boxes + 6 * (size_t)i;
        o[0] = __builtin_fminf(a[0], b[0]); o[1] = __builtin_fminf(a[1], b[1]); o[2] = __builtin_fminf(a[2], b[2]);
        o[3] = __builtin_fmaxf(a[3], b[3]); o[4] = __builtin_fmaxf(a[4], b[4]); o[5] = __builtin_fmaxf(a[5], b[5]);
    }
    frontier_arrive(i, parent, arrive, next, nextCount, i > 0); // node 0 is the root
}

// ---- collapse of the binary tree into four-wide nodes, top down, one launch per level ----
// A task = (binary internal node, index of the four-wide node to write).  The node's children are
// the binary node's two children, with the larger (by surface area) internal ones replaced by
// their own children until four slots are used.  A binary subtree of at most kLeafMax triangles
// becomes a leaf (its triangles are contiguous in leaf order).
__device__ __forceinline__ int tris_under(int c, const int* __restrict__ count) { return c < 0 ? 1 : count[c]; }
__device__ __forceinline__ bool is_leaf_subtree(int c, const int* __restrict__ count) { return tris_under(c, count) <= kLeafMax; }

__device__ __forceinline__ const float* box_of(int c, const float* __restrict__ boxes, int nLeaves)
{
    return boxes + 6 * (size_t)(c < 0 ? nLeaves - 1 + (~c) : c);
}

// scale of a node along one axis: the smallest power of two with extent / scale <= 250 (five steps of slack for the
// outward rounding of the child planes)
__device__ __forceinline__ float node_scale(float ext)
{
    int e;
    const float m = frexpf(ext * (1.0f / 250.0f), &e);   // ext/250 = m * 2^e, m in [0.5, 1)
    e = (ext > 0.0f) ? (m == 0.5f ? e - 1 : e) : -120;
    if (e < -120) e = -120;
    return ldexpf(1.0f, e);
}
// child planes as 8-bit offsets from the node's origin, lower planes rounded down, upper planes rounded up
__device__ __forceinline__ void quantise_planes(float lo, float hi, float origin, float sc, int& q0, int& q1)
{
    q0 = (int)floorf((lo - origin) / sc);
    q0 = q0 < 0 ? 0 : (q0 > 255 ? 255 : q0);
    while (q0 > 0 && origin + (float)q0 * sc > lo) q0--;
    q1 = (int)ceilf((hi - origin) / sc);
    q1 = q1 < 0 ? 0 : (q1 > 255 ? 255 : q1);
    while (q1 < 255 && origin + (float)q1 * sc < hi) q1++;
}

__global__ __launch_bounds__(256) void collapse_kernel(const int2* __restrict__ children, const int* __restrict__ count,
                                                       const float* __restrict__ boxes, int nLeaves,
                                                       const int2* __restrict__ qIn, const uint32_t* __restrict__ firstIn, uint32_t qCount,
                                                       int2* __restrict__ qOut, uint32_t* __restrict__ firstOut, uint32_t* __restrict__ qOutCount,
                                                       uint32_t* __restrict__ nodeCount, BvhNode* __restrict__ nodes,
                                                       uint32_t* __restrict__ pos)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qCount) return;
    const int2 task = qIn[i];
    uint32_t first = firstIn[i]; // leaf slot of the subtree's leftmost triangle
    int c0, c1, c2 = 0, c3 = 0, n = 2;
    { const int2 c = children[task.x]; c0 = c.x; c1 = c.y; }
    for (int iter = 0; iter < 2; iter++) {
        int bestK = -1; float bestA = -1.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= n) break;
            const int c = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
            if (is_leaf_subtree(c, count)) continue;
            const float* b = boxes + 6 * (size_t)c;
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            float A = dx * dy + dy * dz + dz * dx;
            if (!(A >= 0.0f)) A = 0.0f; // NaN boxes (non-finite vertices) still get expanded
            if (A > bestA) { bestA = A; bestK = k; }
        }
        if (bestK < 0) break;
        const int e = bestK == 0 ? c0 : (bestK == 1 ? c1 : (bestK == 2 ? c2 : c3));
        const int2 cc = children[e];
        // replace slot bestK by the two children, keeping the left-to-right order of the binary tree
        if (bestK == 0) { c3 = c2; c2 = c1; c0 = cc.x; c1 = cc.y; }
        else if (bestK == 1) { c3 = c2; c1 = cc.x; c2 = cc.y; }
        else { c2 = cc.x; c3 = cc.y; }
        n++;
    }
    const int cs[4] = { c0, c1, c2, c3 };
    int m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < n && !is_leaf_subtree(cs[k], count)) m++;
    uint32_t base = 0, qbase = 0;
    if (m > 0) { base = atomicAdd(nodeCount, (uint32_t)m); qbase = atomicAdd(qOutCount, (uint32_t)m); }
    // node box = union of the child boxes, clamped so that the extents stay finite
    const float kBig = 1e30f;
    float lo[4][3], hi[4][3];
    float nlo[3] = { kBig, kBig, kBig }, nhi[3] = { -kBig, -kBig, -kBig };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k >= n) break;
        const float* b = box_of(cs[k], boxes, nLeaves);
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[k][a] = __builtin_fminf(__builtin_fmaxf(b[a], -kBig), kBig);       // NaN -> -kBig / +kBig: covers everything
            hi[k][a] = __builtin_fmaxf(__builtin_fminf(b[3 + a], kBig), -kBig);
            nlo[a] = __builtin_fminf(nlo[a], lo[k][a]);
            nhi[a] = __builtin_fmaxf(nhi[a], hi[k][a]);
        }
    }
    BvhNode nd;
    nd.ox = nlo[0]; nd.oy = nlo[1]; nd.oz = nlo[2];
    float sc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) sc[a] = node_scale(nhi[a] - nlo[a]);
    nd.sx = sc[0]; nd.sy = sc[1]; nd.sz = sc[2];
    uint32_t ql[3] = { 0, 0, 0 }, qh[3] = { 0, 0, 0 };
    int j = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < n) {
            const int c = cs[k];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                int q0, q1;
                quantise_planes(lo[k][a], hi[k][a], nlo[a], sc[a], q0, q1);
                ql[a] |= (uint32_t)q0 << (8 * k);
                qh[a] |= (uint32_t)q1 << (8 * k);
            }
            const int under = tris_under(c, count);
            if (under <= kLeafMax) {
                // a leaf: its triangles take the slots first .. first + under - 1, left to right
                nd.child[k] = ~(int32_t)((first << 3) | (uint32_t)under);
                int stk[8]; int sp = 0; stk[sp++] = c;
                uint32_t p = first;
                while (sp > 0) {
                    const int x = stk[--sp];
                    if (x < 0) pos[~x] = p++;
                    else { const int2 cc = children[x]; stk[sp++] = cc.y; stk[sp++] = cc.x; }
                }
            } else {
                nd.child[k] = (int32_t)(base + (uint32_t)j);
                qOut[qbase + (uint32_t)j] = make_int2(c, (int)(base + (uint32_t)j));
                firstOut[qbase + (uint32_t)j] = first;
                j++;
            }
            first += (uint32_t)under;
        } else {
            // unused slot: inverted box, empty leaf
#pragma unroll
            for (int a = 0; a < 3; a++) ql[a] |= 255u << (8 * k);
            nd.child[k] = ~(int32_t)0;
        }
    }
    nd.lox = ql[0]; nd.loy = ql[1]; nd.loz = ql[2];
    nd.hix = qh[0]; nd.hiy = qh[1]; nd.hiz = qh[2];
    nodes[task.y] = nd;
}

// single-triangle scene: one node whose first slot is the triangle's leaf
__global__ void emit_single_kernel(const float* __restrict__ boxes, BvhNode* __restrict__ nodes)
{
    const float kBig = 1e30f;
    BvhNode nd;
    float lo[3], hi[3], sc[3];
    uint32_t qh[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = __builtin_fminf(__builtin_fmaxf(boxes[a], -kBig), kBig);
        hi[a] = __builtin_fmaxf(__builtin_fminf(boxes[3 + a], kBig), -kBig);
        const float ext = hi[a] - lo[a];
        int e;
        const float m = frexpf(ext * (1.0f / 250.0f), &e);
        e = (ext > 0.0f) ? (m == 0.5f ? e - 1 : e) : -120;
        if (e < -120) e = -120;
        sc[a] = ldexpf(1.0f, e);
        int q1 = (int)ceilf(ext / sc[a]);
        q1 = q1 < 0 ? 0 : (q1 > 255 ? 255 : q1);
        while (q1 < 255 && lo[a] + (float)q1 * sc[a] < hi[a]) q1++;
        qh[a] = (uint32_t)q1;
    }
    nd.ox = lo[0]; nd.oy = lo[1]; nd.oz = lo[2];
    nd.sx = sc[0]; nd.sy = sc[1]; nd.sz = sc[2];
    nd.lox = nd.loy = nd.loz = 0xFFFFFF00u;
    nd.hix = qh[0]; nd.hiy = qh[1]; nd.hiz = qh[2];
    nd.child[0] = ~(int32_t)1;   // first 0, count 1
    nd.child[1] = nd.child[2] = nd.child[3] = ~(int32_t)0;
    nodes[0] = nd;
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ----
// The quality builder.  The clusters start as the triangles in Morton order; every pass each cluster
// looks kPlocRadius places left and right for the neighbour whose union with it has the smallest
// surface area, mutual choices merge into a new binary node, and the survivors are compacted in order.
// Measured against the radix tree above (same collapse, same traversal): see DESIGN.md section 4.
// A cluster is named like a child: >= 0 internal node, < 0 the sorted triangle ~c.
#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = RT_PLOC_RADIUS;

__global__ __launch_bounds__(256) void ploc_init_kernel(int* __restrict__ cl, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cl[i] = ~i;
}

// nn[i] = the neighbour within the radius that minimises (area of the union, distance in places, parity of
// the lower place, lower place): a total order on pairs that both ends see alike, so the best pair overall
// is always mutual and every pass merges at least once.  Non-finite areas count as FLT_MAX.
__global__ __launch_bounds__(256) void ploc_nn_kernel(const int* __restrict__ cl, int m, const float* __restrict__ boxes, int nLeaves,
                                                      int* __restrict__ nn)
{
    __shared__ float sb[6][256 + 2 * kPlocRadius];
    const int base = (int)(blockIdx.x * 256u) - kPlocRadius;
    for (int t = threadIdx.x; t < 256 + 2 * kPlocRadius; t += 256) {
        const int g = base + t;
        if (g >= 0 && g < m) {
            const float* b = box_of(cl[g], boxes, nLeaves);
#pragma unroll
            for (int k = 0; k < 6; k++) sb[k][t] = b[k];
        }
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * 256u) + (int)threadIdx.x;
    if (i >= m) return;
    const int ti = (int)threadIdx.x + kPlocRadius;
    const float a0 = sb[0][ti], a1 = sb[1][ti], a2 = sb[2][ti], a3 = sb[3][ti], a4 = sb[4][ti], a5 = sb[5][ti];
    float bestA = 0.0f; int bestJ = -1;
    for (int d = -kPlocRadius; d <= kPlocRadius; d++) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= m) continue;
        const int tj = ti + d;
        const float dx = __builtin_fmaxf(a3, sb[3][tj]) - __builtin_fminf(a0, sb[0][tj]);
        const float dy = __builtin_fmaxf(a4, sb[4][tj]) - __builtin_fminf(a1, sb[1][tj]);
        const float dz = __builtin_fmaxf(a5, sb[5][tj]) - __builtin_fminf(a2, sb[2][tj]);
        float A = dx * dy + dy * dz + dz * dx;
        if (!(A < FLT_MAX)) A = FLT_MAX;
        bool take = bestJ < 0 || A < bestA;
        if (!take && A == bestA) {
            // equal areas (coincident or gridded geometry): the closer place first, then pairs that start on an
            // even place, then the lower pair -- with everything equal the places pair up (0,1), (2,3), ... and the
            // pass halves the clusters instead of merging one pair
            const int d1 = d < 0 ? -d : d, d2 = bestJ < i ? i - bestJ : bestJ - i;
            const int lo1 = i < j ? i : j, lo2 = i < bestJ ? i : bestJ;
            take = d1 < d2 || (d1 == d2 && ((lo1 & 1) < (lo2 & 1) || ((lo1 & 1) == (lo2 & 1) && lo1 < lo2)));
        }
        if (take) { bestA = A; bestJ = j; }
    }
    nn[i] = bestJ;
}

// flags[i]: low word 1 if cluster i survives the pass (alone, or as the lower end of a merging pair),
// high word 1 if it is that lower end (a new node is created in its place)
__global__ __launch_bounds__(256) void ploc_flag_kernel(const int* __restrict__ nn, int m, unsigned long long* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    flags[i] = mutual ? (i < j ? 0x100000001ull : 0ull) : 1ull;
}

__global__ __launch_bounds__(256) void ploc_merge_kernel(const int* __restrict__ cl, const int* __restrict__ nn, int m,
                                                         const unsigned long long* __restrict__ flags,
                                                         const unsigned long long* __restrict__ prefix, int nodeBase, int nLeaves,
                                                         int2* __restrict__ children, int* __restrict__ count, float* __restrict__ boxes,
                                                         int* __restrict__ out, uint32_t* __restrict__ totals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i], pre = prefix[i];
    if (i == m - 1) { totals[0] = (uint32_t)(pre + f); totals[1] = (uint32_t)((pre + f) >> 32); }
    if ((f & 1ull) == 0ull) return;
    int c = cl[i];
    if (f >> 32) {
        const int d = cl[nn[i]];
        const int id = nodeBase + (int)(pre >> 32);
        const float* a = box_of(c, boxes, nLeaves);
        const float* b = box_of(d, boxes, nLeaves);
        float* o = boxes + 6 * (size_t)id;
        o[0] = __builtin_fminf(a[0], b[0]); o[1] = __builtin_fminf(a[1], b[1]); o[2] = __builtin_fminf(a[2], b[2]);
        o[3] = __builtin_fmaxf(a[3], b[3]); o[4] = __builtin_fmaxf(a[4], b[4]); o[5] = __builtin_fmaxf(a[5], b[5]);
        children[id] = make_int2(c, d); // the lower-placed cluster stays on the left
        count[id] = tris_under(c, count) + tris_under(d, count);
        c = id;
    }
    out[(uint32_t)pre] = c;
}

// ---- refit: the tree's topology stays, its boxes follow the triangles (refit_scene) ----
// the leaf-order copies again: slot d keeps its triangle (the unrotated copy names it), whose record has moved
__global__ __launch_bounds__(256) void regather_kernel(const TriIsect* __restrict__ triRef, uint32_t nTris, TriIsect* __restrict__ triLeaf)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nTris) return;
    const TriIsect t = triRef[triLeaf[2 * (size_t)nTris + d].gid];
    TriIsect r = t;
    r.v0x = t.v0y; r.v0y = t.v0z; r.v0z = t.v0x; r.v1x = t.v1y; r.v1y = t.v1z; r.v1z = t.v1x; r.v2x = t.v2y; r.v2y = t.v2z; r.v2z = t.v2x;
    triLeaf[d] = r;
    r.v0x = t.v0z; r.v0y = t.v0x; r.v0z = t.v0y; r.v1x = t.v1z; r.v1y = t.v1x; r.v1z = t.v1y; r.v2x = t.v2z; r.v2y = t.v2x; r.v2z = t.v2y;
    triLeaf[(size_t)nTris + d] = r;
    triLeaf[2 * (size_t)nTris + d] = t;
}

// One level of the four-wide tree, deepest first: a node's child boxes come from the triangles of its leaves and from
// the boxes the level below has just written (float boxes kept beside the nodes for the duration: fitting from the decoded
// 8-bit planes would widen every level by its rounding), then the node is encoded as the collapse encodes it.
__global__ __launch_bounds__(256) void refit_level_kernel(BvhNode* __restrict__ nodes, float* __restrict__ nodeBox,
                                                          const TriIsect* __restrict__ triWorld, float sceneMaxAbs, uint32_t begin, uint32_t end)
{
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    BvhNode nd = nodes[i];
    const float kBig = 1e30f;
    float lo[4][3], hi[4][3];
    bool used[4];
    float nlo[3] = { kBig, kBig, kBig }, nhi[3] = { -kBig, -kBig, -kBig };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t c = nd.child[k];
        float b[6];
        used[k] = true;
        if (c >= 0) {
#pragma unroll
            for (int a = 0; a < 6; a++) b[a] = nodeBox[6 * (size_t)c + a];
        } else {
            const uint32_t code = (uint32_t)(~c), first = code >> 3, cnt = code & 7u;
            used[k] = cnt != 0u;
            b[0] = b[1] = b[2] = FLT_MAX; b[3] = b[4] = b[5] = -FLT_MAX;
            for (uint32_t t = 0; t < cnt; t++) {
                float mn[3], mx[3];
                tri_box_walk(triWorld[first + t], sceneMaxAbs, mn, mx);
#pragma unroll
                for (int a = 0; a < 3; a++) { b[a] = __builtin_fminf(b[a], mn[a]); b[3 + a] = __builtin_fmaxf(b[3 + a], mx[a]); }
            }
        }
        if (!used[k]) continue;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[k][a] = __builtin_fminf(__builtin_fmaxf(b[a], -kBig), kBig);       // NaN -> -kBig / +kBig: covers everything
            hi[k][a] = __builtin_fmaxf(__builtin_fminf(b[3 + a], kBig), -kBig);
            nlo[a] = __builtin_fminf(nlo[a], lo[k][a]);
            nhi[a] = __builtin_fmaxf(nhi[a], hi[k][a]);
        }
    }
    float sc[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { sc[a] = node_scale(nhi[a] - nlo[a]); nodeBox[6 * (size_t)i + a] = nlo[a]; nodeBox[6 * (size_t)i + 3 + a] = nhi[a]; }
    nd.ox = nlo[0]; nd.oy = nlo[1]; nd.oz = nlo[2];
    nd.sx = sc[0]; nd.sy = sc[1]; nd.sz = sc[2];
    uint32_t ql[3] = { 0, 0, 0 }, qh[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (used[k]) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                int q0, q1;
                quantise_planes(lo[k][a], hi[k][a], nlo[a], sc[a], q0, q1);
                ql[a] |= (uint32_t)q0 << (8 * k);
                qh[a] |= (uint32_t)q1 << (8 * k);
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; a++) ql[a] |= 255u << (8 * k);   // unused slot: inverted box
        }
    }
    nd.lox = ql[0]; nd.loy = ql[1]; nd.loz = ql[2];
    nd.hix = qh[0]; nd.hiy = qh[1]; nd.hiz = qh[2];
    nodes[i] = nd;
}

// summed surface area of the node boxes (as encoded), the measure of how far a refitted tree has degraded
__global__ __launch_bounds__(256) void node_area_kernel(const BvhNode* __restrict__ nodes, uint32_t nNodes, float* __restrict__ acc)
{
    float sum = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nNodes; i += gridDim.x * blockDim.x) {
        const BvhNode nd = nodes[i];
        uint32_t lo[3] = { 255u, 255u, 255u }, hi[3] = { 0u, 0u, 0u };
        const uint32_t ql[3] = { nd.lox, nd.loy, nd.loz }, qh[3] = { nd.hix, nd.hiy, nd.hiz };
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const uint32_t l = (ql[a] >> (8 * k)) & 255u, h = (qh[a] >> (8 * k)) & 255u;
                if (l <= h) { lo[a] = l < lo[a] ? l : lo[a]; hi[a] = h > hi[a] ? h : hi[a]; }
            }
        const float dx = lo[0] <= hi[0] ? (float)(hi[0] - lo[0]) * nd.sx : 0.0f, dy = lo[1] <= hi[1] ? (float)(hi[1] - lo[1]) * nd.sy : 0.0f,
                    dz = lo[2] <= hi[2] ? (float)(hi[2] - lo[2]) * nd.sz : 0.0f;
        const float A = dx * dy + dy * dz + dz * dx;
        if (A < 1e30f) sum += A;   // boxes blown up by non-finite vertices count for nothing
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0 && sum != 0.0f) atomicAdd(acc, sum);
}

#define RT_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return false; }     \
    } while (0)

static inline uint32_t host_f2o(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

bool build_scene(const Segment* dSegs, int nSegs, const ObjXform* dXf, uint32_t nTris, BuiltScene& out, int builder,
                 hipStream_t stream, std::string& err)
{
    out.nTris = nTris;
    out.nNodes = 0;
    out.sceneMaxAbs = 0.0f;
    out.nThin = 0;
    out.levelBegin.clear();
    out.buildArea = out.lastArea = 0.0f;
    if (nTris == 0) return true;
    const uint32_t grid = (nTris + 255) / 256;
    const uint32_t nInternal = nTris > 1 ? nTris - 1 : 1;

    if (nTris > out.capTris) {
        hipFree(out.triRef); hipFree(out.triLeaf); hipFree(out.triShade); hipFree(out.nodes);
        out.triRef = nullptr; out.triLeaf = nullptr; out.triShade = nullptr; out.nodes = nullptr; out.capTris = 0;
        RT_HIP(hipMalloc((void**)&out.triRef, sizeof(TriIsect) * (size_t)nTris));
        RT_HIP(hipMalloc((void**)&out.triLeaf, 3 * sizeof(TriIsect) * (size_t)nTris));
        RT_HIP(hipMalloc((void**)&out.triShade, sizeof(TriShade) * (size_t)nTris));
        RT_HIP(hipMalloc((void**)&out.nodes, sizeof(BvhNode) * (size_t)nInternal));
        out.capTris = nTris;
    }
    out.nNodes = nInternal;

    // scratch: carved out of one arena that survives the commit (a re-commit of a scene of the same
    // size allocates nothing)
    uint32_t *dBounds = nullptr, *valsA = nullptr, *valsB = nullptr, *done = nullptr, *remaining = nullptr, *thinCount = nullptr;
    ThinTri* thinTmp = nullptr;
    unsigned long long *keysA = nullptr, *keysB = nullptr;
    int2 *queueA = nullptr, *queueB = nullptr;
    float* boxes = nullptr;
    int2* children = nullptr;
    int *count = nullptr, *parent = nullptr;
    void* sortTmp = nullptr;
    auto cleanup = [&]() {};
#define RT_HIPC(call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return false; } \
    } while (0)

    size_t tmpBytes = 0, scanBytes = 0;
    RT_HIPC(rocprim::radix_sort_pairs(nullptr, tmpBytes, keysA, keysB, valsA, valsB, (size_t)nTris, 0, 63, stream));
    RT_HIPC(rocprim::exclusive_scan(nullptr, scanBytes, keysA, keysB, 0ull, (size_t)nTris, rocprim::plus<unsigned long long>(), stream));
    const size_t sortBytes = tmpBytes;
    if (scanBytes > tmpBytes) tmpBytes = scanBytes;
    {
        const size_t n = nTris, ni = nInternal;
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t oBounds = take(6 * sizeof(uint32_t)), oKeysA = take(8 * n), oKeysB = take(8 * n), oValsA = take(4 * n),
                     oValsB = take(4 * n), oSort = take(tmpBytes ? tmpBytes : 16), oBoxes = take(sizeof(float) * 6 * 2 * n),
                     oChildren = take(sizeof(int2) * ni), oCount = take(sizeof(int) * ni), oDone = take(4 * ni),
                     oRemaining = take(4), oThinCount = take(4), oArea = take(4), oThin = take(sizeof(ThinTri) * n), oParent = take(sizeof(int) * (2 * n)), oQueueA = take(sizeof(int2) * ni), oQueueB = take(sizeof(int2) * ni);
        if (off > out.arenaBytes) {
            hipFree(out.arena); out.arena = nullptr; out.arenaBytes = 0;
            RT_HIP(hipMalloc(&out.arena, off));
            out.arenaBytes = off;
        }
        char* base = (char*)out.arena;
        dBounds = (uint32_t*)(base + oBounds); keysA = (unsigned long long*)(base + oKeysA); keysB = (unsigned long long*)(base + oKeysB);
        valsA = (uint32_t*)(base + oValsA); valsB = (uint32_t*)(base + oValsB); sortTmp = base + oSort; boxes = (float*)(base + oBoxes);
        children = (int2*)(base + oChildren); count = (int*)(base + oCount); done = (uint32_t*)(base + oDone);
        thinCount = (uint32_t*)(base + oThinCount); thinTmp = (ThinTri*)(base + oThin);
        remaining = (uint32_t*)(base + oRemaining); parent = (int*)(base + oParent); queueA = (int2*)(base + oQueueA); queueB = (int2*)(base + oQueueB);
        out.sBounds = dBounds; out.sThinCount = thinCount; out.sThinTmp = thinTmp; out.sBoxes = boxes; out.sArea = (float*)(base + oArea);
    }

    const uint32_t hb[6] = { host_f2o(FLT_MAX), host_f2o(FLT_MAX), host_f2o(FLT_MAX),
                             host_f2o(-FLT_MAX), host_f2o(-FLT_MAX), host_f2o(-FLT_MAX) };
    RT_HIPC(hipMemcpyAsync(dBounds, hb, sizeof(hb), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(flatten_kernel, dim3(grid < 4096u ? grid : 4096u), dim3(256), 0, stream, dSegs, nSegs, dXf, nTris, out.triRef,
                       out.triShade, dBounds);
    RT_HIPC(hipGetLastError());

    uint32_t hbOut[6];
    RT_HIPC(hipMemcpyAsync(hbOut, dBounds, sizeof(hbOut), hipMemcpyDeviceToHost, stream));
    RT_HIPC(hipStreamSynchronize(stream));
    float maxAbs = 0.0f;
    for (int k = 0; k < 6; k++) {
        const float f = o2f(hbOut[k]);
        if (f < FLT_MAX && f > -FLT_MAX && fabsf(f) > maxAbs) maxAbs = fabsf(f);
    }
    out.sceneMaxAbs = maxAbs;

    // the triangles no box can stand for (thin_kernel)
    RT_HIPC(hipMemsetAsync(thinCount, 0, 4, stream));
    hipLaunchKernelGGL(thin_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, nTris, maxAbs * 7.62939453125e-06f, thinCount, thinTmp);
    RT_HIPC(hipGetLastError());
    {
        uint32_t n = 0;
        RT_HIPC(hipMemcpyAsync(&n, thinCount, 4, hipMemcpyDeviceToHost, stream));
        RT_HIPC(hipStreamSynchronize(stream));
        if (n > out.capThin) {
            hipFree(out.thin); out.thin = nullptr; out.capThin = 0;
            RT_HIP(hipMalloc((void**)&out.thin, sizeof(ThinTri) * (size_t)n));
            out.capThin = n;
        }
        if (n) RT_HIPC(hipMemcpyAsync(out.thin, thinTmp, sizeof(ThinTri) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        out.nThin = n;
    }

    MortonPlan plan{ 0, 0 };
    {
        // x, y, z in turn as in a plain Morton code, except that an axis whose cell is already thinner than
        // RT_MORTON_THIN times the longest cell sits its turn out (always-longest-first measured better on the
        // flat terrains and worse on the compact model of C2)
        float ext[3] = { o2f(hbOut[3]) - o2f(hbOut[0]), o2f(hbOut[4]) - o2f(hbOut[1]), o2f(hbOut[5]) - o2f(hbOut[2]) };
        int used[3] = { 0, 0, 0 };
        for (int k = 0; k < 3; k++) if (!(ext[k] > 0.0f) || !(ext[k] < FLT_MAX)) ext[k] = 0.0f;
        int turn = 0;
        for (int b = 0; b < 63; b++) {
            float longest = 0.0f;
            for (int k = 0; k < 3; k++) if (used[k] < 30 && ext[k] > longest) longest = ext[k];
            int ax = -1;
            for (int t = 0; t < 3 && ax < 0; t++) {
                const int k = (turn + t) % 3;
                if (used[k] < 30 && ext[k] >= RT_MORTON_THIN * longest) ax = k;
            }
            if (ax < 0) for (int k = 0; k < 3 && ax < 0; k++) if (used[k] < 30) ax = k;
            turn = (ax + 1) % 3;
            used[ax]++; ext[ax] *= 0.5f;
            if (b < 32) plan.lo |= (unsigned long long)ax << (2 * b); else plan.hi |= (unsigned long long)ax << (2 * (b - 32));
        }
    }
    hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, nTris, dBounds, plan, keysA, valsA);
    RT_HIPC(hipGetLastError());

    {
        size_t bytes = sortBytes;
        RT_HIPC(rocprim::radix_sort_pairs(sortTmp, bytes, keysA, keysB, valsA, valsB, (size_t)nTris, 0, 63, stream));
    }
    // from here on: valsB = sorted order, keysB = sorted keys; keysA and valsA are free
    uint32_t* pos = (uint32_t*)keysA;      // final leaf slot of sorted triangle j (written by the collapse)

    hipLaunchKernelGGL(leafbox_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, valsB, nTris, maxAbs, boxes);
    RT_HIPC(hipGetLastError());

    if (nTris == 1) {
        // boxes[0..5] of "node N-1+0" = index 0
        const uint32_t zero = 0;
        RT_HIPC(hipMemcpyAsync(pos, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(256), 0, stream, out.triRef, valsB, pos, nTris, out.triLeaf);
        hipLaunchKernelGGL(emit_single_kernel, dim3(1), dim3(1), 0, stream, boxes, out.nodes);
        RT_HIPC(hipGetLastError());
        RT_HIPC(hipStreamSynchronize(stream));
        cleanup();
        return true;
    }

    const int nInt = (int)nTris - 1;
    const uint32_t gridI = ((uint32_t)nInt + 255) / 256;
    int root = 0;
    if (builder == 1) {
        // ---- radix tree (fast rebuilds)
        RT_HIPC(hipMemsetAsync(done, 0, sizeof(uint32_t) * (size_t)nInt, stream));
        hipLaunchKernelGGL(karras_kernel, dim3(gridI), dim3(256), 0, stream, keysB, (int)nTris, children, count, parent);
        RT_HIPC(hipGetLastError());
        // box fit: frontiers ping-pong in the collapse's queues (free until then); `done` counts arrivals
        int* cur = (int*)queueA;
        int* next = (int*)queueB;
        RT_HIPC(hipMemsetAsync(remaining, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(fit_leaves_kernel, dim3(grid), dim3(256), 0, stream, parent, (int)nTris, done, cur, remaining);
        RT_HIPC(hipGetLastError());
        uint32_t cnt = 0;
        for (uint32_t pass = 0;; pass++) {
            RT_HIPC(hipMemcpyAsync(&cnt, remaining, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            RT_HIPC(hipStreamSynchronize(stream));
            if (cnt == 0) break;
            if (pass > (uint32_t)nInt) { err = "BVH box fit did not terminate"; cleanup(); return false; }
            RT_HIPC(hipMemsetAsync(remaining, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(fit_frontier_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, children, parent, (int)nTris, boxes,
                               done, cur, cnt, next, remaining);
            RT_HIPC(hipGetLastError());
            int* t = cur; cur = next; next = t;
        }
    } else {
        // ---- PLOC: clusters ping-pong in the collapse's queues, neighbours in `parent`, flags and their
        // prefix sums in the two key arrays (the sort is done with them), totals in the bounds cell
        int* cur = (int*)queueA;
        int* next = (int*)queueB;
        int* nn = parent;
        unsigned long long *flags = keysA, *prefix = keysB;
        uint32_t* totals = dBounds;
        hipLaunchKernelGGL(ploc_init_kernel, dim3(grid), dim3(256), 0, stream, cur, (int)nTris);
        int m = (int)nTris, nodeBase = 0;
        for (int pass = 0; m > 1; pass++) {
            if (pass > (int)nTris) { err = "PLOC did not terminate"; cleanup(); return false; }
            const uint32_t g = ((uint32_t)m + 255) / 256;
            hipLaunchKernelGGL(ploc_nn_kernel, dim3(g), dim3(256), 0, stream, cur, m, boxes, (int)nTris, nn);
            hipLaunchKernelGGL(ploc_flag_kernel, dim3(g), dim3(256), 0, stream, nn, m, flags);
            RT_HIPC(hipGetLastError());
            size_t bytes = scanBytes;
            RT_HIPC(rocprim::exclusive_scan(sortTmp, bytes, flags, prefix, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), stream));
            hipLaunchKernelGGL(ploc_merge_kernel, dim3(g), dim3(256), 0, stream, cur, nn, m, flags, prefix, nodeBase, (int)nTris, children,
                               count, boxes, next, totals);
            RT_HIPC(hipGetLastError());
            uint32_t tot[2] = { 0, 0 };
            RT_HIPC(hipMemcpyAsync(tot, totals, sizeof(tot), hipMemcpyDeviceToHost, stream));
            RT_HIPC(hipStreamSynchronize(stream));
            if (tot[1] == 0 || (int)tot[0] >= m) { err = "PLOC pass merged nothing"; cleanup(); return false; }
            m = (int)tot[0]; nodeBase += (int)tot[1];
            int* t = cur; cur = next; next = t;
        }
        if (nodeBase != nInt) { err = "PLOC node count mismatch"; cleanup(); return false; }
        root = nInt - 1; // the last node created
    }

    // collapse: `remaining` doubles as the next level's task count, done[0] as the node allocator; the
    // leaf slot of each task's leftmost triangle travels in two arrays beside the queues
    uint32_t *firstA = valsA, *firstB = (uint32_t*)parent + nTris;
    const int2 rootTask = make_int2(root, 0);
    const uint32_t one = 1, zero = 0;
    RT_HIPC(hipMemcpyAsync(queueA, &rootTask, sizeof(rootTask), hipMemcpyHostToDevice, stream));
    RT_HIPC(hipMemcpyAsync(firstA, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
    RT_HIPC(hipMemcpyAsync(done, &one, sizeof(one), hipMemcpyHostToDevice, stream));
    uint32_t qCount = 1;
    int depthLevels = 0;
    out.levelBegin.push_back(0u); out.levelBegin.push_back(1u);   // level 0 is the root; a level's launch allocates the next level's nodes
    for (int level = 0; qCount > 0; level++, depthLevels++) {
        if (level > 4 * kStackDepth) { err = "BVH collapse did not terminate"; cleanup(); return false; }
        RT_HIPC(hipMemsetAsync(remaining, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(collapse_kernel, dim3((qCount + 255) / 256), dim3(256), 0, stream, children, count, boxes,
                           (int)nTris, queueA, firstA, qCount, queueB, firstB, remaining, done, out.nodes, pos);
        RT_HIPC(hipGetLastError());
        uint32_t allocated = 0;
        RT_HIPC(hipMemcpyAsync(&qCount, remaining, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RT_HIPC(hipMemcpyAsync(&allocated, done, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RT_HIPC(hipStreamSynchronize(stream));
        out.levelBegin.push_back(allocated);
        int2* t = queueA; queueA = queueB; queueB = t;
        uint32_t* tf = firstA; firstA = firstB; firstB = tf;
    }
    // the full-depth traversal stack (exact kernels, redo launch) holds kStackDepth entries and a visit pushes at most 3;
    // measured depths: 11 (3.7 K triangles) .. 19 (13.5 M).  A tree this deep would drop subtrees silently: refuse it.
    if (3 * depthLevels + 2 > kStackDepth) {
        err = "BVH too deep for the traversal stack (" + std::to_string(depthLevels) + " four-wide levels)";
        cleanup();
        return false;
    }
    if (getenv("RT_AMD_DEBUG_BUILD")) fprintf(stderr, "[rt_amd] builder %d: %u triangles, four-wide tree depth %d\n", builder, nTris, depthLevels);
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, valsB, pos, nTris, out.triLeaf);
    RT_HIPC(hipGetLastError());
    uint32_t nNodes = 0;
    RT_HIPC(hipMemcpyAsync(&nNodes, done, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    RT_HIPC(hipStreamSynchronize(stream));
    // the summed area of the node boxes: what later refits are measured against
    RT_HIPC(hipMemsetAsync(out.sArea, 0, sizeof(float), stream));
    hipLaunchKernelGGL(node_area_kernel, dim3(grid < 1024u ? grid : 1024u), dim3(256), 0, stream, out.nodes, nNodes, out.sArea);
    RT_HIPC(hipGetLastError());
    RT_HIPC(hipMemcpyAsync(&out.buildArea, out.sArea, sizeof(float), hipMemcpyDeviceToHost, stream));
    RT_HIPC(hipStreamSynchronize(stream));
    out.nNodes = nNodes;
    out.lastArea = out.buildArea;
    cleanup();
    return true;
#undef RT_HIPC
}

bool refit_scene(const Segment* dSegs, int nSegs, const ObjXform* dXf, uint32_t nTris, BuiltScene& out, float maxGrowth,
                 hipStream_t stream, std::string& err)
{
    err.clear();
    if (nTris < 2 || nTris != out.nTris || out.levelBegin.size() < 3 || !out.sBoxes || !(out.buildArea > 0.0f)) return false;
    const uint32_t grid = (nTris + 255) / 256;
    const uint32_t hb[6] = { host_f2o(FLT_MAX), host_f2o(FLT_MAX), host_f2o(FLT_MAX),
                             host_f2o(-FLT_MAX), host_f2o(-FLT_MAX), host_f2o(-FLT_MAX) };
    RT_HIP(hipMemcpyAsync(out.sBounds, hb, sizeof(hb), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(flatten_kernel, dim3(grid < 4096u ? grid : 4096u), dim3(256), 0, stream, dSegs, nSegs, dXf, nTris, out.triRef,
                       out.triShade, out.sBounds);
    RT_HIP(hipGetLastError());
    uint32_t hbOut[6];
    RT_HIP(hipMemcpyAsync(hbOut, out.sBounds, sizeof(hbOut), hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    float maxAbs = 0.0f;
    for (int k = 0; k < 6; k++) {
        const float f = o2f(hbOut[k]);
        if (f < FLT_MAX && f > -FLT_MAX && fabsf(f) > maxAbs) maxAbs = fabsf(f);
    }
    out.sceneMaxAbs = maxAbs;
    // the triangles no box can stand for, listed again (their lines have moved, and the threshold follows the scene's extent)
    RT_HIP(hipMemsetAsync(out.sThinCount, 0, 4, stream));
    hipLaunchKernelGGL(thin_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, nTris, maxAbs * 7.62939453125e-06f, out.sThinCount, out.sThinTmp);
    RT_HIP(hipGetLastError());
    // leaf-order copies, then the levels of the tree, deepest first
    hipLaunchKernelGGL(regather_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, nTris, out.triLeaf);
    RT_HIP(hipGetLastError());
    for (size_t L = out.levelBegin.size() - 1; L-- > 0;) {
        const uint32_t b = out.levelBegin[L], e = out.levelBegin[L + 1];
        if (e <= b) continue;
        hipLaunchKernelGGL(refit_level_kernel, dim3((e - b + 255) / 256), dim3(256), 0, stream, out.nodes, out.sBoxes,
                           out.triLeaf + 2 * (size_t)nTris, maxAbs, b, e);
    }
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemsetAsync(out.sArea, 0, sizeof(float), stream));
    hipLaunchKernelGGL(node_area_kernel, dim3(grid < 1024u ? grid : 1024u), dim3(256), 0, stream, out.nodes, out.nNodes, out.sArea);
    RT_HIP(hipGetLastError());
    uint32_t n = 0;
    float area = 0.0f;
    RT_HIP(hipMemcpyAsync(&n, out.sThinCount, 4, hipMemcpyDeviceToHost, stream));
    RT_HIP(hipMemcpyAsync(&area, out.sArea, sizeof(float), hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    if (n > out.capThin) {
        hipFree(out.thin); out.thin = nullptr; out.capThin = 0;
        RT_HIP(hipMalloc((void**)&out.thin, sizeof(ThinTri) * (size_t)n));
        out.capThin = n;
    }
    if (n) RT_HIP(hipMemcpyAsync(out.thin, out.sThinTmp, sizeof(ThinTri) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    out.nThin = n;
    out.lastArea = area;
    RT_HIP(hipStreamSynchronize(stream));
    if (getenv("RT_AMD_DEBUG_BUILD")) fprintf(stderr, "[rt_amd] refit: node area %.4g, %.3f x the built tree's\n", area, area / out.buildArea);
    // a tree whose boxes have grown past the limit (objects that have moved through each other) is rebuilt by the caller
    return area <= maxGrowth * out.buildArea;
}

void free_scene(BuiltScene& s)
{
    hipFree(s.triRef); hipFree(s.triLeaf); hipFree(s.triShade); hipFree(s.nodes); hipFree(s.arena); hipFree(s.thin);
    s = BuiltScene();
}

} // namespace rt
