// rt_commit.hip -- scene upload for gfx950: flatten every instance into world-space records
// and build the acceleration structure on the GPU.
//
// Replaces the per-frame instance table of Scene::getGPUObjectDataVector (Scene.cpp:3-29) and the
// pointer tables of DisplayObject::sendToGPU (DisplayObject.cpp:30-81).  The reference builds no
// acceleration structure; this file adds one (LBVH: Morton codes -> radix sort -> Karras radix
// tree -> level-synchronous box fit -> 64-byte two-child nodes with up to kLeafMax triangles per
// leaf).  World-space positions and normals are produced with the reference's own expressions
// (matVecMul, GPURaytracer.cu:728-743; normals through the transposed inverse, :463-469) so the
// render kernel tests exactly the triangles the reference would.
#include "rt_device.h"
#include "rt_commit.h"
#include <cstring>
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
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float mnx = FLT_MAX, mny = FLT_MAX, mnz = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX, mxz = -FLT_MAX;
    if (g < nTris) {
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
        ti.v2x = v[2][0]; ti.v2y = v[2][1]; ti.v2z = v[2][2]; ti.pad = 0;
        triRef[g] = ti;

        TriShade ts;
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

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ void tri_box(const TriIsect& t, float* mn, float* mx)
{
    mn[0] = __builtin_fminf(t.v0x, __builtin_fminf(t.v1x, t.v2x)); mx[0] = __builtin_fmaxf(t.v0x, __builtin_fmaxf(t.v1x, t.v2x));
    mn[1] = __builtin_fminf(t.v0y, __builtin_fminf(t.v1y, t.v2y)); mx[1] = __builtin_fmaxf(t.v0y, __builtin_fmaxf(t.v1y, t.v2y));
    mn[2] = __builtin_fminf(t.v0z, __builtin_fminf(t.v1z, t.v2z)); mx[2] = __builtin_fmaxf(t.v0z, __builtin_fmaxf(t.v1z, t.v2z));
}

__global__ __launch_bounds__(256) void morton_kernel(const TriIsect* __restrict__ triRef, uint32_t nTris,
                                                     const uint32_t* __restrict__ bounds, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ vals)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nTris) return;
    const float bx = o2f(bounds[0]), by = o2f(bounds[1]), bz = o2f(bounds[2]);
    const float ex = o2f(bounds[3]) - bx, ey = o2f(bounds[4]) - by, ez = o2f(bounds[5]) - bz;
    float mn[3], mx[3];
    tri_box(triRef[g], mn, mx);
    const float cx = 0.5f * (mn[0] + mx[0]), cy = 0.5f * (mn[1] + mx[1]), cz = 0.5f * (mn[2] + mx[2]);
    float fx = ex > 0.0f ? (cx - bx) / ex : 0.0f, fy = ey > 0.0f ? (cy - by) / ey : 0.0f, fz = ez > 0.0f ? (cz - bz) / ez : 0.0f;
    fx = __builtin_fminf(__builtin_fmaxf(fx * 1024.0f, 0.0f), 1023.0f); // NaN -> 0
    fy = __builtin_fminf(__builtin_fmaxf(fy * 1024.0f, 0.0f), 1023.0f);
    fz = __builtin_fminf(__builtin_fmaxf(fz * 1024.0f, 0.0f), 1023.0f);
    keys[g] = (expand_bits10((uint32_t)fx) << 2) | (expand_bits10((uint32_t)fy) << 1) | expand_bits10((uint32_t)fz);
    vals[g] = g;
}

// leaf-order copy of the intersection records + leaf boxes (node index N-1+j)
__global__ __launch_bounds__(256) void gather_kernel(const TriIsect* __restrict__ triRef, const uint32_t* __restrict__ order,
                                                     uint32_t nTris, TriIsect* __restrict__ triLeaf,
                                                     float* __restrict__ boxes)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nTris) return;
    const TriIsect t = triRef[order[j]];
    triLeaf[j] = t;
    float mn[3], mx[3];
    tri_box(t, mn, mx);
    float* b = boxes + 6 * (size_t)(nTris - 1 + j);
    b[0] = mn[0]; b[1] = mn[1]; b[2] = mn[2]; b[3] = mx[0]; b[4] = mx[1]; b[5] = mx[2];
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees":
// internal node i of the binary radix tree over the sorted keys (ties broken by position).
__device__ __forceinline__ int delta(const uint32_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((uint32_t)(i ^ j));
    return __clz(a ^ b);
}

__global__ __launch_bounds__(256) void karras_kernel(const uint32_t* __restrict__ keys, int n, int2* __restrict__ children,
                                                     int2* __restrict__ ranges)
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
    ranges[i] = make_int2(first, last);
}

// Level-synchronous box fit.  Pass p finishes every internal node whose children were finished
// in an earlier pass (done[c] in [1, p)); kernel boundaries provide the visibility, so no
// in-kernel inter-workgroup hand-off is needed (MI355X_MICROARCH.md: per-XCD L2s are not coherent).
__global__ __launch_bounds__(256) void fit_kernel(const int2* __restrict__ children, int nInternal, int nLeaves,
                                                  float* __restrict__ boxes, uint32_t* __restrict__ done, uint32_t pass,
                                                  uint32_t* __restrict__ remaining)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nInternal) return;
    if (done[i] != 0) return;
    const int2 c = children[i];
    const bool lOk = c.x < 0 || (done[c.x] != 0 && done[c.x] < pass);
    const bool rOk = c.y < 0 || (done[c.y] != 0 && done[c.y] < pass);
    if (!(lOk && rOk)) { atomicAdd(remaining, 1u); return; }
    const float* a = boxes + 6 * (size_t)(c.x < 0 ? nLeaves - 1 + (~c.x) : c.x);
    const float* b = boxes + 6 * (size_t)(c.y < 0 ? nLeaves - 1 + (~c.y) : c.y);
    float* o = boxes + 6 * (size_t)i;
    o[0] = __builtin_fminf(a[0], b[0]); o[1] = __builtin_fminf(a[1], b[1]); o[2] = __builtin_fminf(a[2], b[2]);
    o[3] = __builtin_fmaxf(a[3], b[3]); o[4] = __builtin_fmaxf(a[4], b[4]); o[5] = __builtin_fmaxf(a[5], b[5]);
    done[i] = pass;
}

__device__ __forceinline__ int32_t encode_child(int c, const int2* __restrict__ ranges)
{
    if (c < 0) return ~(int32_t)(((uint32_t)(~c) << 3) | 0u);
    const int2 r = ranges[c];
    const int count = r.y - r.x + 1;
    if (count <= kLeafMax) return ~(int32_t)(((uint32_t)r.x << 3) | (uint32_t)(count - 1));
    return c;
}

__global__ __launch_bounds__(256) void emit_kernel(const int2* __restrict__ children, const int2* __restrict__ ranges,
                                                   const float* __restrict__ boxes, int nInternal, int nLeaves,
                                                   BvhNode* __restrict__ nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nInternal) return;
    const int2 c = children[i];
    const float* a = boxes + 6 * (size_t)(c.x < 0 ? nLeaves - 1 + (~c.x) : c.x);
    const float* b = boxes + 6 * (size_t)(c.y < 0 ? nLeaves - 1 + (~c.y) : c.y);
    BvhNode nd;
    nd.lmin[0] = a[0]; nd.lmin[1] = a[1]; nd.lmin[2] = a[2]; nd.lmax[0] = a[3]; nd.lmax[1] = a[4]; nd.lmax[2] = a[5];
    nd.rmin[0] = b[0]; nd.rmin[1] = b[1]; nd.rmin[2] = b[2]; nd.rmax[0] = b[3]; nd.rmax[1] = b[4]; nd.rmax[2] = b[5];
    nd.left = encode_child(c.x, ranges);
    nd.right = encode_child(c.y, ranges);
    nd.pad[0] = nd.pad[1] = 0;
    nodes[i] = nd;
}

// single-triangle scene: one node, the second child is an empty box
__global__ void emit_single_kernel(const float* __restrict__ boxes, BvhNode* __restrict__ nodes)
{
    BvhNode nd;
    for (int k = 0; k < 3; k++) {
        nd.lmin[k] = boxes[k]; nd.lmax[k] = boxes[3 + k];
        nd.rmin[k] = FLT_MAX; nd.rmax[k] = -FLT_MAX;
    }
    nd.left = ~(int32_t)0; nd.right = ~(int32_t)0;
    nd.pad[0] = nd.pad[1] = 0;
    nodes[0] = nd;
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

bool build_scene(const Segment* dSegs, int nSegs, const ObjXform* dXf, uint32_t nTris, BuiltScene& out,
                 hipStream_t stream, std::string& err)
{
    out.nTris = nTris;
    out.nNodes = 0;
    out.sceneMaxAbs = 0.0f;
    if (nTris == 0) return true;
    const uint32_t grid = (nTris + 255) / 256;

    RT_HIP(hipMalloc((void**)&out.triRef, sizeof(TriIsect) * (size_t)nTris));
    RT_HIP(hipMalloc((void**)&out.triLeaf, sizeof(TriIsect) * (size_t)nTris));
    RT_HIP(hipMalloc((void**)&out.triShade, sizeof(TriShade) * (size_t)nTris));
    const uint32_t nInternal = nTris > 1 ? nTris - 1 : 1;
    RT_HIP(hipMalloc((void**)&out.nodes, sizeof(BvhNode) * (size_t)nInternal));
    out.nNodes = nInternal;

    // scratch
    uint32_t *dBounds = nullptr, *keysA = nullptr, *keysB = nullptr, *valsA = nullptr, *valsB = nullptr, *done = nullptr,
             *remaining = nullptr;
    float* boxes = nullptr;
    int2 *children = nullptr, *ranges = nullptr;
    void* sortTmp = nullptr;
    auto cleanup = [&]() {
        hipFree(dBounds); hipFree(keysA); hipFree(keysB); hipFree(valsA); hipFree(valsB); hipFree(done);
        hipFree(remaining); hipFree(boxes); hipFree(children); hipFree(ranges); hipFree(sortTmp);
    };
#define RT_HIPC(call)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return false; } \
    } while (0)

    RT_HIPC(hipMalloc((void**)&dBounds, 6 * sizeof(uint32_t)));
    const uint32_t hb[6] = { host_f2o(FLT_MAX), host_f2o(FLT_MAX), host_f2o(FLT_MAX),
                             host_f2o(-FLT_MAX), host_f2o(-FLT_MAX), host_f2o(-FLT_MAX) };
    RT_HIPC(hipMemcpyAsync(dBounds, hb, sizeof(hb), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(flatten_kernel, dim3(grid), dim3(256), 0, stream, dSegs, nSegs, dXf, nTris, out.triRef, out.triShade,
                       dBounds);
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

    RT_HIPC(hipMalloc((void**)&keysA, sizeof(uint32_t) * (size_t)nTris));
    RT_HIPC(hipMalloc((void**)&keysB, sizeof(uint32_t) * (size_t)nTris));
    RT_HIPC(hipMalloc((void**)&valsA, sizeof(uint32_t) * (size_t)nTris));
    RT_HIPC(hipMalloc((void**)&valsB, sizeof(uint32_t) * (size_t)nTris));
    hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, nTris, dBounds, keysA, valsA);
    RT_HIPC(hipGetLastError());

    size_t tmpBytes = 0;
    RT_HIPC(rocprim::radix_sort_pairs(nullptr, tmpBytes, keysA, keysB, valsA, valsB, (size_t)nTris, 0, 30, stream));
    RT_HIPC(hipMalloc(&sortTmp, tmpBytes ? tmpBytes : 16));
    RT_HIPC(rocprim::radix_sort_pairs(sortTmp, tmpBytes, keysA, keysB, valsA, valsB, (size_t)nTris, 0, 30, stream));

    RT_HIPC(hipMalloc((void**)&boxes, sizeof(float) * 6 * (size_t)(2 * (size_t)nTris)));
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, stream, out.triRef, valsB, nTris, out.triLeaf, boxes);
    RT_HIPC(hipGetLastError());

    if (nTris == 1) {
        // boxes[0..5] of "node N-1+0" = index 0
        hipLaunchKernelGGL(emit_single_kernel, dim3(1), dim3(1), 0, stream, boxes, out.nodes);
        RT_HIPC(hipGetLastError());
        RT_HIPC(hipStreamSynchronize(stream));
        cleanup();
        return true;
    }

    const int nInt = (int)nTris - 1;
    const uint32_t gridI = ((uint32_t)nInt + 255) / 256;
    RT_HIPC(hipMalloc((void**)&children, sizeof(int2) * (size_t)nInt));
    RT_HIPC(hipMalloc((void**)&ranges, sizeof(int2) * (size_t)nInt));
    RT_HIPC(hipMalloc((void**)&done, sizeof(uint32_t) * (size_t)nInt));
    RT_HIPC(hipMalloc((void**)&remaining, sizeof(uint32_t)));
    RT_HIPC(hipMemsetAsync(done, 0, sizeof(uint32_t) * (size_t)nInt, stream));
    hipLaunchKernelGGL(karras_kernel, dim3(gridI), dim3(256), 0, stream, keysB, (int)nTris, children, ranges);
    RT_HIPC(hipGetLastError());

    bool fitted = false;
    for (uint32_t pass = 1; pass <= 128; pass++) {
        RT_HIPC(hipMemsetAsync(remaining, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(fit_kernel, dim3(gridI), dim3(256), 0, stream, children, nInt, (int)nTris, boxes, done, pass,
                           remaining);
        RT_HIPC(hipGetLastError());
        if (pass >= 8 || nInt < 256) { // trees are at least log2(n) deep; poll after a few passes
            uint32_t rem = 0;
            RT_HIPC(hipMemcpyAsync(&rem, remaining, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            RT_HIPC(hipStreamSynchronize(stream));
            if (rem == 0) { fitted = true; break; }
        }
    }
    if (!fitted) { err = "BVH box fit did not converge in 128 passes"; cleanup(); return false; }

    hipLaunchKernelGGL(emit_kernel, dim3(gridI), dim3(256), 0, stream, children, ranges, boxes, nInt, (int)nTris, out.nodes);
    RT_HIPC(hipGetLastError());
    RT_HIPC(hipStreamSynchronize(stream));
    cleanup();
    return true;
#undef RT_HIPC
}

void free_scene(BuiltScene& s)
{
    hipFree(s.triRef); hipFree(s.triLeaf); hipFree(s.triShade); hipFree(s.nodes);
    s = BuiltScene();
}

} // namespace rt
