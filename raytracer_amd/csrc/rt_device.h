// rt_device.h -- device data model of librt_amd (HBM layout).
//
// The reference chases four dependent pointers from a ray to its first vertex
// (GPUObjectDataVector -> GPUObjectData -> LLGPUObjectData -> GPUMeshData -> arrays,
// SURVEY.md 2.2).  Here rt_scene_commit flattens every instance into world space once, with
// the reference's own arithmetic (matVecMul, GPURaytracer.cu:728-743), so that the render
// kernel reads fixed-size records addressed by one index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr uint32_t kNoHit = 0xFFFFFFFFu;

// Lane mask of a predicate.  HIP's __ballot takes an int, which turns every vote of a bool into a select (0 / 1) and a
// compare against 0 on top of the compare that made the bool: two vector instructions per vote, six per turn of the
// traversal loops.
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long rt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX 2
#endif
constexpr int kLeafMax = RT_LEAF_MAX;        // triangles per BVH leaf
constexpr int kStackDepth = 128;   // exact kernels: up to 3 pushes per level of the four-wide tree
constexpr int kLdsStack = 32;      // entries per lane kept in LDS; deeper entries spill to scratch
#ifndef RT_WF_STACK
#define RT_WF_STACK 28
#endif
constexpr int kWfStack = RT_WF_STACK; // LDS stack entries per lane in wf_trace (overflow -> redo): 28 KB per workgroup + 9 KB of ray pool (30 entries: one workgroup per CU less, measured)
constexpr int kBlock = 256;        // workgroup size of every tracing kernel
constexpr int kCounterSlots = 256; // statistics counters are striped over this many 128-byte lines
constexpr int kCounterStride = 16; // u64 per slot

// Unsigned division by a launch constant as a multiplication (Granlund & Montgomery 1994, figure 4.1: exact for every
// 32-bit n and every d >= 1): a 32-bit division by a run-time value is some thirty vector instructions, three of them at
// quarter rate, and wf_gen did three per sample.
struct FastDiv { uint32_t m, s1, s2; };
inline FastDiv make_fast_div(uint32_t d)
{
    FastDiv f; f.m = 1u; f.s1 = 0u; f.s2 = 0u;
    if (d <= 1u) return f;                       // n / 1: t = 0, q = n
    uint32_t L = 0; while ((1ull << L) < (unsigned long long)d) L++;   // ceil(log2 d)
    f.m = (uint32_t)((((1ull << L) - d) << 32) / d + 1ull);
    f.s1 = 1u; f.s2 = L - 1u;
    return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f)
{
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}
#endif

// 48 B: one world-space triangle, three 16-byte loads.  Two arrays of these exist: `triRef`
// in reference order (object, mesh, triangle: index = gid) and `triLeaf` in BVH leaf order.
struct TriIsect {
    float v0x, v0y, v0z; uint32_t gid;   // rank in reference iteration order = tie-break key
    float v1x, v1y, v1z; uint32_t obj;   // object index (array order of rt_scene_commit)
    float v2x, v2y, v2z; uint32_t mat;   // TriShade::matFlags: the trace kernels hand the winner's material to the shade kernel
};
static_assert(sizeof(TriIsect) == 48, "TriIsect layout");

// 128 B = one cache line, reference order: everything the shading of the winner needs -- a copy of its world-space
// positions (48 B) in front of the 80 B of attributes, so that an incoherent hit costs one line instead of the
// two to four that a 48-byte and an 80-byte record at unrelated addresses straddle
struct __attribute__((aligned(128))) TriShade {
    TriIsect pos;
    float n0[3], n1[3], n2[3];   // inverse-transpose transformed vertex normals (un-normalised)
    float uv0[2], uv1[2], uv2[2];
    uint32_t obj, mesh, tri;     // hit ids: object, mesh-in-object, triangle-in-mesh
    uint32_t matFlags;           // material id | hasNormals << 31
    uint32_t pad;
};
static_assert(sizeof(TriShade) == 128, "TriShade layout");

// 96 B per object: what intersectsObjectBoundingBox (GPURaytracer.cu:267-315) reads
struct ObjCull {
    float inv[16];
    float bmin[3];
    float bmax[3];
    uint32_t triStart;  // first gid of this object
    uint32_t triCount;
};
static_assert(sizeof(ObjCull) == 96, "ObjCull layout");

struct DevTexture {
    const uchar4* data;
    int32_t w, h;
};

// 128 B: the material constants and, inline, the descriptors of its four maps -- the shading chain is then
// hit -> material -> texels instead of hit -> material -> descriptor table -> texels (each step a dependent
// fetch at ~800 cycles of L2 latency; measured, that chain, not bandwidth, was what wf_shade waited for)
struct DevMaterial {
    float albedo[4];
    float roughness, metal;
    float emission[3];
    float strength;
    int32_t texNormal, texAlbedo, texRoughness, texMetal;  // texture handle or -1 (the maps below are bound iff >= 0)
    uint32_t pad[2];
    DevTexture tex[4];                                       // normal, albedo, roughness, metal
};
static_assert(sizeof(DevMaterial) == 128, "DevMaterial layout");

// 64 B four-wide BVH node with quantised child boxes (after Ylitie, Karras & Laine 2017,
// "Efficient incoherent ray traversal on GPUs through compressed wide BVHs").  Measured on MI355X
// (tools/ubench/gather.hip): a divergent per-lane fetch of a 64-byte record costs ~3 cycles per
// lane of CU throughput when the table is L2-resident and a 128-byte record ~8; beyond L2 the cost
// is per request (~10 cycles per lane) whatever the size.  So the node count per ray has to drop
// without the node growing: four children per 64-byte fetch.
// Child boxes are stored as 8-bit offsets from the node's own box: plane = origin + q * scale,
// scale a power of two per axis, lower planes rounded down and upper planes rounded up, so a
// decoded box always contains the exact one.
// child >= 0: internal node index; child < 0: leaf, ~child = (first << 3) | count, count 0..kLeafMax
// (count 0 = unused slot, whose box is also inverted: lo = 255, hi = 0).
struct BvhNode {
    float ox, oy, oz;        // origin = lower corner of the node's box
    float sx, sy, sz;        // scale per axis (power of two)
    uint32_t lox, loy, loz;  // byte c = child c
    uint32_t hix, hiy, hiz;
    int32_t child[4];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode layout");

// A triangle of (nearly) no area, as the line it lies on: the end of its longest edge, the edge's unit direction, and the two
// constants of the proximity test (rt_trace.h, thin_near): k = 2^-18 / length, len2 = length^2.
struct ThinTri {
    float px, py, pz, k;
    float ux, uy, uz, len2;
    uint32_t gid, pad[3];
};
static_assert(sizeof(ThinTri) == 48, "ThinTri layout");

struct SceneView {
    const TriIsect* triRef;    // [nTris] reference order
    const TriIsect* triLeaf;   // [3][nTris] BVH leaf order; copy k holds the coordinates rotated for rays whose
                               // dominant direction axis is k: (x,y,z) -> (k+1, k+2, k) mod 3
    const TriShade* triShade;  // [nTris] reference order
    const ObjCull* objects;    // [nObjects]
    const DevMaterial* materials;
    const BvhNode* nodes;
    uint32_t nTris;
    uint32_t nObjects;
    int32_t rootIsValid;       // 0 when the scene holds no triangle
    float sceneMaxAbs;         // max |coordinate| over all world-space vertices
    float bandRel;             // relative width of the tie band (rt_trace.h band_of): 2e-4, or wider for ill-conditioned instances
    const ThinTri* thin;       // [nThin] the triangles no box can stand for (rt_commit.hip, thin_kernel)
    uint32_t nThin;
};

// ---- wavefront pipeline (rt_render.hip, "wf_" kernels) ----
// One slot per sample of a chunk of 8x8 pixel tiles: slot = pixelInChunk * spp + sampleIndex.
// Per bounce level L a queue lists the slots whose path is still alive; it is cut into kWfStripes
// sub-queues with their own counts and cursors (a single hot counter saturates at ~88 atomics/us).
#ifdef RT_DIAG_NO_THIN_ALL   // diagnostic build: no listed-line test anywhere (camera rays, exact walks)
#define RT_DIAG_NO_THIN_GEN 1
#endif
#ifndef RT_EXACT_PASS
#define RT_EXACT_PASS 1   // 1: the tie-band rays of a level are listed and traced again by wf_exact before the level is shaded;
#endif                    // 0: every tie-band ray sends its sample to the redo launch instead (four launches per frame fewer, whole paths redone)
#ifndef RT_EXACT_GRID
#define RT_EXACT_GRID 256u
#endif
constexpr int kWfStripes = 64;
constexpr int kWfCtrlStride = 32;          // words between two counters (one 128-byte line each)
constexpr uint32_t kWfHole = 0xFFFFFFFFu;  // queue entry without a path
constexpr uint32_t kWfMiss = 0x0FFFFFFFu;  // hitG: no triangle
constexpr uint32_t kWfDanger = 0x80000000u; // hitG flag: the sample's path must be traced again by the redo launch

struct WfBufs {
    // per level, indexed by queue position (stripe, offset), ping-pong over levels like the queue:
    // a level's shade launch reads its entries front to back and writes the survivors' next to each other
    float4* rayO;        // [2][kWfStripes][subCap] origin, maxDistance (levels >= 1; level 0 = the camera's, wf_ray_origin)
    float4* rayD;        // [2][kWfStripes][subCap] direction, RNG state
    float4* accO;        // [2][kWfStripes][subCap] outgoing light, checksum bits
    float4* accB;        // [2][kWfStripes][subCap] beta accumulation
    float4* hitA;        // [kWfStripes][subCap] t, b0, b1, b2 of the closest hit (trace -> shade of one level)
    uint2* hitM;         // [kWfStripes][subCap] gid | flags, object: with the object named here the shade kernel starts the triangle
                         // and the object-cull fetches of a hit at once instead of one after the other
    float4* out;         // [slot] the sample's outgoing light so far (final once its path has ended; wf_resolve applies the exposure), checksum bits
    uint8_t* outRays;    // [slot] closest-hit queries of the finished sample
    uint32_t* redoSlots; // [kWfStripes][subCap] the samples (slots) whose path the redo launch traces again with the exact in-order
                         // emulation inline, listed under the stripe they live in (a sample is listed at most once: its path ends there)
    uint32_t* exactList; // [kWfStripes][subCap] hit positions of the level's rays inside the tie band, listed per stripe (RT_EXACT_PASS)
    uint32_t* ctrl;      // [level][2][kWfStripes][kWfCtrlStride]: counts, cursors; then one redo count per stripe; then per level one tie-band count per stripe
    uint32_t subCap;
    uint32_t nSlots;
    uint32_t nPix;       // pixels of the chunk = tiles * 64 (edge tiles include pixels outside the rectangle)
    uint32_t tileBegin;  // the chunk's tiles are tileBegin + i * tileStride (tile index over the render rectangle,
    uint32_t tileStride; // row-major): chunks interleave, so that each gets the same mix of cheap and costly tiles
};

// words of WfBufs::ctrl for `levels` bounce levels, and where the two kinds of single counters sit in it
__host__ __device__ inline size_t wf_ctrl_words(int levels)
{
    return (size_t)(levels + 1) * 2 * kWfStripes * kWfCtrlStride + (size_t)((1 + levels) * kWfStripes) * kWfCtrlStride;
}
__host__ __device__ inline size_t wf_ctrl_exact(int levels, int level, int stripe)
{
    return (size_t)(levels + 1) * 2 * kWfStripes * kWfCtrlStride + (size_t)((1 + level) * kWfStripes + stripe) * kWfCtrlStride;
}
__host__ __device__ inline size_t wf_ctrl_redo(int levels, int stripe)
{
    return (size_t)(levels + 1) * 2 * kWfStripes * kWfCtrlStride + (size_t)stripe * kWfCtrlStride;
}
struct RenderArgs {
    SceneView scene;
    // CameraParams (GPURaytracer.hpp:60-69)
    float camOrigin[4];
    float camRotation[4];
    float exposure;
    int32_t width, height;
    const float4* rays;        // W*H pre-rotation directions
    // RendererParams (:78-87)
    int32_t bounceCount;
    float maxDistance;
    float aoIntensity;
    uint32_t frameCount;
    uint32_t progressiveFrameCount;
    int32_t antiAliasing;
    uint32_t sampleCount;
    // SkyLightParams (:71-76)
    float skyDir[4], lightColor[4], skyColor[4];
    // tile + output
    int32_t x0, y0, x1, y1;
    float4* fb;
    uint4* ids;                // may be null
    unsigned long long* counters; // [0] rays, [1] slow-path rays, [2] node visits, [3] tri tests, [4] shaded hits (8 slots)
    int32_t tilesX, tilesY;
    // persistent path kernel: 8x8-pixel tiles handed out through 8 striped counters
    uint32_t* tileCounters;       // 8 counters, 32 words apart
    int32_t tiles8X, tiles8Y;
    // pixels whose result depends on the exact in-order emulation are re-rendered by a second launch
    uint32_t* redoList;           // pixel indices y*width + x
    uint32_t* redoCount;
    const uint32_t* pixelList;    // non-null: render these pixels instead of the tile
    const uint32_t* pixelCount;
    // rt_render_bands (wavefront pipeline): the rectangle is cut into bands of bandTileRows rows of 8x8 tiles and only
    // bands bandFirst, bandFirst + bandStride, ... are rendered; the pipeline's tile rows count through those bands
    // only (tiles8Y = owned bands * bandTileRows).  bandTileRows == 0: the whole rectangle.
    int32_t bandTileRows, bandFirst, bandStride;
    // divisions of the wavefront pipeline by its three launch constants (launch_render_wf fills them in; fast_div)
    FastDiv divSpp, divTilesX, divBandRows;
};

} // namespace rt
