// rt_api.hip -- implementation of the C ABI declared in include/rt_api.h.
// Host-side bookkeeping only; the device work is in rt_commit.hip and rt_render.hip.
#include "../../include/rt_api.h"
#include "rt_commit.h"
#include "rt_detmath.h"
#include "../host/rt_jpeg.hpp"
#include "../host/rt_png.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr size_t kWorkBytes = sizeof(uint32_t) * 512;
constexpr size_t kWfChunkDefault = 192u << 20;   // samples; C3 (531 M): 8 chunks of 64 Mi 66.7 ms, 4 of 128 Mi 64.1, 3 of 177 M 62.2
constexpr size_t kWfChunkFloor = 64u << 20;      // what the default falls back to, by halves, when the workspace does not fit the device
constexpr int kWfLanesDefault = 3;        // most chunks in flight: the process has 4 hardware queues and one carries the copies, a 4th lane shares a queue and serialises
constexpr size_t kCntBytes = sizeof(unsigned long long) * rt::kCounterSlots * rt::kCounterStride;
// A refitted tree whose node boxes sum to more than this times the built tree's area is rebuilt.  Measured on C4 (331 bins
// jittered independently, tools/refit_probe.py): area x 1.29 -> frame +15 %, x 1.54 -> +18 %, x 1.74 -> +34 %, x 1.93 -> +53 %,
// against 3.9 ms that a refit (1.6 ms) saves over a build (5.5 ms): past x 1.5 the next frame loses more than the commit won.
// A refitted tree whose node boxes sum to more than this (percent) times the built tree's area is rebuilt.  Measured on C4
// (tools/refit_probe.py): the frame on a refitted tree costs +0.3 ms at 1.10 x the area, +1.1 at 1.17, +1.4 at 1.25, +4.0 at
// 1.5, against the 3.6 ms a refit saves over a build -- and on the next commit the rebuilt tree refits from 1.0 again.
constexpr int kRefitGrowthDefault = 130;

thread_local std::string g_createError;

// ---- static workspace of rt_render_noalloc (callers that cannot allocate: HIP graph capture, serving loops) ----
// One copy per GPU, held until the process ends.  `scratch` is only a host-side stand-in for the symbol: its
// device address on each GPU is taken once with hipGetSymbolAddress (rt_ctx_create, that device current) and every
// use below goes through that pointer.
constexpr size_t kScratchBytes = 64ull << 20;   // 64 MiB
constexpr int kScratchMaxLevels = 16;           // bounceCount limit of rt_render_noalloc
constexpr int kScratchMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
std::mutex g_scratchMutex;
void* g_scratchPtr[kScratchMaxDevices] = {};    // [device] address of that GPU's copy of `scratch`

struct MeshRec {
    bool live = false;
    float* pos = nullptr; int32_t* idx = nullptr; float* uv = nullptr; float* nrm = nullptr;
    int32_t nVerts = 0, nTris = 0;
};
// red: the first channel alone, one byte per texel, behind the RGBA texels in the same allocation (what a roughness or
// metal binding samples, rt_render.hip tex2d_norm_red)
struct TexRec { bool live = false; uchar4* data = nullptr; uint8_t* red = nullptr; int32_t w = 0, h = 0; };
struct MatRec { bool live = false; rt_material_desc desc; };
struct ObjRec {
    bool live = false;
    std::vector<rt_id> meshes; std::vector<int32_t> matIdx; std::vector<rt_id> mats;
    float bmin[3], bmax[3];
};
struct FbRec {
    bool live = false, owned = false, idsEnabled = false;
    float4* data = nullptr; uint4* ids = nullptr; int32_t w = 0, h = 0;
};

} // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    std::vector<MeshRec> meshes;
    std::vector<TexRec> textures;
    std::vector<MatRec> materials;
    std::vector<ObjRec> objects;
    std::vector<FbRec> fbs;

    rt::BuiltScene built;
    bool committed = false;
    // refit bookkeeping: the segment list, mesh epoch and builder of the last FULL build (rt_scene_commit)
    std::vector<rt::Segment> builtSegs; uint64_t builtEpoch = 0, meshEpoch = 0; int builtBuilder = 0; bool builtValid = false;
    int commitKind = 0;                       // last rt_scene_commit: 1 = full build, 2 = refit
    int optRefit = kRefitGrowthDefault;       // 0 = never refit; else the growth limit in percent
    int optThinList = 1;                      // 0: the listed-line tests of the collinear triangles are off (diagnostic)
    rt::Segment* dSegs = nullptr; rt::ObjXform* dXf = nullptr;
    size_t capObjectsBytes = 0, capMaterialsBytes = 0, capSegsBytes = 0, capXfBytes = 0;
    rt::ObjCull* dObjects = nullptr; uint32_t nObjects = 0;
    rt::DevMaterial* dMaterials = nullptr; size_t capMaterials = 0;
    std::vector<rt::DevMaterial> hMaterials;   // host copy of dMaterials: which textures the committed scene can reach

    float4* dRays = nullptr; int32_t raysW = 0, raysH = 0;
    unsigned long long* dCounters = nullptr;   // 8 x u64
    unsigned long long* hCounters = nullptr;   // pinned
    hipEvent_t evBegin = nullptr, evEnd = nullptr;
    hipEvent_t slots[16] = {};
    bool renderedOnce = false;
    float commitMs = 0.0f;
    int optTraversal = 0;
    int optWfLanes = 0;                       // 0 = by the number of chunks
    int optWfLane = 0;                        // first workspace lane of the next render calls (RT_OPT_WF_LANE)
    int optBuilder = 0;                       // 0 = PLOC, 1 = radix tree
    size_t optWfChunk = kWfChunkDefault;     // samples in flight per chunk of the wavefront pipeline
    void* wfArena = nullptr; size_t wfArenaBytes = 0;   // one allocation, carved per lane by wf_carve
    rt::WfLanes wf; size_t wfCapSlots = 0; int wfLevels = 0; int wfLanes = 0; // its lanes and their workspace
    int optCounters = 0;
    int optAccumulate = 0;
    int numCUs = 256;
    uint32_t* dWork = nullptr;      // 8 tile counters (32 words apart) + redo count at word 256
    uint32_t* dRedoList = nullptr; size_t redoCap = 0;
    float bandRel = 2e-4f;          // tie band of the committed scene, see band_rel_of
    uchar4* dPresent = nullptr; size_t presentCap = 0; float* dSrgb = nullptr;   // rt_present: converted frame, threshold table
    void* scratchPtr = nullptr;     // this device's copy of the static workspace (rt_render_noalloc)
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg)
{
    if (c) c->error = msg; else g_createError = msg;
    return code;
}

#define RT_CHECK(c, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), RT_ERR_HIP, std::string(#call) + " failed: " + hipGetErrorString(e_)); \
    } while (0)

#define RT_BIND(c)                                                                              \
    do {                                                                                        \
        if (!(c)) return fail(nullptr, RT_ERR_INVALID, "null context");                         \
        RT_CHECK((c), hipSetDevice((c)->device));                                               \
    } while (0)

template <class T> rt_id alloc_slot(std::vector<T>& v)
{
    for (size_t i = 0; i < v.size(); i++)
        if (!v[i].live) return (rt_id)i;
    v.emplace_back();
    return (rt_id)(v.size() - 1);
}
template <class T> bool valid(const std::vector<T>& v, rt_id id) { return id >= 0 && (size_t)id < v.size() && v[id].live; }

rt::DevMaterial to_dev(const rt_material_desc& d, const std::vector<TexRec>& textures)
{
    rt::DevMaterial m;
    memset(&m, 0, sizeof(m));
    const rt_id t[4] = { d.texNormal, d.texAlbedo, d.texRoughness, d.texMetal };
    for (int i = 0; i < 4; i++)
        if (t[i] >= 0 && (size_t)t[i] < textures.size() && textures[t[i]].live) {
            m.tex[i].data = i >= 2 ? reinterpret_cast<const uchar4*>(textures[t[i]].red) : textures[t[i]].data;
            m.tex[i].w = textures[t[i]].w; m.tex[i].h = textures[t[i]].h;
        }
    memcpy(m.albedo, d.albedo, sizeof(m.albedo));
    m.roughness = d.roughness; m.metal = d.metal;
    memcpy(m.emission, d.emissionColor, sizeof(m.emission));
    m.strength = d.emissionStrength;
    m.texNormal = d.texNormal; m.texAlbedo = d.texAlbedo; m.texRoughness = d.texRoughness; m.texMetal = d.texMetal;
    return m;
}

int check_material_desc(rt_ctx* c, const rt_material_desc* d)
{
    if (!d) return fail(c, RT_ERR_INVALID, "null material desc");
    const rt_id t[4] = { d->texNormal, d->texAlbedo, d->texRoughness, d->texMetal };
    for (int i = 0; i < 4; i++)
        if (t[i] != -1 && !valid(c->textures, t[i])) return fail(c, RT_ERR_INVALID, "material references an invalid texture handle");
    return RT_OK;
}

// Relative width of the tie band for a set of instances (exactness).  The BVH answer equals
// the reference's in-order answer unless another object has a hit within rounding distance of the winner; that
// distance is set by how far the reference's object-space cull (origin and direction through the host's float
// inverse, GPURaytracer.cu:267-315) can disagree with the world-space triangle test: up to
// cond(model) * 2^-23 * (extent + |translation|), and 1 / cos(incidence) times that for a grazing ray.  The band is
// rel * (t + max |coordinate|); rel = 2e-4 covers cond * 1 / cos up to ~1700 (cond = 1 for any rigid or uniformly
// scaled instance); beyond that it grows with the worst instance: 64 * cond * 2^-23.  A singular or non-finite
// matrix makes it 1e30 (finite, so that rel * 0 stays 0): every ray that hits two objects then takes the exact
// in-order emulation; a ray that hits one object only does not (trav_result: otherT stays FLT_MAX).
float band_rel_of(const float* model16, const float* inv16, int n)
{
    double worst = 1.0;
    for (int i = 0; i < n; i++) {
        const float* m = model16 + 16 * (size_t)i;
        const float* v = inv16 + 16 * (size_t)i;
        double nm = 0.0, nv = 0.0;
        for (int r = 0; r < 3; r++) { // infinity norms of the 3x3 parts (column-major storage)
            nm = std::max(nm, (double)std::fabs(m[r]) + std::fabs(m[4 + r]) + std::fabs(m[8 + r]));
            nv = std::max(nv, (double)std::fabs(v[r]) + std::fabs(v[4 + r]) + std::fabs(v[8 + r]));
        }
        const double cond = nm * nv;
        if (!(cond < 1e30)) return 1e30f;   // also NaN
        worst = std::max(worst, cond);
    }
    const double rel = 64.0 * worst * 1.1920929e-07;
    return (float)std::min(1e30, std::max(2e-4, rel));
}

// Lays one lane's wavefront workspace out in [base, base + return value): ~177 bytes per sample in flight
// (path state of both ping-pong halves 128, hit 24, result 17, pixel flag + redo list 8).
// base == nullptr: sizing only.  Every array starts on a 256-byte boundary.
size_t wf_carve(uint8_t* base, size_t slots, int levels, rt::WfBufs& w)
{
    w = rt::WfBufs{};
    const size_t pieces = (slots + 255) / 256;
    w.subCap = (uint32_t)(((pieces + rt::kWfStripes - 1) / rt::kWfStripes) * 256);
    const size_t queueWords = 2 * (size_t)rt::kWfStripes * w.subCap; // positions of both ping-pong halves
    size_t off = 0;
    auto take = [&](size_t bytes) -> uint8_t* {
        uint8_t* p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    w.rayO = (float4*)take(16 * queueWords);
    w.rayD = (float4*)take(16 * queueWords);
    w.accO = (float4*)take(16 * queueWords);
    w.accB = (float4*)take(16 * queueWords);
    w.hitA = (float4*)take(16 * (queueWords / 2));
    w.hitM = (uint2*)take(8 * (queueWords / 2));
    w.out = (float4*)take(16 * slots);
    w.outRays = (uint8_t*)take(slots);
    w.redoSlots = (uint32_t*)take(4 * (queueWords / 2));
    w.exactList = (uint32_t*)take(RT_EXACT_PASS ? 4 * (queueWords / 2) : 0);
    w.ctrl = (uint32_t*)take(4 * rt::wf_ctrl_words(levels));
    return off;
}

void free_wf(rt_ctx* c)
{
    hipFree(c->wfArena); c->wfArena = nullptr; c->wfArenaBytes = 0;
    for (int j = 0; j < rt::kWfMaxLanes; j++) c->wf.bufs[j] = rt::WfBufs{};
    c->wfCapSlots = 0; c->wfLevels = 0; c->wfLanes = 0;
}

// workspace of the wavefront pipeline: one arena, `lanes` lanes of `slots` samples each.
// polite: the caller can do with less (the default chunk size, which halves itself): the arena then takes at most three
// quarters of what the device has free -- a second context, or torch's caching allocator beside this one, keeps room --
// and the arena in use is only given up once the new size has passed that test.
hipError_t ensure_wf(rt_ctx* c, size_t slots, int levels, int lanes, bool polite)
{
    if (slots <= c->wfCapSlots && levels <= c->wfLevels && lanes <= c->wfLanes) return hipSuccess;
    if (slots < c->wfCapSlots) slots = c->wfCapSlots;
    if (levels < c->wfLevels) levels = c->wfLevels;
    if (lanes < c->wfLanes) lanes = c->wfLanes;
    rt::WfBufs w;
    const size_t perLane = wf_carve(nullptr, slots, levels, w);
    if (polite) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && perLane * (size_t)lanes > (freeB + c->wfArenaBytes) / 4 * 3) return hipErrorOutOfMemory;
    }
    hipError_t e = hipDeviceSynchronize();   // every stream: calls on other lanes (RT_OPT_WF_LANE) may be in flight in the arena that is about to go
    if (e != hipSuccess) return e;
    free_wf(c);
    e = hipMalloc((void**)&c->wfArena, perLane * (size_t)lanes);
    if (e != hipSuccess) { c->wfArena = nullptr; return e; }
    c->wfArenaBytes = perLane * (size_t)lanes;
    for (int j = 0; j < lanes; j++) wf_carve((uint8_t*)c->wfArena + perLane * (size_t)j, slots, levels, c->wf.bufs[j]);
    c->wfCapSlots = slots; c->wfLevels = levels; c->wfLanes = lanes;
    return hipSuccess;
}

rt::SceneView scene_view(const rt_ctx* c)
{
    rt::SceneView s;
    s.triRef = c->built.triRef; s.triLeaf = c->built.triLeaf; s.triShade = c->built.triShade;
    s.objects = c->dObjects; s.materials = c->dMaterials; s.nodes = c->built.nodes;
    s.nTris = c->built.nTris; s.nObjects = c->nObjects;
    s.rootIsValid = c->built.nTris > 0 ? 1 : 0;
    s.sceneMaxAbs = c->built.sceneMaxAbs;
    s.bandRel = c->bandRel;
    s.thin = c->built.thin; s.nThin = c->optThinList ? c->built.nThin : 0u;
    return s;
}

__global__ void detmath_kernel(int op, const float* __restrict__ in, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float r;
    switch (op) {
    case 0: r = rtd::log_f(x); break;
    case 1: r = rtd::cos_f(x); break;
    case 2: r = rtd::exp_f(x); break;
    case 3: r = rtd::pow_f(x, 5.0f); break;
    case 4: r = rtd::pow_f(x, 1000.0f); break;
    case 5: r = rtd::rsqrt_f(x); break;
    case 7: r = rtd::byte_unorm(__float_as_uint(x)); break;
    case 8: { // exhaustive check of u32_unorm: the input is the top 16 bits of v; the result is the number of mismatches
        const uint32_t hi = __float_as_uint(x) << 16;
        uint32_t bad = 0;
        for (uint32_t lo = 0; lo < 65536u; lo++) {
            const uint32_t v = hi | lo;
            const float want = (float)((double)v / 4294967295.0);
            bad += __float_as_uint(rtd::u32_unorm(v)) != __float_as_uint(want) ? 1u : 0u;
        }
        r = (float)bad;
        break;
    }
    case 9: { // pow_f(x, 5) against pow5_f, and pow1000_f against pow_f(x, 1000): input = top 16 bits of x; result = mismatches
        const uint32_t hi = __float_as_uint(x) << 16;
        uint32_t bad = 0;
        for (uint32_t lo = 0; lo < 65536u; lo++) {
            const float v = __uint_as_float(hi | lo);
            if (!(v >= 0.0f && v <= 1.0f)) continue;
            bad += __float_as_uint(rtd::pow_f(v, 5.0f)) != __float_as_uint(rtd::pow5_f(v)) ? 1u : 0u;
            bad += __float_as_uint(rtd::pow_f(v, 1000.0f)) != __float_as_uint(rtd::pow1000_f(v)) ? 65536u : 0u;
        }
        r = (float)bad;
        break;
    }
    default: {
        uint32_t seed = __float_as_uint(x);
        seed = seed * 747796405u + 2891336453u;
        uint32_t v = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
        v = (v >> 22) ^ v;
        r = (float)((double)v / 4294967295.0);
    }
    }
    out[i] = r;
}

// float4 linear -> RGBA8 sRGB (rt_present): comparisons against the threshold table only, so that the bytes are the host's
__global__ void present_srgb8_kernel(const float4* __restrict__ fb, int w, int h, int flip, const float* __restrict__ thr,
                                     uchar4* __restrict__ out)
{
    __shared__ float t[256];
    t[threadIdx.x] = thr[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)w * (size_t)h) return;
    const int y = (int)(i / (size_t)w), x = (int)(i % (size_t)w);
    const float4 p = fb[(size_t)(flip ? h - 1 - y : y) * (size_t)w + (size_t)x];
    auto enc = [&](float v) -> unsigned char { // number of thresholds <= v; NaN counts as 0
        int lo = 0, hi = 255;                  // invariant: t[lo - 1] <= v < t[hi] (t[255] = +inf)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (v >= t[mid]) lo = mid + 1; else hi = mid; }
        return (unsigned char)lo;
    };
    const float a = p.w != p.w ? 0.0f : (p.w < 0.0f ? 0.0f : (p.w > 1.0f ? 1.0f : p.w));
    out[i] = make_uchar4(enc(p.x), enc(p.y), enc(p.z), (unsigned char)(a * 255.0f + 0.5f));
}

} // namespace

namespace rt {
void launch_present_srgb8(const float4* fb, int w, int h, bool flip, const float* thr, uchar4* out, hipStream_t stream)
{
    const size_t n = (size_t)w * (size_t)h;
    hipLaunchKernelGGL(present_srgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fb, w, h, flip ? 1 : 0, thr, out);
}
void launch_detmath(int op, const float* in, int n, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, in, n, out);
}
} // namespace rt

extern "C" {

uint8_t* rt_image_load(const char* path, int32_t* width, int32_t* height, int32_t* channels)
{
    if (!path || !width || !height || !channels) return nullptr;
    std::vector<unsigned char> pix; int w = 0, h = 0, n = 0;
    if (!rth::png::decode(path, pix, w, h, n) && !rth::jpeg::decode(path, pix, w, h, n)) return nullptr;
    uint8_t* out = (uint8_t*)malloc(pix.size() ? pix.size() : 1);
    if (!out) return nullptr;
    memcpy(out, pix.data(), pix.size());
    *width = w; *height = h; *channels = n;
    return out;
}

void rt_image_free(uint8_t* pixels) { free(pixels); }

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_version(void) { return "raytracer_amd 0.3.0 (gfx950) build " RT_BUILD_ID; }   // the id: csrc/Makefile

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_createError.c_str(); }

int rt_ctx_create(int device, rt_ctx** out)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(nullptr, RT_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    RT_CHECK(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only");
    RT_CHECK(nullptr, hipSetDevice(device));
    void* scratchPtr = nullptr;
    if (device < kScratchMaxDevices) { // every GPU holds its own copy of the array at its own address
        std::lock_guard<std::mutex> lock(g_scratchMutex);
        if (!g_scratchPtr[device]) {
            void* p = nullptr;
            RT_CHECK(nullptr, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratchPtr[device] = p;
        }
        scratchPtr = g_scratchPtr[device];
    }
    rt_ctx* c = new rt_ctx();
    c->device = device;
    c->scratchPtr = scratchPtr;
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->dCounters, kCntBytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->hCounters, kCntBytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->dWork, kWorkBytes);
    for (int j = 1; j < rt::kWfMaxLanes && e == hipSuccess; j++) {
        e = hipStreamCreateWithFlags(&c->wf.streams[j], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->wf.join[j], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->wf.fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c->evBegin);
    if (e == hipSuccess) e = hipEventCreate(&c->evEnd);
    for (int i = 0; i < 16 && e == hipSuccess; i++) e = hipEventCreate(&c->slots[i]);
    if (e != hipSuccess) {
        std::string m = std::string("context set-up failed: ") + hipGetErrorString(e);
        rt_ctx_destroy(c);
        return fail(nullptr, RT_ERR_HIP, m);
    }
    memset(c->hCounters, 0, kCntBytes);
    c->stream = c->ownStream;
    *out = c;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* c)
{
    if (!c) return RT_OK;
    hipSetDevice(c->device);
    if (c->ownStream) hipStreamSynchronize(c->ownStream);
    hipDeviceSynchronize();
    for (auto& m : c->meshes) if (m.live) { hipFree(m.pos); hipFree(m.idx); hipFree(m.uv); hipFree(m.nrm); }
    for (auto& t : c->textures) if (t.live) hipFree(t.data);
    for (auto& f : c->fbs) if (f.live) { if (f.owned) hipFree(f.data); hipFree(f.ids); }
    rt::free_scene(c->built);
    hipFree(c->dSegs); hipFree(c->dXf);
    hipFree(c->dObjects); hipFree(c->dMaterials); hipFree(c->dRays); hipFree(c->dPresent); hipFree(c->dSrgb); hipFree(c->dCounters); hipFree(c->dWork); hipFree(c->dRedoList);
    free_wf(c);
    if (c->hCounters) hipHostFree(c->hCounters);
    if (c->evBegin) hipEventDestroy(c->evBegin);
    if (c->evEnd) hipEventDestroy(c->evEnd);
    for (int i = 0; i < 16; i++) if (c->slots[i]) hipEventDestroy(c->slots[i]);
    for (int j = 1; j < rt::kWfMaxLanes; j++) {
        if (c->wf.streams[j]) { hipStreamSynchronize(c->wf.streams[j]); hipStreamDestroy(c->wf.streams[j]); }
        if (c->wf.join[j]) hipEventDestroy(c->wf.join[j]);
    }
    if (c->wf.fork) hipEventDestroy(c->wf.fork);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
    return RT_OK;
}

int rt_set_stream(rt_ctx* c, void* s)
{
    RT_BIND(c);
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return RT_OK;
}

int rt_set_option(rt_ctx* c, int option, int value)
{
    RT_BIND(c);
    switch (option) {
    case RT_OPT_TRAVERSAL:
        if (value < 0 || value > 2)
            return fail(c, RT_ERR_INVALID, "RT_OPT_TRAVERSAL takes 0 (BVH, wavefront pipeline), 1 (brute force) or 2 (BVH, single persistent kernel)");
        c->optTraversal = value; return RT_OK;
    case RT_OPT_WF_LANES:
        if (value < 0 || value > rt::kWfMaxLanes) return fail(c, RT_ERR_INVALID, "RT_OPT_WF_LANES takes 0 (default) .. 4");
        c->optWfLanes = value; return RT_OK;
    case RT_OPT_WF_LANE:
        if (value < 0 || value >= rt::kWfMaxLanes) return fail(c, RT_ERR_INVALID, "RT_OPT_WF_LANE takes 0 .. 3");
        c->optWfLane = value; return RT_OK;
    case RT_OPT_REFIT:
        if (value < 0 || (value > 1 && value <= 100)) return fail(c, RT_ERR_INVALID, "RT_OPT_REFIT takes 0 (off), 1 (default limit) or a growth limit in percent above 100");
        c->optRefit = value == 1 ? kRefitGrowthDefault : value; return RT_OK;
    case RT_OPT_THIN_LIST:
        if (value < 0 || value > 1) return fail(c, RT_ERR_INVALID, "RT_OPT_THIN_LIST takes 0 or 1");
        c->optThinList = value; return RT_OK;
    case RT_OPT_BUILDER:
        if (value < 0 || value > 1) return fail(c, RT_ERR_INVALID, "RT_OPT_BUILDER takes 0 (PLOC, default) or 1 (radix tree, fast rebuilds)");
        c->optBuilder = value; return RT_OK;
    case RT_OPT_WF_CHUNK:
        if (value < 0) return fail(c, RT_ERR_INVALID, "RT_OPT_WF_CHUNK takes a sample count >= 0 (0 = default)");
        c->optWfChunk = value ? (size_t)value : kWfChunkDefault; return RT_OK;
    case RT_OPT_COUNTERS: c->optCounters = value ? 1 : 0; return RT_OK;
    case RT_OPT_ACCUMULATE_STATS:
        c->optAccumulate = value ? 1 : 0;
        RT_CHECK(c, hipMemsetAsync(c->dCounters, 0, kCntBytes, c->stream));
        return RT_OK;
    default: return fail(c, RT_ERR_INVALID, "unknown option");
    }
}

int rt_sync(rt_ctx* c)
{
    RT_BIND(c);
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_get_stats(rt_ctx* c, rt_stats* out)
{
    RT_BIND(c);
    if (!out) return fail(c, RT_ERR_INVALID, "null stats pointer");
    memset(out, 0, sizeof(*out));
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->renderedOnce) {
        RT_CHECK(c, hipMemcpy(c->hCounters, c->dCounters, kCntBytes, hipMemcpyDeviceToHost));
        unsigned long long sum[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int sl = 0; sl < rt::kCounterSlots; sl++)
            for (int k = 0; k < 8; k++) sum[k] += c->hCounters[(size_t)sl * rt::kCounterStride + k];
        out->rays = sum[0]; out->slow_path = sum[1];
        out->node_visits = sum[2]; out->tri_tests = sum[3]; out->hits = sum[4];
        out->packet_node_tests = sum[5]; out->packet_node_fetches = sum[6];
        float ms = 0.0f;
        RT_CHECK(c, hipEventElapsedTime(&ms, c->evBegin, c->evEnd));
        out->kernel_ms = ms;
    }
    out->triangles = c->built.nTris; out->bvh_nodes = c->built.nNodes; out->commit_ms = c->commitMs;
    return RT_OK;
}

/* ---- scene upload ---- */

int rt_mesh_create(rt_ctx* c, const float* pos, int32_t nVerts, const int32_t* idx, int32_t nTris, const float* uv,
                   const float* nrm, rt_id* out)
{
    RT_BIND(c);
    if (!out) return fail(c, RT_ERR_INVALID, "null out pointer");
    if (nVerts < 0 || nTris < 0 || (nVerts > 0 && !pos) || (nTris > 0 && !idx))
        return fail(c, RT_ERR_INVALID, "mesh arrays missing");
    for (int64_t i = 0; i < (int64_t)nTris * 3; i++)
        if (idx[i] < 0 || idx[i] >= nVerts) return fail(c, RT_ERR_INVALID, "mesh index out of range");
    MeshRec m;
    m.nVerts = nVerts; m.nTris = nTris;
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        if (!h || bytes == 0) return hipSuccess;
        hipError_t e = hipMalloc(d, bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    hipError_t e = up((void**)&m.pos, pos, sizeof(float) * 3 * (size_t)nVerts);
    if (e == hipSuccess) e = up((void**)&m.idx, idx, sizeof(int32_t) * 3 * (size_t)nTris);
    if (e == hipSuccess) e = up((void**)&m.uv, uv, sizeof(float) * 2 * (size_t)nVerts);
    if (e == hipSuccess) e = up((void**)&m.nrm, nrm, sizeof(float) * 3 * (size_t)nVerts);
    if (e != hipSuccess) {
        hipFree(m.pos); hipFree(m.idx); hipFree(m.uv); hipFree(m.nrm);
        return fail(c, RT_ERR_HIP, std::string("mesh upload failed: ") + hipGetErrorString(e));
    }
    m.live = true;
    c->meshEpoch++;   // a new mesh may sit where a destroyed one sat: the next commit builds, it does not refit
    const rt_id id = alloc_slot(c->meshes);
    c->meshes[id] = m;
    *out = id;
    return RT_OK;
}

int rt_mesh_destroy(rt_ctx* c, rt_id id)
{
    RT_BIND(c);
    if (!valid(c->meshes, id)) return fail(c, RT_ERR_INVALID, "invalid mesh handle");
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    MeshRec& m = c->meshes[id];
    hipFree(m.pos); hipFree(m.idx); hipFree(m.uv); hipFree(m.nrm);
    m = MeshRec();
    c->meshEpoch++;
    return RT_OK;
}

int rt_texture_create(rt_ctx* c, const uint8_t* rgba8, int32_t w, int32_t h, rt_id* out)
{
    RT_BIND(c);
    if (!out || !rgba8 || w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID, "texture needs rgba8 data and positive extents");
    TexRec t; t.w = w; t.h = h;
    const size_t n = (size_t)w * (size_t)h;
    RT_CHECK(c, hipMalloc((void**)&t.data, n * 5));
    t.red = reinterpret_cast<uint8_t*>(t.data) + n * 4;
    std::vector<uint8_t> red(n);
    for (size_t i = 0; i < n; i++) red[i] = rgba8[i * 4];
    hipError_t e = hipMemcpy(t.data, rgba8, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t.red, red.data(), n, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(t.data); return fail(c, RT_ERR_HIP, std::string("texture upload failed: ") + hipGetErrorString(e)); }
    t.live = true;
    const rt_id id = alloc_slot(c->textures);
    c->textures[id] = t;
    *out = id;
    return RT_OK;
}

int rt_texture_destroy(rt_ctx* c, rt_id id)
{
    RT_BIND(c);
    if (!valid(c->textures, id)) return fail(c, RT_ERR_INVALID, "invalid texture handle");
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    // the committed tables hold the texel pointer: a committed material that still binds this texture
    // would sample freed memory, so the scene then has to be committed again before the next rt_render
    if (c->committed) {
        for (const rt::DevMaterial& m : c->hMaterials)
            if (m.texNormal == id || m.texAlbedo == id || m.texRoughness == id || m.texMetal == id) c->committed = false;
    }
    hipFree(c->textures[id].data);
    c->textures[id] = TexRec();
    return RT_OK;
}

int rt_material_create(rt_ctx* c, const rt_material_desc* d, rt_id* out)
{
    RT_BIND(c);
    if (!out) return fail(c, RT_ERR_INVALID, "null out pointer");
    int r = check_material_desc(c, d);
    if (r) return r;
    const rt_id id = alloc_slot(c->materials);
    c->materials[id].live = true;
    c->materials[id].desc = *d;
    *out = id;
    return RT_OK;
}

int rt_material_update(rt_ctx* c, rt_id id, const rt_material_desc* d)
{
    RT_BIND(c);
    if (!valid(c->materials, id)) return fail(c, RT_ERR_INVALID, "invalid material handle");
    int r = check_material_desc(c, d);
    if (r) return r;
    c->materials[id].desc = *d;
    // hot update of a committed scene (the reference re-sends the GPUMaterial, Material.hpp:155-159)
    if (c->committed && (size_t)id < c->capMaterials) {
        // the record carries the descriptors of its maps, so a texture created after the commit binds like any other
        const rt::DevMaterial m = to_dev(*d, c->textures);
        RT_CHECK(c, hipStreamSynchronize(c->stream));
        RT_CHECK(c, hipMemcpy(c->dMaterials + id, &m, sizeof(m), hipMemcpyHostToDevice));
        c->hMaterials[id] = m;
    }
    return RT_OK;
}

int rt_material_destroy(rt_ctx* c, rt_id id)
{
    RT_BIND(c);
    if (!valid(c->materials, id)) return fail(c, RT_ERR_INVALID, "invalid material handle");
    c->materials[id] = MatRec();
    return RT_OK;
}

int rt_object_create(rt_ctx* c, const rt_id* meshes, const int32_t* matIdx, int32_t nMeshes, const rt_id* mats,
                     int32_t nMats, const float bmin[3], const float bmax[3], rt_id* out)
{
    RT_BIND(c);
    if (!out || nMeshes < 0 || nMats < 0 || (nMeshes > 0 && (!meshes || !matIdx)) || (nMats > 0 && !mats) || !bmin || !bmax)
        return fail(c, RT_ERR_INVALID, "object arrays missing");
    ObjRec o;
    for (int i = 0; i < nMats; i++) {
        if (!valid(c->materials, mats[i])) return fail(c, RT_ERR_INVALID, "object references an invalid material handle");
        o.mats.push_back(mats[i]);
    }
    for (int i = 0; i < nMeshes; i++) {
        if (!valid(c->meshes, meshes[i])) return fail(c, RT_ERR_INVALID, "object references an invalid mesh handle");
        if (matIdx[i] < 0 || matIdx[i] >= nMats) return fail(c, RT_ERR_INVALID, "mesh material index outside the object's material list");
        o.meshes.push_back(meshes[i]);
        o.matIdx.push_back(matIdx[i]);
    }
    memcpy(o.bmin, bmin, sizeof(o.bmin));
    memcpy(o.bmax, bmax, sizeof(o.bmax));
    o.live = true;
    const rt_id id = alloc_slot(c->objects);
    c->objects[id] = o;
    *out = id;
    return RT_OK;
}

int rt_object_destroy(rt_ctx* c, rt_id id)
{
    RT_BIND(c);
    if (!valid(c->objects, id)) return fail(c, RT_ERR_INVALID, "invalid object handle");
    c->objects[id] = ObjRec();
    return RT_OK;
}

int rt_scene_commit(rt_ctx* c, const rt_id* objects, const float* model16, const float* invModel16, int32_t n)
{
    RT_BIND(c);
    if (n < 0 || (n > 0 && (!objects || !model16 || !invModel16))) return fail(c, RT_ERR_INVALID, "commit arrays missing");
    RT_CHECK(c, hipStreamSynchronize(c->stream));

    std::vector<rt::Segment> segs;
    std::vector<rt::ObjXform> xf((size_t)n);
    std::vector<rt::ObjCull> cull((size_t)n);
    uint64_t total = 0;
    for (int i = 0; i < n; i++) {
        if (!valid(c->objects, objects[i])) return fail(c, RT_ERR_INVALID, "commit references an invalid object handle");
        const ObjRec& o = c->objects[objects[i]];
        memcpy(xf[i].model, model16 + 16 * (size_t)i, 64);
        memcpy(xf[i].inv, invModel16 + 16 * (size_t)i, 64);
        memcpy(cull[i].inv, invModel16 + 16 * (size_t)i, 64);
        memcpy(cull[i].bmin, o.bmin, 12);
        memcpy(cull[i].bmax, o.bmax, 12);
        cull[i].triStart = (uint32_t)total;
        for (size_t k = 0; k < o.meshes.size(); k++) {
            if (!valid(c->meshes, o.meshes[k])) return fail(c, RT_ERR_INVALID, "object references a destroyed mesh");
            const rt_id mat = o.mats[o.matIdx[k]];
            if (!valid(c->materials, mat)) return fail(c, RT_ERR_INVALID, "object references a destroyed material");
            const MeshRec& m = c->meshes[o.meshes[k]];
            if (m.nTris == 0) continue;
            rt::Segment s;
            s.gidStart = (uint32_t)total; s.triCount = (uint32_t)m.nTris;
            s.obj = (uint32_t)i; s.meshInObj = (uint32_t)k; s.materialId = (uint32_t)mat; s.pad = 0;
            s.pos = m.pos; s.idx = m.idx; s.uv = m.uv; s.nrm = m.nrm;
            segs.push_back(s);
            total += (uint64_t)m.nTris;
        }
        cull[i].triCount = (uint32_t)(total - cull[i].triStart);
    }
    if (total >= (1ull << 28)) return fail(c, RT_ERR_INVALID, "scene exceeds 2^28 triangles");

    // tables: kept across commits while they fit (a scene that is re-committed every frame allocates nothing)
    c->committed = false;
    auto fit = [&](void** p, size_t& cap, size_t bytes) -> hipError_t {
        if (bytes <= cap) return hipSuccess;
        hipFree(*p); *p = nullptr; cap = 0;
        const hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    };
    c->nObjects = 0;
    if (n > 0) {
        RT_CHECK(c, fit((void**)&c->dObjects, c->capObjectsBytes, sizeof(rt::ObjCull) * (size_t)n));
        RT_CHECK(c, hipMemcpy(c->dObjects, cull.data(), sizeof(rt::ObjCull) * (size_t)n, hipMemcpyHostToDevice));
        c->nObjects = (uint32_t)n;
    }
    {
        std::vector<rt::DevMaterial> mats(c->materials.size() ? c->materials.size() : 1);
        memset(mats.data(), 0, sizeof(rt::DevMaterial) * mats.size());
        for (size_t i = 0; i < c->materials.size(); i++)
            if (c->materials[i].live) mats[i] = to_dev(c->materials[i].desc, c->textures);
            else mats[i].texNormal = mats[i].texAlbedo = mats[i].texRoughness = mats[i].texMetal = -1; // unused slot: binds nothing
        if (c->materials.empty()) mats[0].texNormal = mats[0].texAlbedo = mats[0].texRoughness = mats[0].texMetal = -1;
        c->capMaterials = 0;
        RT_CHECK(c, fit((void**)&c->dMaterials, c->capMaterialsBytes, sizeof(rt::DevMaterial) * mats.size()));
        RT_CHECK(c, hipMemcpy(c->dMaterials, mats.data(), sizeof(rt::DevMaterial) * mats.size(), hipMemcpyHostToDevice));
        c->capMaterials = mats.size();
        c->hMaterials = mats;
    }
    if (!segs.empty()) {
        RT_CHECK(c, fit((void**)&c->dSegs, c->capSegsBytes, sizeof(rt::Segment) * segs.size()));
        RT_CHECK(c, fit((void**)&c->dXf, c->capXfBytes, sizeof(rt::ObjXform) * (size_t)n));
        RT_CHECK(c, hipMemcpy(c->dSegs, segs.data(), sizeof(rt::Segment) * segs.size(), hipMemcpyHostToDevice));
        RT_CHECK(c, hipMemcpy(c->dXf, xf.data(), sizeof(rt::ObjXform) * (size_t)n, hipMemcpyHostToDevice));
    }
    hipEventRecord(c->slots[14], c->stream);
    std::string err;
    // The dynamic-scene path: the same instances of the same meshes under other matrices (what the reference's per-frame
    // instance table amounts to, Scene.cpp:3-29) keep the tree of the last full build and refit its boxes; anything else,
    // or a refitted tree whose boxes have grown past the limit (RT_OPT_REFIT), is built anew.
    const bool sameStructure = c->optRefit && c->builtValid && c->builtEpoch == c->meshEpoch && c->builtBuilder == c->optBuilder &&
                               segs.size() == c->builtSegs.size() &&
                               (segs.empty() || memcmp(segs.data(), c->builtSegs.data(), sizeof(rt::Segment) * segs.size()) == 0);
    bool ok = false;
    c->commitKind = 0;
    if (sameStructure && total > 1) {
        ok = rt::refit_scene(c->dSegs, (int)segs.size(), c->dXf, (uint32_t)total, c->built, 0.01f * (float)c->optRefit, c->stream, err);
        if (ok) c->commitKind = 2;
        else if (!err.empty()) { c->builtValid = false; rt::free_scene(c->built); return fail(c, RT_ERR_HIP, "scene refit failed: " + err); }
    }
    if (!ok) {
        ok = rt::build_scene(c->dSegs, (int)segs.size(), c->dXf, (uint32_t)total, c->built, c->optBuilder, c->stream, err);
        if (ok) { c->commitKind = 1; c->builtSegs = segs; c->builtEpoch = c->meshEpoch; c->builtBuilder = c->optBuilder; c->builtValid = true; }
    }
    hipEventRecord(c->slots[15], c->stream);
    hipStreamSynchronize(c->stream);
    if (!ok) { c->builtValid = false; rt::free_scene(c->built); return fail(c, RT_ERR_HIP, "scene build failed: " + err); }
    hipEventElapsedTime(&c->commitMs, c->slots[14], c->slots[15]);
    c->bandRel = band_rel_of(model16, invModel16, n);
    c->committed = true;
    return RT_OK;
}

int rt_scene_commit_kind(const rt_ctx* c) { return c ? c->commitKind : 0; }

int rt_camera_set_rays(rt_ctx* c, const float* rays4, int32_t w, int32_t h)
{
    RT_BIND(c);
    if (!rays4 || w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID, "ray table needs data and positive extents");
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->raysW != w || c->raysH != h || !c->dRays) {
        hipFree(c->dRays); c->dRays = nullptr;
        RT_CHECK(c, hipMalloc((void**)&c->dRays, sizeof(float4) * (size_t)w * (size_t)h));
        c->raysW = w; c->raysH = h;
    }
    RT_CHECK(c, hipMemcpy(c->dRays, rays4, sizeof(float4) * (size_t)w * (size_t)h, hipMemcpyHostToDevice));
    return RT_OK;
}

/* ---- output ---- */

int rt_framebuffer_create(rt_ctx* c, int32_t w, int32_t h, rt_id* out)
{
    RT_BIND(c);
    if (!out || w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID, "framebuffer needs positive extents");
    FbRec f; f.w = w; f.h = h; f.owned = true;
    RT_CHECK(c, hipMalloc((void**)&f.data, sizeof(float4) * (size_t)w * (size_t)h));
    // on the context's stream: hipMemset runs on the null stream without waiting for the host, and the context's stream
    // (non-blocking) does not order itself against that one -- an rt_framebuffer_write or rt_render right after this call
    // raced with the clear (2 of 28 000 random frames of tests/soak_parity.py had a zeroed 4 KB page of the previous frame)
    hipError_t e = hipMemsetAsync(f.data, 0, sizeof(float4) * (size_t)w * (size_t)h, c->stream);
    if (e != hipSuccess) { hipFree(f.data); return fail(c, RT_ERR_HIP, "framebuffer clear failed"); }
    f.live = true;
    const rt_id id = alloc_slot(c->fbs);
    c->fbs[id] = f;
    *out = id;
    return RT_OK;
}

int rt_framebuffer_wrap(rt_ctx* c, void* dptr, int32_t w, int32_t h, rt_id* out)
{
    RT_BIND(c);
    if (!out || !dptr || w <= 0 || h <= 0) return fail(c, RT_ERR_INVALID, "framebuffer wrap needs a device pointer and positive extents");
    if (((uintptr_t)dptr & 15u) != 0) return fail(c, RT_ERR_INVALID, "framebuffer pointer must be 16-byte aligned");
    FbRec f; f.w = w; f.h = h; f.owned = false; f.data = (float4*)dptr; f.live = true;
    const rt_id id = alloc_slot(c->fbs);
    c->fbs[id] = f;
    *out = id;
    return RT_OK;
}

int rt_framebuffer_destroy(rt_ctx* c, rt_id id)
{
    RT_BIND(c);
    if (!valid(c->fbs, id)) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle");
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    FbRec& f = c->fbs[id];
    if (f.owned) hipFree(f.data);
    hipFree(f.ids);
    f = FbRec();
    return RT_OK;
}

int rt_framebuffer_read(rt_ctx* c, rt_id id, float* host)
{
    RT_BIND(c);
    if (!valid(c->fbs, id) || !host) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle or null host pointer");
    const FbRec& f = c->fbs[id];
    RT_CHECK(c, hipMemcpyAsync(host, f.data, sizeof(float4) * (size_t)f.w * (size_t)f.h, hipMemcpyDeviceToHost, c->stream));
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_framebuffer_write(rt_ctx* c, rt_id id, const float* host)
{
    RT_BIND(c);
    if (!valid(c->fbs, id) || !host) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle or null host pointer");
    const FbRec& f = c->fbs[id];
    RT_CHECK(c, hipMemcpyAsync(f.data, host, sizeof(float4) * (size_t)f.w * (size_t)f.h, hipMemcpyHostToDevice, c->stream));
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

const float* rt_srgb8_thresholds(void)
{
    static float table[255];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int k = 1; k <= 255; k++) { // inverse sRGB transfer function at (k - 0.5) / 255, IEC 61966-2-1
            const double e = ((double)k - 0.5) / 255.0;
            const double lin = e <= 0.04045 ? e / 12.92 : std::pow((e + 0.055) / 1.055, 2.4);
            table[k - 1] = (float)lin;
        }
    });
    return table;
}

int rt_present(rt_ctx* c, rt_id id, int format, int flipRows, void* host)
{
    RT_BIND(c);
    if (!valid(c->fbs, id) || !host) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle or null host pointer");
    const FbRec& f = c->fbs[id];
    const size_t n = (size_t)f.w * (size_t)f.h;
    if (format == RT_PRESENT_RGBA32F) {
        if (!flipRows) {
            RT_CHECK(c, hipMemcpyAsync(host, f.data, sizeof(float4) * n, hipMemcpyDeviceToHost, c->stream));
        } else { // row by row, last source row first
            for (int y = 0; y < f.h; y++)
                RT_CHECK(c, hipMemcpyAsync((char*)host + sizeof(float4) * (size_t)f.w * (size_t)y, f.data + (size_t)(f.h - 1 - y) * f.w,
                                           sizeof(float4) * (size_t)f.w, hipMemcpyDeviceToHost, c->stream));
        }
        RT_CHECK(c, hipStreamSynchronize(c->stream));
        return RT_OK;
    }
    if (format != RT_PRESENT_SRGB8) return fail(c, RT_ERR_INVALID, "rt_present: unknown format");
    if (n > c->presentCap) { // staging for the converted frame; grows with the largest frame presented
        RT_CHECK(c, hipStreamSynchronize(c->stream));
        hipFree(c->dPresent); c->dPresent = nullptr; c->presentCap = 0;
        RT_CHECK(c, hipMalloc((void**)&c->dPresent, 4 * n));
        c->presentCap = n;
    }
    if (!c->dSrgb) {
        RT_CHECK(c, hipMalloc((void**)&c->dSrgb, sizeof(float) * 256));
        float t[256];
        memcpy(t, rt_srgb8_thresholds(), sizeof(float) * 255);
        t[255] = INFINITY;
        RT_CHECK(c, hipMemcpy(c->dSrgb, t, sizeof(t), hipMemcpyHostToDevice));
    }
    rt::launch_present_srgb8(f.data, f.w, f.h, flipRows != 0, c->dSrgb, c->dPresent, c->stream);
    RT_CHECK(c, hipGetLastError());
    RT_CHECK(c, hipMemcpyAsync(host, c->dPresent, 4 * n, hipMemcpyDeviceToHost, c->stream));
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_framebuffer_enable_ids(rt_ctx* c, rt_id id, int enable)
{
    RT_BIND(c);
    if (!valid(c->fbs, id)) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle");
    FbRec& f = c->fbs[id];
    if (enable && !f.ids) {
        RT_CHECK(c, hipMalloc((void**)&f.ids, sizeof(uint4) * (size_t)f.w * (size_t)f.h));
        RT_CHECK(c, hipMemsetAsync(f.ids, 0xFF, sizeof(uint4) * (size_t)f.w * (size_t)f.h, c->stream));   // stream order, as above
    }
    f.idsEnabled = enable != 0;
    return RT_OK;
}

int rt_framebuffer_read_ids(rt_ctx* c, rt_id id, uint32_t* host)
{
    RT_BIND(c);
    if (!valid(c->fbs, id) || !host) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle or null host pointer");
    const FbRec& f = c->fbs[id];
    if (!f.ids) return fail(c, RT_ERR_STATE, "hit-id AOV was never enabled on this framebuffer");
    RT_CHECK(c, hipMemcpyAsync(host, f.ids, sizeof(uint4) * (size_t)f.w * (size_t)f.h, hipMemcpyDeviceToHost, c->stream));
    RT_CHECK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// argument checks and parameter packing shared by rt_render and rt_render_noalloc
static int render_args(rt_ctx* c, const rt_camera_params* cam, const rt_renderer_params* rnd, const rt_sky_params* sky,
                       rt_rect tile, rt_id fb, const char* who, rt::RenderArgs& a)
{
    if (!cam || !rnd || !sky) return fail(c, RT_ERR_INVALID, "null parameter block");
    if (!c->committed) return fail(c, RT_ERR_STATE, std::string(who) + " before rt_scene_commit");
    if (!valid(c->fbs, fb)) return fail(c, RT_ERR_INVALID, "invalid framebuffer handle");
    const FbRec& f = c->fbs[fb];
    if (cam->width != f.w || cam->height != f.h) return fail(c, RT_ERR_INVALID, "camera extents differ from the framebuffer's");
    if (!c->dRays || c->raysW != cam->width || c->raysH != cam->height)
        return fail(c, RT_ERR_STATE, "camera ray table missing or of different extents (rt_camera_set_rays)");
    if (tile.x0 < 0 || tile.y0 < 0 || tile.x1 > f.w || tile.y1 > f.h || tile.x0 > tile.x1 || tile.y0 > tile.y1)
        return fail(c, RT_ERR_INVALID, "tile outside the framebuffer");
    if (rnd->sampleCount == 0) return fail(c, RT_ERR_INVALID, "sampleCount must be at least 1");
    if (cam->origin[3] != 1.0f) return fail(c, RT_ERR_INVALID, "camera origin[3] must be 1 (a point, GPURaytracer.cu:29)");

    memset(&a, 0, sizeof(a));
    a.scene = scene_view(c);
    memcpy(a.camOrigin, cam->origin, 16); memcpy(a.camRotation, cam->rotation, 16);
    a.exposure = cam->exposure; a.width = cam->width; a.height = cam->height; a.rays = c->dRays;
    a.bounceCount = rnd->bounceCount; a.maxDistance = rnd->maxDistance; a.aoIntensity = rnd->aoIntensity;
    a.frameCount = rnd->frameCount; a.progressiveFrameCount = rnd->progressiveFrameCount;
    a.antiAliasing = rnd->antiAliasingEnabled; a.sampleCount = rnd->sampleCount;
    memcpy(a.skyDir, sky->direction, 16); memcpy(a.lightColor, sky->lightColor, 16); memcpy(a.skyColor, sky->skyColor, 16);
    a.x0 = tile.x0; a.y0 = tile.y0; a.x1 = tile.x1; a.y1 = tile.y1;
    a.fb = f.data; a.ids = f.idsEnabled ? f.ids : nullptr; a.counters = c->dCounters;
    a.tileCounters = c->dWork; a.redoCount = c->dWork + 256; a.redoList = nullptr;
    a.pixelList = nullptr; a.pixelCount = nullptr;
    return RT_OK;
}

// rt_render and rt_render_bands: bandRows == 0 renders the whole tile
static int render_impl(rt_ctx* c, const rt_camera_params* cam, const rt_renderer_params* rnd, const rt_sky_params* sky,
                       rt_rect tile, rt_id fb, int32_t bandRows, int32_t bandFirst, int32_t bandStride)
{
    RT_BIND(c);
    rt::RenderArgs a;
    const int rc = render_args(c, cam, rnd, sky, tile, fb, bandRows ? "rt_render_bands" : "rt_render", a);
    if (rc) return rc;
    if (bandRows) {
        if (bandRows < 0 || bandRows % 8 != 0 || bandFirst < 0 || bandStride < 1)
            return fail(c, RT_ERR_INVALID, "rt_render_bands: bandRows must be a positive multiple of 8, firstBand >= 0, bandStride >= 1");
        a.bandTileRows = bandRows / 8; a.bandFirst = bandFirst; a.bandStride = bandStride;
    }
    const bool wavefront = c->optTraversal == 0 && rnd->bounceCount <= 200;
    if (c->optTraversal == 2 || (c->optTraversal == 0 && !wavefront)) {
        // persistent path kernel: one redo list over the tile's pixels (the wavefront pipeline keeps its own per lane)
        const size_t tilePixels = (size_t)((tile.x1 - tile.x0 + 7) / 8) * (size_t)((tile.y1 - tile.y0 + 7) / 8) * 64;
        if (tilePixels > c->redoCap) { // grows only when a larger tile than ever before is rendered
            RT_CHECK(c, hipStreamSynchronize(c->stream));
            hipFree(c->dRedoList); c->dRedoList = nullptr; c->redoCap = 0;
            RT_CHECK(c, hipMalloc((void**)&c->dRedoList, sizeof(uint32_t) * tilePixels));
            c->redoCap = tilePixels;
        }
    }
    a.redoList = c->dRedoList;
    RT_CHECK(c, hipMemsetAsync(c->dWork, 0, kWorkBytes, c->stream));

    if (!c->optAccumulate) RT_CHECK(c, hipMemsetAsync(c->dCounters, 0, kCntBytes, c->stream));
    RT_CHECK(c, hipEventRecord(c->evBegin, c->stream));
    // mode 0: wavefront pipeline (the single persistent kernel, mode 2, remains for very deep paths:
    // the per-sample query count of the pipeline is a byte)
    if (wavefront) {
        const size_t tiles = (size_t)((tile.x1 - tile.x0 + 7) / 8) * (size_t)rt::wf_tile_rows(a);
        const size_t perTile = 64 * (size_t)rnd->sampleCount;
        if (perTile >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "sampleCount too large for one 8x8 tile of the wavefront workspace");
        if (tiles > 0) {
            // chunks: whole tiles, at most optWfChunk samples each.  lanes: a frame that fits one chunk runs in one
            // lane (C2 7.6 / C4 15.9 / C5 16.2 ms against 8.0 / 16.7 / 17.0 ms in three: with the axis-parallel
            // rays fixed the trace launches have no long tails left to hide); a frame of several chunks runs up
            // to three at a time, each one's tail under the others' bulk (C3, 8 chunks: 10.8 / 11.3 / 11.5 Grays/s
            // with 1 / 2 / 3 lanes).  An explicit RT_OPT_WF_LANES is taken as given, up to one lane per tile.
            size_t chunk = c->optWfChunk;
            int lanes = 1;
            size_t chunkTiles = 1;
            hipError_t e = hipSuccess;
            for (;;) {
                const size_t wantChunks = (tiles * perTile + chunk - 1) / chunk;
                lanes = c->optWfLanes ? (int)(tiles < (size_t)c->optWfLanes ? tiles : (size_t)c->optWfLanes)
                                      : (int)(wantChunks < (size_t)kWfLanesDefault ? wantChunks : (size_t)kWfLanesDefault);
                lanes = lanes < 1 ? 1 : lanes;
                if (lanes > rt::kWfMaxLanes - c->optWfLane) lanes = rt::kWfMaxLanes - c->optWfLane;   // RT_OPT_WF_LANE: the lanes past the call's first
                chunkTiles = (tiles + (size_t)lanes - 1) / (size_t)lanes;
                const size_t capTiles = chunk / perTile ? chunk / perTile : 1;
                if (chunkTiles > capTiles) chunkTiles = capTiles;
                const bool canHalve = c->optWfChunk == kWfChunkDefault && chunk / 2 >= kWfChunkFloor;
                e = ensure_wf(c, chunkTiles * perTile, rnd->bounceCount > 0 ? rnd->bounceCount : 0, c->optWfLane + lanes, canHalve);
                // the default chunk is sized for a 288 GB device (up to 107 GB of workspace for a frame of three chunks); where
                // that does not fit, smaller chunks do the same work a few percent slower.  An explicit RT_OPT_WF_CHUNK is taken as given.
                if (e != hipErrorOutOfMemory || !canHalve) break;
                (void)hipGetLastError();
                chunk /= 2;
            }
            if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("wavefront workspace: ") + hipGetErrorString(e));
            c->wf.n = lanes; c->wf.first = c->optWfLane; c->wf.tilesPerChunk = (uint32_t)chunkTiles;
            RT_CHECK(c, hipEventRecord(c->evBegin, c->stream));
            rt::launch_render_wf(a, c->wf, c->numCUs, c->stream);
        }
    } else if (!bandRows) {
        rt::launch_render(a, c->optTraversal == 1 ? 1 : 0, c->numCUs, c->stream);
    } else { // the single-kernel modes take one launch per band; their work counters start at zero each time
        rt::RenderArgs b = a;
        b.bandTileRows = 0;
        for (int32_t y = tile.y0 + bandFirst * bandRows; y < tile.y1; y += bandStride * bandRows) {
            b.y0 = y; b.y1 = y + bandRows < tile.y1 ? y + bandRows : tile.y1;
            if (y != tile.y0 + bandFirst * bandRows) RT_CHECK(c, hipMemsetAsync(c->dWork, 0, kWorkBytes, c->stream));
            rt::launch_render(b, c->optTraversal == 1 ? 1 : 0, c->numCUs, c->stream);
        }
    }
    RT_CHECK(c, hipGetLastError());
    RT_CHECK(c, hipEventRecord(c->evEnd, c->stream));
    c->renderedOnce = true;
    return RT_OK;
}

int rt_render(rt_ctx* c, const rt_camera_params* cam, const rt_renderer_params* rnd, const rt_sky_params* sky, rt_rect tile,
              rt_id fb)
{
    return render_impl(c, cam, rnd, sky, tile, fb, 0, 0, 1);
}

int rt_render_bands(rt_ctx* c, const rt_camera_params* cam, const rt_renderer_params* rnd, const rt_sky_params* sky,
                    rt_rect tile, int32_t bandRows, int32_t firstBand, int32_t bandStride, rt_id fb)
{
    if (bandRows == 0) return fail(c, RT_ERR_INVALID, "rt_render_bands: bandRows must be a positive multiple of 8");
    return render_impl(c, cam, rnd, sky, tile, fb, bandRows, firstBand, bandStride);
}

// largest sample count per chunk whose workspace (wf_carve) fits the static array
static size_t scratch_slots(int levels)
{
    rt::WfBufs w;
    size_t lo = 0, hi = kScratchBytes / 64;   // slots, in units of one 8x8 tile at 1 spp
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        if (wf_carve(nullptr, mid * 64, levels, w) <= kScratchBytes) lo = mid; else hi = mid - 1;
    }
    return lo * 64;
}

int rt_render_noalloc(rt_ctx* c, const rt_camera_params* cam, const rt_renderer_params* rnd, const rt_sky_params* sky,
                      rt_rect tile, rt_id fb)
{
    RT_BIND(c);
    rt::RenderArgs a;
    const int rc = render_args(c, cam, rnd, sky, tile, fb, "rt_render_noalloc", a);
    if (rc) return rc;
    if (!c->scratchPtr) return fail(c, RT_ERR_STATE, "rt_render_noalloc: no static workspace on this device index");
    if (rnd->bounceCount > kScratchMaxLevels)
        return fail(c, RT_ERR_INVALID, "rt_render_noalloc: bounceCount beyond the maximum of 16");
    const int levels = rnd->bounceCount > 0 ? rnd->bounceCount : 0;
    const size_t capSlots = scratch_slots(levels);
    const size_t perTile = 64 * (size_t)rnd->sampleCount;
    if (perTile > capSlots)
        return fail(c, RT_ERR_INVALID, "rt_render_noalloc: sampleCount beyond the maximum of " + std::to_string(capSlots / 64) +
                                           " at this bounceCount (the samples of one 8x8 tile must fit the 64 MiB static workspace)");
    const size_t tiles = (size_t)((tile.x1 - tile.x0 + 7) / 8) * (size_t)((tile.y1 - tile.y0 + 7) / 8);
    RT_CHECK(c, hipMemsetAsync(c->dWork, 0, kWorkBytes, c->stream));
    if (!c->optAccumulate) RT_CHECK(c, hipMemsetAsync(c->dCounters, 0, kCntBytes, c->stream));
    RT_CHECK(c, hipEventRecord(c->evBegin, c->stream));
    if (tiles > 0) {
        // one lane on the caller's stream; the chunks (whole tiles, as many as fit) run one after the other, each
        // clearing the control words it needs at its start (launch_render_wf), so nothing of the last call is read
        size_t chunkTiles = capSlots / perTile;
        if (chunkTiles > tiles) chunkTiles = tiles;
        rt::WfLanes L = c->wf;
        L.n = 1; L.first = 0; L.tilesPerChunk = (uint32_t)chunkTiles;
        wf_carve((uint8_t*)c->scratchPtr, chunkTiles * perTile, levels, L.bufs[0]);
        rt::launch_render_wf(a, L, c->numCUs, c->stream);
    }
    RT_CHECK(c, hipGetLastError());
    RT_CHECK(c, hipEventRecord(c->evEnd, c->stream));
    c->renderedOnce = true;
    return RT_OK;
}

int rt_trace_rays(rt_ctx* c, const float* rays, int32_t n, uint32_t* out)
{
    RT_BIND(c);
    if (!c->committed) return fail(c, RT_ERR_STATE, "rt_trace_rays before rt_scene_commit");
    if (n < 0 || (n > 0 && (!rays || !out))) return fail(c, RT_ERR_INVALID, "ray arrays missing");
    if (n == 0) return RT_OK;
    float4* dr = nullptr; uint4* dout = nullptr;
    RT_CHECK(c, hipMalloc((void**)&dr, 32 * (size_t)n));
    hipError_t e = hipMalloc((void**)&dout, 16 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(dr, rays, 32 * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !c->optAccumulate) e = hipMemsetAsync(c->dCounters, 0, kCntBytes, c->stream);
    if (e == hipSuccess) {
        rt::launch_trace_rays(scene_view(c), dr, n, dout, c->dCounters, c->optTraversal, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dr); hipFree(dout);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("rt_trace_rays failed: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_event_record(rt_ctx* c, int slot)
{
    RT_BIND(c);
    if (slot < 0 || slot >= 14) return fail(c, RT_ERR_INVALID, "event slot must be 0..13");
    RT_CHECK(c, hipEventRecord(c->slots[slot], c->stream));
    return RT_OK;
}

int rt_event_elapsed_ms(rt_ctx* c, int a, int b, float* ms)
{
    RT_BIND(c);
    if (a < 0 || a >= 14 || b < 0 || b >= 14 || !ms) return fail(c, RT_ERR_INVALID, "event slot must be 0..13");
    RT_CHECK(c, hipEventSynchronize(c->slots[b]));
    RT_CHECK(c, hipEventElapsedTime(ms, c->slots[a], c->slots[b]));
    return RT_OK;
}

int rt_detmath_eval(rt_ctx* c, int op, const float* in, int32_t n, float* out)
{
    RT_BIND(c);
    if (op < 0 || op > 9 || n < 0 || (n > 0 && (!in || !out))) return fail(c, RT_ERR_INVALID, "bad detmath request");
    if (n == 0) return RT_OK;
    float *di = nullptr, *dout = nullptr;
    RT_CHECK(c, hipMalloc((void**)&di, 4 * (size_t)n));
    hipError_t e = hipMalloc((void**)&dout, 4 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(di, in, 4 * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { rt::launch_detmath(op, di, n, dout, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(di); hipFree(dout);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("rt_detmath_eval failed: ") + hipGetErrorString(e));
    return RT_OK;
}

} // extern "C"
