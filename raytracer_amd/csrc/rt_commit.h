// rt_commit.h -- host-visible interface of the scene flatten + BVH build (rt_commit.hip) and of
// the kernel launchers (rt_render.hip).
#pragma once
#include "rt_device.h"
#include <string>
#include <vector>

namespace rt {

// one (object, mesh) pair of the committed scene; triangles [gidStart, gidStart + triCount)
struct Segment {
    uint32_t gidStart, triCount;
    uint32_t obj, meshInObj;
    uint32_t materialId, pad;
    const float* pos;
    const int32_t* idx;
    const float* uv;   // may be null
    const float* nrm;  // may be null
};

struct ObjXform {
    float model[16];
    float inv[16];
};

struct BuiltScene {
    TriIsect* triRef = nullptr;
    TriIsect* triLeaf = nullptr;
    TriShade* triShade = nullptr;
    BvhNode* nodes = nullptr;
    uint32_t nTris = 0;
    uint32_t nNodes = 0;
    float sceneMaxAbs = 0.0f;
    ThinTri* thin = nullptr;        // the (near) zero-area triangles, see thin_kernel
    uint32_t nThin = 0, capThin = 0;
    // kept across commits (dynamic scenes re-commit every frame, Scene.cpp:3-29): the record arrays
    // while the triangle count fits, and one arena for every scratch array of the build
    uint32_t capTris = 0;
    void* arena = nullptr;
    size_t arenaBytes = 0;
    // what a refit needs of the last full build (refit_scene): the four-wide nodes of level L are levelBegin[L] ..
    // levelBegin[L + 1] - 1 (the collapse allocates them level by level), places in the arena, and the summed surface
    // area of the node boxes the build ended with
    std::vector<uint32_t> levelBegin;
    uint32_t* sBounds = nullptr; uint32_t* sThinCount = nullptr; ThinTri* sThinTmp = nullptr; float* sBoxes = nullptr; float* sArea = nullptr;
    float buildArea = 0.0f;
    float lastArea = 0.0f;          // ... and the area after the last build or refit
};

// builder: 0 = PLOC (quality), 1 = Karras radix tree (fast rebuilds)
bool build_scene(const Segment* dSegs, int nSegs, const ObjXform* dXf, uint32_t nTris, BuiltScene& out, int builder,
                 hipStream_t stream, std::string& err);
// The dynamic-scene path (the reference's per-frame cost for a moved object is one glm::inverse and a
// 136-byte copy, Scene.cpp:3-29): the same instances and meshes under other matrices.  Flattens again, keeps the topology
// of the four-wide tree and fits its boxes bottom up, one launch per level; no Morton codes, no sort, no clustering, no
// collapse.  Returns false with err empty when the tree cannot be refitted (never built, a single triangle) or has
// degraded -- the summed area of the node boxes exceeds maxGrowth times that of the last full build -- and the caller
// rebuilds.  Any tree over the same triangles gives the same pixels: only the frame time depends on this choice.
bool refit_scene(const Segment* dSegs, int nSegs, const ObjXform* dXf, uint32_t nTris, BuiltScene& out, float maxGrowth,
                 hipStream_t stream, std::string& err);
void free_scene(BuiltScene& s);

void launch_render(const RenderArgs& a, int mode, int numCUs, hipStream_t stream);
// lanes of the wavefront pipeline: lane 0 runs on the caller's stream, the others on streams[1..n-1]
constexpr int kWfMaxLanes = 4;
struct WfLanes {
    int n = 0;
    int first = 0;                    // the call's lanes are first .. first + n - 1 (RT_OPT_WF_LANE); lane `first` runs on the caller's stream
    uint32_t tilesPerChunk = 0;
    hipStream_t streams[kWfMaxLanes] = {};
    hipEvent_t fork = nullptr;
    hipEvent_t join[kWfMaxLanes] = {};
    WfBufs bufs[kWfMaxLanes] = {};
};
int wf_tile_rows(const RenderArgs& a);
void launch_render_wf(const RenderArgs& a, const WfLanes& L, int numCUs, hipStream_t mainStream);
void launch_trace_rays(const SceneView& s, const float4* rays, int n, uint4* out, unsigned long long* counters, int mode,
                       hipStream_t stream);
void launch_present_srgb8(const float4* fb, int w, int h, bool flip, const float* thr, uchar4* out, hipStream_t stream);
void launch_detmath(int op, const float* in, int n, float* out, hipStream_t stream);

} // namespace rt
