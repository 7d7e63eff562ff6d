/*
 * rt_api.h -- C ABI of librt_amd.so: the MI355X (gfx950) implementation of the per-pixel
 * render path of elliot-chung/raytracer.  Plain pointers and sizes only; no C++/torch types.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference's src/).  Conventions:
 *  - every function returns 0 on success, a negative rt_status otherwise, and never calls
 *    exit() (the reference's checkCudaErrors does: helper_cuda.h:583-595);
 *    rt_last_error(ctx) returns the message of the last failure on that context;
 *  - handles are small non-negative integers scoped to their context;
 *  - one context per GPU, driven by one host thread at a time (the reference is single
 *    threaded and synchronous: processor/GPU/GPURaytracer.cu:65-66);
 *  - matrices are 16 floats, column-major, exactly linal.hpp:5-9 (mat4transfer);
 *  - framebuffers are float4 RGBA32F, row 0 = bottom row of the picture (Camera.hpp:65).
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;
typedef int32_t rt_id;

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / handle */
    RT_ERR_HIP = -2,       /* a HIP runtime call failed; message names it */
    RT_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    RT_ERR_STATE = -4      /* call order (e.g. render before commit) */
};

/* processor/GPU/GPURaytracer.hpp:60-69 (CameraParams; the ray texture is set separately) */
typedef struct {
    float origin[4];   /* x, y, z, w; w must be 1 (the reference always passes glm::vec4(position, 1.0f), GPURaytracer.cu:29): the
                          kernels transform the origin as a point and do not read w; rt_render rejects any other value */
    float rotation[4]; /* quaternion x,y,z,w */
    float exposure;
    int32_t width;
    int32_t height;
} rt_camera_params;

/* processor/GPU/GPURaytracer.hpp:78-87 (RendererParams) */
typedef struct {
    int32_t bounceCount;
    float maxDistance;
    float aoIntensity;
    uint32_t frameCount;
    uint32_t progressiveFrameCount;
    int32_t antiAliasingEnabled;
    uint32_t sampleCount;
} rt_renderer_params;

/* processor/GPU/GPURaytracer.hpp:71-76 (SkyLightParams) */
typedef struct {
    float direction[4];
    float lightColor[4];
    float skyColor[4];
} rt_sky_params;

typedef struct { int32_t x0, y0, x1, y1; } rt_rect; /* half-open pixel rectangle (image tile) */

/* Material.hpp:53-77 (GPUMaterial) as Material::sendToGPU fills it (Material.cu:30-49) */
typedef struct {
    float albedo[4];
    float roughness;
    float metal;
    float emissionColor[3];
    float emissionStrength;
    rt_id texNormal, texAlbedo, texRoughness, texMetal; /* -1 = none */
} rt_material_desc;

typedef struct {
    uint64_t rays;         /* closest-hit queries (GPURaytracer.cu:247 calls) of the last rt_render */
    float    kernel_ms;    /* HIP-event duration of the last rt_render's kernel(s) on the ctx stream */
    uint32_t triangles;    /* flattened world-space triangles of the committed scene */
    uint32_t bvh_nodes;    /* nodes of the committed acceleration structure */
    float    commit_ms;    /* device time of the last rt_scene_commit */
    uint64_t slow_path;    /* rays that took the exact in-order fallback in the last rt_render */
    uint64_t node_visits;  /* BVH node visits of the last rt_render */
    uint64_t tri_tests;    /* triangle tests of the last rt_render */
    uint64_t hits;         /* closest-hit queries that hit and were shaded */
    uint64_t packet_node_tests;   /* the part of node_visits done by camera-ray packets (one test per ray and node) ... */
    uint64_t packet_node_fetches; /* ... and the node fetches behind them: a packet of 64 rays fetches a node once.
                                     (Diagnostic builds, -DRT_UTIL_COUNTERS / _TIMERS, report other tallies in these two.) */
} rt_stats;

enum rt_option {
    RT_OPT_TRAVERSAL = 1, /* 0 = BVH, wavefront pipeline (default); 1 = brute force in reference order;
                             2 = BVH, single persistent kernel.  All three give the same pixels. */
    RT_OPT_COUNTERS = 2,  /* reserved */
    RT_OPT_ACCUMULATE_STATS = 3, /* 1 = rt_render no longer zeroes the counters; setting it zeroes them once */
    RT_OPT_WF_CHUNK = 4,  /* samples in flight per chunk of the wavefront pipeline (0 = default, 192 Mi; halved down to
                             64 Mi when the device cannot hold the workspace); the workspace holds ~185 bytes per sample
                             and lane and is sized by the frame: 6 GB for 1080p x 16 spp, 98 GB (three chunks in flight)
                             for 4K x 64 spp */
    RT_OPT_WF_LANES = 5,  /* chunks in flight at once, each on its own HIP stream (0 = default: one per chunk of the
                             frame, at most 3; 1..4) */
    RT_OPT_BUILDER = 6,   /* acceleration structure built by rt_scene_commit: 0 = PLOC (default: slower commit, faster
                             frames), 1 = Karras radix tree (scenes re-committed every frame).  Same pixels. */
    RT_OPT_REFIT = 8,     /* 1 (default): rt_scene_commit of the same objects and meshes under other matrices keeps the tree of
                             the last full build and refits its boxes (a moved 1 M-triangle scene in 1.6 ms instead of 5.5),
                             unless the refitted boxes sum to more than 1.3 times the built tree's area (the next frames
                             would then lose to the looser boxes what the commit saved, and the tree is built anew); 0: every commit builds;
                             101 ...: that limit in percent.  Same pixels either way. */
    RT_OPT_THIN_LIST = 9, /* 1 (default): camera rays and the exact walks are tested against the listed lines of the scene's
                             collinear triangles; 0: those tests are off and the grown sliver boxes of
                             the tree stand alone -- a diagnostic, for the test that shows they do for rays that start inside
                             the scene's bounds. */
    RT_OPT_WF_LANE = 7    /* frames in flight: the workspace lane (0 .. 3, default 0) the following rt_render / rt_render_bands
                             calls use first.  Two calls that use different lanes AND run on different streams (rt_set_stream)
                             may overlap on the GPU -- frame i + 1 fills the machine while frame i's last rays finish, which is
                             worth 5-10 % of throughput on short frames (one rank's share of a frame split over 8 GPUs);
                             calls on one lane must stay in stream order.  Each lane holds its own workspace (RT_OPT_WF_CHUNK).
                             While calls overlap, rt_stats is meaningful only with RT_OPT_ACCUMULATE_STATS (the counters
                             are shared), and kernel_ms not at all.  No counterpart in the synchronous reference. */
};

/* ---- context ---- */
int rt_ctx_create(int device, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx); /* ctx may be NULL: last error of a failed rt_ctx_create */
int rt_set_stream(rt_ctx* ctx, void* hip_stream); /* NULL = the context's own stream */
int rt_set_option(rt_ctx* ctx, int option, int value);
int rt_sync(rt_ctx* ctx);                     /* replaces cudaDeviceSynchronize, GPURaytracer.cu:66 */
int rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* ---- scene upload ---- */
/* Mesh::Mesh (Mesh.hpp:30-62): pos 3*nVerts, idx 3*nTris, uv 2*nVerts or NULL, nrm 3*nVerts or NULL */
int rt_mesh_create(rt_ctx* ctx, const float* pos, int32_t nVerts, const int32_t* idx, int32_t nTris,
                   const float* uv, const float* nrm, rt_id* out);
int rt_mesh_destroy(rt_ctx* ctx, rt_id mesh);
/* Texture::Texture (Material.cu:162-232) after decode + 4-channel expansion: rgba8, row 0 first */
int rt_texture_create(rt_ctx* ctx, const uint8_t* rgba8, int32_t width, int32_t height, rt_id* out);
int rt_texture_destroy(rt_ctx* ctx, rt_id texture);
/* Material::sendToGPU (Material.cu:30-49) */
int rt_material_create(rt_ctx* ctx, const rt_material_desc* desc, rt_id* out);
int rt_material_update(rt_ctx* ctx, rt_id material, const rt_material_desc* desc);
int rt_material_destroy(rt_ctx* ctx, rt_id material);
/* DisplayObject::sendToGPU (DisplayObject.cpp:30-81): mesh list, per-mesh material index,
 * material list, composite bounds in object space */
int rt_object_create(rt_ctx* ctx, const rt_id* meshes, const int32_t* materialIndices, int32_t nMeshes,
                     const rt_id* materials, int32_t nMaterials, const float bmin[3], const float bmax[3],
                     rt_id* out);
int rt_object_destroy(rt_ctx* ctx, rt_id object);
/* Scene::getGPUObjectDataVector (Scene.cpp:3-29): instance table in iteration order; flattens
 * the instances to world space and (re)builds the acceleration structure on the GPU */
int rt_scene_commit(rt_ctx* ctx, const rt_id* objects, const float* model16, const float* invModel16, int32_t n);
/* What the last rt_scene_commit did: 1 = built the acceleration structure, 2 = refitted the one of the last build (the same
 * object handles in the same order, no mesh created or destroyed since, RT_OPT_REFIT on, and the refitted boxes within the
 * growth limit), 0 = nothing committed.  No counterpart in the reference, which has no such structure. */
int rt_scene_commit_kind(const rt_ctx* ctx);

/* Camera::calcRays upload (Camera.hpp:73-102): W*H float4 pre-rotation directions (host memory) */
int rt_camera_set_rays(rt_ctx* ctx, const float* rays4, int32_t width, int32_t height);

/* ---- output (replaces the CUDA-GL surface, RenderWindow.cpp:215-234) ---- */
int rt_framebuffer_create(rt_ctx* ctx, int32_t width, int32_t height, rt_id* out);
/* wrap caller-owned device memory (e.g. a torch tensor) of width*height float4 */
int rt_framebuffer_wrap(rt_ctx* ctx, void* device_ptr, int32_t width, int32_t height, rt_id* out);
int rt_framebuffer_destroy(rt_ctx* ctx, rt_id fb);
int rt_framebuffer_read(rt_ctx* ctx, rt_id fb, float* host_rgba);         /* synchronises */
int rt_framebuffer_write(rt_ctx* ctx, rt_id fb, const float* host_rgba);  /* seed progressive state */
/* hit-id AOV: 4 uint32 per pixel = object, mesh, triangle of sample 0's primary hit
 * (0xFFFFFFFF = miss) and a checksum over every path segment of every sample */
int rt_framebuffer_enable_ids(rt_ctx* ctx, rt_id fb, int enable);
int rt_framebuffer_read_ids(rt_ctx* ctx, rt_id fb, uint32_t* host_ids);

/* The display and screenshot side of the reference's window (the "presenter"), without GL:
 *  - RT_PRESENT_RGBA32F: the frame as the float4 texels RenderWindow uploads to its RGBA32F texture
 *    (glTexImage2D / glTexSubImage2D, RenderWindow.cpp:172-176, 282-283): host_dst takes width*height*16 bytes and goes
 *    straight to glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w, h, GL_RGBA, GL_FLOAT, host_dst); row 0 = bottom, as GL wants it;
 *  - RT_PRESENT_SRGB8: what takeScreenshot reads back (RenderWindow.cpp:389-421: glReadPixels of the GL_FRAMEBUFFER_SRGB
 *    back buffer, RGBA8): clamp to [0,1], sRGB transfer function, 8 bits, converted on the GPU; host_dst takes
 *    width*height*4 bytes.  flip_rows != 0 also does the screenshot's vertical flip (row 0 = top, what lodepng is handed).
 * The 8-bit step is pinned: byte = number of k in 1..255 with value >= T[k-1], T = rt_srgb8_thresholds() (the linear
 * value whose sRGB encoding is (k - 0.5) / 255, computed in binary64, rounded to binary32), so device and host agree
 * on every byte.  Synchronises the context stream. */
enum rt_present_format { RT_PRESENT_RGBA32F = 0, RT_PRESENT_SRGB8 = 1 };
int rt_present(rt_ctx* ctx, rt_id fb, int format, int flip_rows, void* host_dst);
const float* rt_srgb8_thresholds(void); /* 255 ascending floats, host memory owned by the library */

/* GPURaytracer::raytrace (GPURaytracer.cu:24-74): one frame of `tile` into fb; asynchronous on
 * the context stream (rt_sync / rt_framebuffer_read wait).  frameCount bookkeeping stays with
 * the caller exactly as Raytracer keeps it (Raytracer.hpp:49-50, GPURaytracer.cu:69-73). */
int rt_render(rt_ctx* ctx, const rt_camera_params* camera, const rt_renderer_params* renderer,
              const rt_sky_params* sky, rt_rect tile, rt_id fb);

/* The image-tile split of one frame over several GPUs (no counterpart in the single-GPU reference; the pixels are those
 * of rt_render): `tile` is cut into bands of bandRows rows (a multiple of 8; band b = rows [y0 + b*bandRows, ...)) and
 * this call renders bands firstBand, firstBand + bandStride, ...  -- rank r of n calls it with (firstBand, bandStride) =
 * (r, n); interleaving balances sky against geometry.  The bands of a call run as ONE pass of the pipeline, not one per
 * band.  Pixel seeds use frame coordinates, so the bands of all ranks compose the rt_render frame bit for bit. */
int rt_render_bands(rt_ctx* ctx, const rt_camera_params* camera, const rt_renderer_params* renderer,
                    const rt_sky_params* sky, rt_rect tile, int32_t bandRows, int32_t firstBand, int32_t bandStride,
                    rt_id fb);

/* rt_render for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same pixels, ids and statistics as
 * rt_render with RT_OPT_TRAVERSAL = 0, but the wavefront workspace (path state, queues, hit records, redo lists)
 * lives in a 64 MiB static device array that every GPU the library runs on holds until the process ends, and the
 * call makes no hipMalloc / hipFree: the frame is rendered in as many consecutive chunks as that array needs
 * (~350 000 samples each), on the context stream.  Limits, rejected with RT_ERR_INVALID: bounceCount <= 16;
 * sampleCount <= ~5 000 (64 * sampleCount samples, one 8x8 tile, must fit).  The scene, the ray table and the
 * framebuffer must exist already.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (two contexts, two streams) race; calls in stream order do not. */
int rt_render_noalloc(rt_ctx* ctx, const rt_camera_params* camera, const rt_renderer_params* renderer,
                      const rt_sky_params* sky, rt_rect tile, rt_id fb);

/* closest-hit queries for caller-supplied rays (GPURaytracer.cu:247-265 semantics).
 * rays: 8 floats each (origin xyz, maxDistance, direction xyz, pad); out: 4 uint32 each
 * (object, mesh, triangle, float bits of t; all 0xFFFFFFFF on miss).  Host pointers.
 * Exact in every traversal mode, including the reference's order-dependent cases: ties between objects, hits that lie
 * before their object's bounding-box entry, and the noise hits its triangle test reports on triangles of no area
 * (collinear vertices), which the triangle's own bounding box does not stand for.  rt_render gives
 * the same pixels as the reference in all of these, by construction: the box the tree holds for a sliver is grown by what
 * the rounding noise can displace the reported hit (the whole length for a collinear triangle), so every walk reaches the
 * test whose answer the reference reports, for camera and bounce rays alike; camera rays, which may start far outside the
 * scene's bounds, are also tested against the listed lines where they are made and take the exact walks when they come
 * near. */
int rt_trace_rays(rt_ctx* ctx, const float* rays, int32_t nRays, uint32_t* out);

/* HIP-event timing on the context stream: record into slot 0..13, elapsed between two slots
 * (slots 14 and 15 belong to the library: they time rt_scene_commit) */
int rt_event_record(rt_ctx* ctx, int slot);
int rt_event_elapsed_ms(rt_ctx* ctx, int slot_begin, int slot_end, float* ms);

/* test tap: evaluate the pinned device math on n host floats.
 * op: 0 log, 1 cos, 2 exp, 3 pow(x,5), 4 pow(x,1000), 5 rsqrt, 6 random_value(seed bits), 7 byte/255 (input = byte as
 * integer bits), 8 exhaustive check of the 32-bit -> [0,1] conversion (input = high 16 bits; output = mismatch count), 9 exhaustive check of the
 * pow(x,5) / pow(x,1000) shortcuts of the shading code against ops 3 / 4 (input = high 16 bits of x; output = mismatch count) */
int rt_detmath_eval(rt_ctx* ctx, int op, const float* in, int32_t n, float* out);

/* stbi_load(path, &w, &h, &n, 0) / stbi_image_free of Texture::Texture (Material.cu:179, 195), host only (no context): decodes
 * a PNG (8- or 16-bit samples, the latter reduced with >> 8; gray, gray+alpha, RGB, RGBA, 8-bit palette as RGB; not
 * interlaced) or a baseline JPEG (stb_image's inverse DCT, chroma upsampling and colour conversion: the texel bytes are those
 * the reference reads, raytracer_amd/host/rt_jpeg.hpp) into width*height*channels bytes, row 0 first, channels as stored
 * (1, 2, 3 or 4).  Returns NULL for an unreadable or unsupported file (progressive JPEG, interlaced PNG, other formats).  The
 * 4-channel expansion of Material.cu:184-197 stays with the caller, as in the reference. */
uint8_t* rt_image_load(const char* path, int32_t* width, int32_t* height, int32_t* channels);
void rt_image_free(uint8_t* pixels);

const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
