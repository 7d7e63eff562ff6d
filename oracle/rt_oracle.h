/*
 * rt_oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the reference's per-pixel
 * render path (elliot-chung/raytracer, src/processor/GPU/GPURaytracer.cu:4-20,78-885 and
 * src/Material.cu:268-323).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (raytracer_amd/) never does.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4), and its kernel is CUDA with texture hardware and libdevice math,
 * which cannot be compiled or run in this image.  The only reference-derived known answer
 * is the README screenshots' sky pixel, sRGB8 (104,104,139); tests/test_oracle.py pins
 * the oracle to it.  Everything else is "parity unpinned by the reference": this
 * restatement is the arbiter.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mesh.hpp:14-25 (GPUMeshData) -- host arrays, AoS exactly as the reference keeps them */
typedef struct {
    const float* pos;   /* 3*nVerts */
    const int*   idx;   /* 3*nTris  */
    const float* uv;    /* 2*nVerts, may be NULL (read as 0,0) */
    const float* nrm;   /* 3*nVerts, may be NULL -> face normal  */
    int nVerts;
    int nTris;
} or_mesh;

/* Material.cu:162-232: RGBA, value = byte/255.0f, bilinear, clamp, normalized coords */
typedef struct {
    const uint8_t* rgba; /* 4*w*h, row 0 first */
    int w, h;
} or_texture;

/* Material.hpp:53-77 (GPUMaterial) as Material::sendToGPU fills it (Material.cu:30-49):
 * only normal/albedo/roughness/metal textures reach the device. -1 = no texture. */
typedef struct {
    float albedo[4];
    float roughness;
    float metal;
    float emissionColor[3];
    float emissionStrength;
    int texNormal, texAlbedo, texRoughness, texMetal;
} or_material;

/* Scene.hpp:13-19 (GPUObjectData) + DisplayObject.hpp:17-28 (LLGPUObjectData) */
typedef struct {
    float model[16];    /* column-major, linal.hpp:5-9 */
    float invModel[16];
    float bmin[3], bmax[3]; /* composite bounds, object space */
    int meshCount;
    const int* meshIds;         /* meshCount indices into scene meshes */
    const int* materialIndices; /* meshCount indices into materialIds */
    int materialCount;
    const int* materialIds;     /* indices into scene materials */
} or_object;

typedef struct {
    int nMeshes;    const or_mesh* meshes;
    int nTextures;  const or_texture* textures;
    int nMaterials; const or_material* materials;
    int nObjects;   const or_object* objects; /* iteration order = array order */
} or_scene;

/* GPURaytracer.hpp:60-87 */
typedef struct {
    float origin[4];
    float rotation[4]; /* quaternion x,y,z,w */
    float exposure;
    int width, height;
    const float* rays; /* 4*width*height pre-rotation directions (Camera.hpp:46-70) */
} or_camera;

typedef struct {
    int bounceCount;
    float maxDistance;
    float aoIntensity;
    uint32_t frameCount;
    uint32_t progressiveFrameCount;
    int antiAliasingEnabled;
    uint32_t sampleCount;
} or_renderer;

typedef struct {
    float direction[4];
    float lightColor[4];
    float skyColor[4];
} or_sky;

/* Render pixels [x0,x1) x [y0,y1) of the frame into fb (4*width*height floats, row 0 =
 * bottom row as in the reference) exactly as raytraceKernel does, including the
 * progressive blend with the previous content of fb.  hitIds (may be NULL) receives 4
 * uint32 per pixel: object, mesh, triangle of the primary hit of sample 0 (0xFFFFFFFF on
 * miss) and a wrapping checksum over every path segment of every sample.  Returns the
 * number of closest-hit queries executed (GPURaytracer.cu:247 calls).  nThreads <= 0
 * means all cores. */
uint64_t or_render(const or_scene* scene, const or_camera* cam, const or_renderer* rnd,
                   const or_sky* sky, int x0, int y0, int x1, int y1,
                   float* fb, uint32_t* hitIds, int nThreads);

/* Closest hit for arbitrary rays (GPURaytracer.cu:247-265 semantics incl. the object
 * AABB cull).  rays: 8 floats each (origin xyz, maxDistance, direction xyz, pad).
 * out: 4 uint32 each (object, mesh, triangle, float bits of t); all 0xFFFFFFFF on miss. */
void or_trace_rays(const or_scene* scene, const float* rays, int nRays, uint32_t* out, int nThreads);

/* unit-test taps */
float    or_random_value(uint32_t* seed);                         /* :819-825 */
int      or_dist_to_triangle(const float* origin, const float* dir, float maxDistance,
                             const float* v0, const float* v1, const float* v2,
                             float* tOut, float* baryOut);       /* :551-637 */
int      or_slab(const float* origin, const float* dir, float maxDistance,
                 const float* invModel, const float* bmin, const float* bmax); /* :267-315 */
void     or_tex2d(const or_texture* t, float u, float v, float* out4);        /* Material.cu:224-228 */
void     or_tex2d_rays(const float* table, int w, int h, float x, float y, float* out4); /* Camera.hpp:93-97 */
void     or_sky_light(const float* dir, const or_sky* sky, float* out4);      /* :4-20 */
void     or_exposure(const float* rgb, float exposure, float* out3);          /* :690-698 */
float    or_logf(float x);
float    or_cosf(float x);
float    or_expf(float x);
float    or_powf(float x, float y);
float    or_rsqrtf(float x);
int      or_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
