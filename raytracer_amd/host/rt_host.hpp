// rt_host.hpp -- C++ host mirror of the reference's scene-description and renderer API, on top
// of the C ABI (include/rt_api.h).  Same class and method names, argument meaning and bool error
// conventions as the reference (src/Material.hpp, Mesh.hpp, DisplayObject.hpp/.cpp, Cube.hpp,
// Sphere.hpp, Scene.hpp/.cpp, Camera.hpp, processor/Raytracer.hpp, processor/GPU/GPURaytracer.hpp),
// minus the ImGui editors, the process-wide registries and the GL interop.  Header-only; link
// with librt_amd.so and zlib.  Build with -ffp-contract=off: the GLM restatements below must
// produce the same bits as raytracer_amd/hostmath.py (tests/test_host_cpp.py checks it).
#pragma once
#include "../../include/rt_api.h"

#include <zlib.h>
#include "rt_assets.hpp"
#include "rt_jpeg.hpp"
#include "rt_png.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace rth {

struct vec3 { float x = 0, y = 0, z = 0; };
struct vec4 { float x = 0, y = 0, z = 0, w = 0; };
struct quat { float x = 0, y = 0, z = 0, w = 0; };  // glm::quat() value-initialises to zero
struct mat4 { float c[4][4]; };                      // c[col][row], GLM convention

inline vec4 col(const mat4& m, int i) { return { m.c[i][0], m.c[i][1], m.c[i][2], m.c[i][3] }; }
inline void setcol(mat4& m, int i, vec4 v) { m.c[i][0] = v.x; m.c[i][1] = v.y; m.c[i][2] = v.z; m.c[i][3] = v.w; }
inline vec4 operator*(vec4 a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline vec4 operator*(vec4 a, vec4 b) { return { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
inline vec4 operator+(vec4 a, vec4 b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline vec4 operator-(vec4 a, vec4 b) { return { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }

inline mat4 identity() { mat4 m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.c[i][j] = i == j ? 1.0f : 0.0f; return m; }
// glm ext/matrix_transform.inl:10-15
inline mat4 translate(const mat4& m, vec3 v) { mat4 r = m; setcol(r, 3, col(m, 0) * v.x + col(m, 1) * v.y + col(m, 2) * v.z + col(m, 3)); return r; }
// glm gtc/quaternion.inl:41-72
inline mat4 mat4_cast(quat q)
{
    mat4 r = identity();
    const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    const float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r.c[0][0] = 1.0f - 2.0f * (qyy + qzz); r.c[0][1] = 2.0f * (qxy + qwz); r.c[0][2] = 2.0f * (qxz - qwy);
    r.c[1][0] = 2.0f * (qxy - qwz); r.c[1][1] = 1.0f - 2.0f * (qxx + qzz); r.c[1][2] = 2.0f * (qyz + qwx);
    r.c[2][0] = 2.0f * (qxz + qwy); r.c[2][1] = 2.0f * (qyz - qwx); r.c[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}
// glm detail/type_mat4x4.inl:630-648
inline mat4 mul(const mat4& a, const mat4& b)
{
    mat4 r;
    for (int c = 0; c < 4; c++) setcol(r, c, col(a, 0) * b.c[c][0] + col(a, 1) * b.c[c][1] + col(a, 2) * b.c[c][2] + col(a, 3) * b.c[c][3]);
    return r;
}
// glm ext/matrix_transform.inl:78-86
inline mat4 scale(const mat4& m, vec3 v) { mat4 r; setcol(r, 0, col(m, 0) * v.x); setcol(r, 1, col(m, 1) * v.y); setcol(r, 2, col(m, 2) * v.z); setcol(r, 3, col(m, 3)); return r; }
// glm detail/func_matrix.inl:294-352
inline mat4 inverse(const mat4& mm)
{
    const float(*m)[4] = mm.c;
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const vec4 f0{ c00, c00, c02, c03 }, f1{ c04, c04, c06, c07 }, f2{ c08, c08, c10, c11 };
    const vec4 f3{ c12, c12, c14, c15 }, f4{ c16, c16, c18, c19 }, f5{ c20, c20, c22, c23 };
    const vec4 v0{ m[1][0], m[0][0], m[0][0], m[0][0] }, v1{ m[1][1], m[0][1], m[0][1], m[0][1] };
    const vec4 v2{ m[1][2], m[0][2], m[0][2], m[0][2] }, v3{ m[1][3], m[0][3], m[0][3], m[0][3] };
    const vec4 i0 = v1 * f0 - v2 * f1 + v3 * f2, i1 = v0 * f0 - v2 * f3 + v3 * f4;
    const vec4 i2 = v0 * f1 - v1 * f3 + v3 * f5, i3 = v0 * f2 - v1 * f4 + v2 * f5;
    const vec4 sa{ 1, -1, 1, -1 }, sb{ -1, 1, -1, 1 };
    mat4 inv; setcol(inv, 0, i0 * sa); setcol(inv, 1, i1 * sb); setcol(inv, 2, i2 * sa); setcol(inv, 3, i3 * sb);
    const vec4 row0{ inv.c[0][0], inv.c[1][0], inv.c[2][0], inv.c[3][0] };
    const vec4 d0 = col(mm, 0) * row0;
    const float dot1 = (d0.x + d0.y) + (d0.z + d0.w);
    const float ood = 1.0f / dot1;
    mat4 r; for (int c = 0; c < 4; c++) setcol(r, c, col(inv, c) * ood);
    return r;
}

// ---------------------------------------------------------------- context
class Device {
public:
    explicit Device(int index = 0) { if (rt_ctx_create(index, &ctx_) != RT_OK) throw std::runtime_error(std::string("rt_ctx_create: ") + rt_last_error(nullptr)); }
    ~Device() { rt_ctx_destroy(ctx_); }
    Device(const Device&) = delete;
    rt_ctx* ctx() const { return ctx_; }
    // library options (include/rt_api.h, enum rt_option): e.g. setOption(RT_OPT_BUILDER, 1) for a scene that is
    // re-committed every frame (the reference's Scene::getGPUObjectDataVector pattern)
    void setOption(int option, int value) const { check(rt_set_option(ctx_, option, value)); }
    void check(int rc) const { if (rc != RT_OK) throw std::runtime_error(std::string("librt_amd: ") + rt_last_error(ctx_)); }
    // path-keyed sharing, per context instead of the reference's process-wide statics (textureMap, Material.cu:165-178;
    // customModels, CustomModel.cpp:10-24): an entry lives while any Texture / CustomModel of that path does
    std::unordered_map<std::string, std::weak_ptr<void>> textureMap, modelMap;
    int textureUploads = 0, modelParses = 0;   // what actually reached the device / the parser (tests)
    // what Scene::commit needs to tell "nothing changed" from "the same handle numbers, other contents": every mesh and
    // object this mirror creates or destroys on the context counts, and so does whichever scene committed last
    unsigned long long uploadEpoch = 0;
    const void* lastCommitter = nullptr;
    int sceneCommits = 0;                      // rt_scene_commit calls made (tests)
private:
    rt_ctx* ctx_ = nullptr;
};

// ---------------------------------------------------------------- Texture / Material (Material.hpp:17-117)
class Texture {
public:
    // stb_image in the reference; this mirror decodes PNG (8- and 16-bit, the latter reduced with >> 8 as stb_image.h:1198
    // does; rt_png.hpp) and baseline JPEG (rt_jpeg.hpp: stb_image's inverse DCT, upsampling and colour conversion, so the
    // texel bytes are the reference's) and refuses everything else: getLoadSuccess() is then false and
    // Material::set*Texture returns false, the reference's own convention for an unreadable file (Material.cu:113-130).
    Texture(Device& dev, const char* path) : path_(path)
    {
        auto it = dev.textureMap.find(path_);
        if (it != dev.textureMap.end()) if (auto live = std::static_pointer_cast<Store>(it->second.lock())) { store_ = live; return; } // already loaded: shared
        store_ = std::make_shared<Store>(dev);
        dev.textureMap[path_] = store_;
        std::vector<unsigned char> pix; int n = 0;
        if (!png::decode(path_, pix, store_->width, store_->height, n) && !jpeg::decode(path_, pix, store_->width, store_->height, n)) return;
        store_->upload(pix.data(), n);
    }
    Texture(Device& dev, const unsigned char* pixels, int w, int h, int nChannels) : store_(std::make_shared<Store>(dev))
    {
        store_->width = w; store_->height = h;
        store_->upload(pixels, nChannels);
    }
    Texture(const Texture&) = delete;
    int getWidth() const { return store_->width; }
    int getHeight() const { return store_->height; }
    bool getLoadSuccess() const { return store_->loadSuccess; }
    std::string getPath() const { return path_; }
    rt_id id() const { return store_->id; }
    // Texture::sampleTexture (Material.cu:253-266), the legacy CPU renderer's nearest-texel fetch: (0,0,0,0) outside [0,1]^2,
    // texel (int)(x * width), (int)(y * height) of the 4-channel float copy (value = byte / 255.0f, :199-206).  The
    // reference reads past the row / the array at x == 1 or y == 1; those reads give 0 here.
    vec4 sampleTexture(float x, float y) const
    {
        if (x < 0.0f || x > 1.0f || y < 0.0f || y > 1.0f) return {};
        const int ix = (int)(x * store_->width), iy = (int)(y * store_->height);
        const size_t index = ((size_t)iy * store_->width + ix) * 4;
        auto at = [&](size_t k) { return k < store_->rgba.size() ? (float)store_->rgba[k] / 255.0f : 0.0f; };
        return { at(index), at(index + 1), at(index + 2), at(index + 3) };
    }
private:
    struct Store { // the device texture the Textures of one path share; freed with the last of them (ref count, Material.cu:234-252)
        explicit Store(Device& d) : dev(d) {}
        ~Store() { if (id >= 0) rt_texture_destroy(dev.ctx(), id); }
        void upload(const unsigned char* idata, int n)
        {
            // 4-channel expansion with the reference's quirk (Material.cu:184-197): G,B = following bytes
            const size_t np = (size_t)width * height, total = np * n;
            std::vector<unsigned char> rgba(np * 4);
            auto at = [&](size_t k) -> unsigned char { return k < total ? idata[k] : 0; };
            for (size_t i = 0; i < np; i++) { rgba[i * 4] = at(i * n); rgba[i * 4 + 1] = at(i * n + 1); rgba[i * 4 + 2] = at(i * n + 2); rgba[i * 4 + 3] = n == 4 ? at(i * n + 3) : 255; }
            loadSuccess = rt_texture_create(dev.ctx(), rgba.data(), width, height, &id) == RT_OK;
            dev.textureUploads++;
            this->rgba = std::move(rgba); // host copy, as the reference keeps one (Texture::data) for its CPU renderer
        }
        Device& dev;
        std::vector<unsigned char> rgba;
        int width = 0, height = 0;
        bool loadSuccess = false;
        rt_id id = -1;
    };
    std::string path_;
    std::shared_ptr<Store> store_;
};

class Material {
public:
    Material(Device& dev, const char* name, vec4 albedo = { 0.9f, 0.9f, 0.9f, 1.0f }, float roughness = 0.5f, float metal = 0.0f,
             vec3 emissionColor = {}, float emission = 0) : dev_(dev), materialName(name), simpleAlbedo(albedo), simpleRoughness(roughness),
                                                            simpleMetal(metal), simpleEmissionColor(emissionColor), emissionStrength(emission) {}
    ~Material() { if (id_ >= 0) rt_material_destroy(dev_.ctx(), id_); }
    Material(const Material&) = delete;
    std::string getName() const { return materialName; }
    bool getSuccess() const { return success; }
    bool setNormalTexture(const char* p) { return setRef(normalTexture, p); }
    bool setAlbedoTexture(const char* p) { return setRef(albedoTexture, p); }
    bool setRoughnessTexture(const char* p) { return setRef(roughnessTexture, p); }
    bool setMetalTexture(const char* p) { return setRef(metalTexture, p); }
    bool setAmbientOcclusionTexture(const char* p) { return setRef(ambientOcclusionTexture, p); } // never reaches the device (Material.cu:41-44)
    bool setEmissionTexture(const char* p) { return setRef(emissionTexture, p); }
    void setAlbedoTexture(std::shared_ptr<Texture> t) { albedoTexture = std::move(t); }
    void setNormalTexture(std::shared_ptr<Texture> t) { normalTexture = std::move(t); }
    void setRoughnessTexture(std::shared_ptr<Texture> t) { roughnessTexture = std::move(t); }
    void setMetalTexture(std::shared_ptr<Texture> t) { metalTexture = std::move(t); }
    // Material::sendToGPU (Material.cu:30-49)
    void sendToGPU()
    {
        rt_material_desc d;
        d.albedo[0] = simpleAlbedo.x; d.albedo[1] = simpleAlbedo.y; d.albedo[2] = simpleAlbedo.z; d.albedo[3] = simpleAlbedo.w;
        d.roughness = simpleRoughness; d.metal = simpleMetal;
        d.emissionColor[0] = simpleEmissionColor.x; d.emissionColor[1] = simpleEmissionColor.y; d.emissionColor[2] = simpleEmissionColor.z;
        d.emissionStrength = emissionStrength;
        d.texNormal = normalTexture ? normalTexture->id() : -1; d.texAlbedo = albedoTexture ? albedoTexture->id() : -1;
        d.texRoughness = roughnessTexture ? roughnessTexture->id() : -1; d.texMetal = metalTexture ? metalTexture->id() : -1;
        dev_.check(id_ < 0 ? rt_material_create(dev_.ctx(), &d, &id_) : rt_material_update(dev_.ctx(), id_, &d));
    }
    rt_id id() { if (id_ < 0) sendToGPU(); return id_; }
    // host-side getters of the legacy CPU renderer (Material.cu:58-110): constant, or nearest texel of the bound map
    vec3 getNormal(float x, float y) const { if (!normalTexture) return { 0.0f, 0.0f, 1.0f }; const vec4 t = normalTexture->sampleTexture(x, y); return { t.x, t.y, t.z }; }
    vec4 getAlbedo(float x, float y) const { return albedoTexture ? albedoTexture->sampleTexture(x, y) : simpleAlbedo; }
    float getRoughness(float x, float y) const { return roughnessTexture ? roughnessTexture->sampleTexture(x, y).x : simpleRoughness; }
    float getMetal(float x, float y) const { return metalTexture ? metalTexture->sampleTexture(x, y).x : simpleMetal; }
    vec3 getAmbientOcclusion(float x, float y) const { if (!ambientOcclusionTexture) return { 1.0f, 1.0f, 1.0f }; const vec4 t = ambientOcclusionTexture->sampleTexture(x, y); return { t.x, t.y, t.z }; }
    vec3 getEmissionColor(float x, float y) const { if (!emissionTexture) return simpleEmissionColor; const vec4 t = emissionTexture->sampleTexture(x, y); return { t.x, t.y, t.z }; }
    float getEmissionStrength() const { return emissionStrength; }
private:
    bool setRef(std::shared_ptr<Texture>& slot, const char* path)
    {
        if (!path || !path[0]) { slot.reset(); return true; } // setTextureRef, Material.cu:113-130
        auto t = std::make_shared<Texture>(dev_, path);
        if (!t->getLoadSuccess()) return false;
        slot = std::move(t);
        return true;
    }
    Device& dev_;
    std::string materialName;
public:
    vec4 simpleAlbedo; float simpleRoughness, simpleMetal; vec3 simpleEmissionColor; float emissionStrength;
private:
    std::shared_ptr<Texture> normalTexture, albedoTexture, roughnessTexture, metalTexture, ambientOcclusionTexture, emissionTexture;
    rt_id id_ = -1;
    bool success = true;
};

// ---------------------------------------------------------------- Mesh (Mesh.hpp:27-62)
class Mesh {
public:
    Mesh(Device& dev, const std::vector<float>& vertices, const std::vector<int>& indices, const std::vector<float>& uvCoords,
         const std::vector<float>& normals) : dev_(dev), vertices(vertices), uvCoords(uvCoords), indices(indices), normals(normals)
    {
        triangleCount = (int)indices.size() / 3;
        for (size_t i = 0; i + 2 < vertices.size(); i += 3) // calcBounds, Mesh.hpp:106-122
            for (int k = 0; k < 3; k++) { minBound[k] = std::min(minBound[k], vertices[i + k]); maxBound[k] = std::max(maxBound[k], vertices[i + k]); }
        dev_.uploadEpoch++;
        dev_.check(rt_mesh_create(dev_.ctx(), vertices.data(), (int)vertices.size() / 3, indices.data(), triangleCount,
                                  uvCoords.empty() ? nullptr : uvCoords.data(), normals.empty() ? nullptr : normals.data(), &id_));
    }
    ~Mesh() { if (id_ >= 0) { rt_mesh_destroy(dev_.ctx(), id_); dev_.uploadEpoch++; } }
    Mesh(const Mesh&) = delete;
    const float* getMinBound() const { return minBound; }
    const float* getMaxBound() const { return maxBound; }
    int getTriangleCount() const { return triangleCount; }
    rt_id id() const { return id_; }
    const std::vector<float>& getVertices() const { return vertices; }
    const std::vector<float>& getUVCoords() const { return uvCoords; }
    const std::vector<int>& getIndices() const { return indices; }
    const std::vector<float>& getNormals() const { return normals; }
private:
    Device& dev_;
    std::vector<float> vertices, uvCoords; std::vector<int> indices; std::vector<float> normals;
    float minBound[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, maxBound[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    int triangleCount = 0;
    rt_id id_ = -1;
};

// ---------------------------------------------------------------- DisplayObject (DisplayObject.hpp:31-105, .cpp:19-105)
class DisplayObject {
public:
    DisplayObject(Device& dev, vec3 position = {}, quat rotation = {}, vec3 scl = { 1, 1, 1 }) : dev_(dev), Position(position), Rotation(rotation), Scale(scl) {}
    virtual ~DisplayObject() { if (id_ >= 0) { rt_object_destroy(dev_.ctx(), id_); dev_.uploadEpoch++; } }
    mat4 getModelMatrix() const { mat4 m = identity(); m = translate(m, Position); m = mul(m, mat4_cast(Rotation)); return scale(m, Scale); }
    mat4 getInverseModelMatrix() const { return inverse(getModelMatrix()); }
    void setPosition(vec3 p) { Position = p; }
    void setRotation(quat q) { Rotation = q; }
    void setScale(vec3 s) { Scale = s; }
    void setMaterial(std::shared_ptr<Material> m) { materials.push_back(std::move(m)); } // setMaterialName without the global registry
    bool isCompositeObject() const { return isComposite; }
    const float* getMaxBound() const { return compositeMaxBounds; }
    const float* getMinBound() const { return compositeMinBounds; }
    bool getLoadSuccess() const { return loadSuccess; }
    const std::vector<std::pair<std::shared_ptr<Mesh>, int>>& getMeshes() const { return meshes; }
    const std::vector<std::shared_ptr<Material>>& getMaterials() const { return materials; }
    Mesh* getMesh() const { return meshes[0].first.get(); }                          // DisplayObject.hpp:49: the first mesh only
    Material* getMaterial() const { return materials[meshes[0].second].get(); }      // :50
    void sendToGPU()
    {
        std::vector<rt_id> mi, ma; std::vector<int32_t> mx;
        for (auto& p : meshes) { mi.push_back(p.first->id()); mx.push_back(p.second); }
        for (auto& m : materials) ma.push_back(m->id());
        if (id_ >= 0) { rt_object_destroy(dev_.ctx(), id_); id_ = -1; }
        dev_.uploadEpoch++;
        dev_.check(rt_object_create(dev_.ctx(), mi.data(), mx.data(), (int)mi.size(), ma.data(), (int)ma.size(), compositeMinBounds, compositeMaxBounds, &id_));
    }
    rt_id id() { if (id_ < 0) sendToGPU(); return id_; }
protected:
    void updateCompositeBounds()
    {
        if (!isComposite) { memcpy(compositeMaxBounds, meshes[0].first->getMaxBound(), 12); memcpy(compositeMinBounds, meshes[0].first->getMinBound(), 12); return; }
        for (int k = 0; k < 3; k++) { compositeMaxBounds[k] = -FLT_MAX; compositeMinBounds[k] = FLT_MAX; }
        for (auto& p : meshes)
            for (int k = 0; k < 3; k++) { compositeMaxBounds[k] = fmaxf(p.first->getMaxBound()[k], compositeMaxBounds[k]); compositeMinBounds[k] = fminf(p.first->getMinBound()[k], compositeMinBounds[k]); }
    }
    Device& dev_;
    vec3 Position; quat Rotation; vec3 Scale;
    float compositeMaxBounds[3] = { 0, 0, 0 }, compositeMinBounds[3] = { 0, 0, 0 };
    std::vector<std::pair<std::shared_ptr<Mesh>, int>> meshes;
    std::vector<std::shared_ptr<Material>> materials;
    bool isComposite = false, loadSuccess = false;
    rt_id id_ = -1;
};

// Cube.hpp: 36 vertices / 12 triangles, UVs, no normals; the constructor drops its transform (Cube.hpp:8)
class Cube : public DisplayObject {
public:
    explicit Cube(Device& dev, vec3 = {}, quat = {}, vec3 = { 1, 1, 1 }) : DisplayObject(dev)
    {
        static const signed char faces[6][4][3] = {
            { { -1, -1, -1 }, { 1, -1, -1 }, { 1, 1, -1 }, { -1, 1, -1 } }, { { 1, -1, 1 }, { -1, -1, 1 }, { -1, 1, 1 }, { 1, 1, 1 } },
            { { -1, -1, 1 }, { -1, -1, -1 }, { -1, 1, -1 }, { -1, 1, 1 } }, { { 1, -1, -1 }, { 1, -1, 1 }, { 1, 1, 1 }, { 1, 1, -1 } },
            { { 1, -1, -1 }, { -1, -1, -1 }, { -1, -1, 1 }, { 1, -1, 1 } }, { { 1, 1, 1 }, { -1, 1, 1 }, { -1, 1, -1 }, { 1, 1, -1 } } };
        static const int order[6] = { 0, 1, 2, 2, 3, 0 };
        static const float cuv[4][2] = { { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 } };
        std::vector<float> v, uv; std::vector<int> idx;
        for (auto& f : faces) for (int k : order) { for (int c = 0; c < 3; c++) v.push_back(0.5f * f[k][c]); uv.push_back(cuv[k][0]); uv.push_back(cuv[k][1]); }
        for (int i = 0; i < 36; i++) idx.push_back(i);
        meshes.push_back({ std::make_shared<Mesh>(dev, v, idx, uv, std::vector<float>()), 0 });
        updateCompositeBounds();
        loadSuccess = true;
    }
};

// Sphere.hpp: 16x12 UV sphere, r = 0.5, normals = 2*position, PI = 3.14159f (Sphere.hpp:7)
class Sphere : public DisplayObject {
public:
    explicit Sphere(Device& dev, vec3 position = {}, quat rotation = {}, vec3 scl = { 1, 1, 1 }) : DisplayObject(dev, position, rotation, scl)
    {
        const int SX = 16, SY = 12; const float PI = 3.14159f;
        std::vector<float> v, n, uv; std::vector<int> idx;
        auto s = [](float a) { return (float)std::sin((double)a); }; auto c = [](float a) { return (float)std::cos((double)a); }; // pinned: binary64, rounded once
        for (int i = 0; i < SX; i++) { v.insert(v.end(), { 0.0f, -0.5f, 0.0f }); n.insert(n.end(), { 0.0f, -1.0f, 0.0f }); uv.push_back((float)(2 * i + 1) / (2 * SX)); uv.push_back(0.0f); }
        for (int i = 1; i < SY; i++) for (int j = 0; j <= SX; j++) {
            const float theta = (float)i / SY * PI, phi = (float)j / SX * 2 * PI;
            const float x = 0.5f * s(theta) * c(phi), y = -0.5f * c(theta), z = 0.5f * s(theta) * s(phi);
            v.insert(v.end(), { x, y, z }); n.insert(n.end(), { x * 2, y * 2, z * 2 }); uv.push_back((float)j / SX); uv.push_back((float)i / SY);
        }
        for (int i = 0; i < SX; i++) { v.insert(v.end(), { 0.0f, 0.5f, 0.0f }); n.insert(n.end(), { 0.0f, 1.0f, 0.0f }); uv.push_back((float)(2 * i + 1) / (2 * SX)); uv.push_back(1.0f); }
        for (int i = 0; i < SX; i++) idx.insert(idx.end(), { i, i + SX + 1, i + SX });
        for (int i = 0; i < SY - 2; i++) for (int j = 0; j < SX; j++) { const int a = SX + (SX + 1) * i + j, b = SX + (SX + 1) * (i + 1) + j; idx.insert(idx.end(), { a, b + 1, b, a, a + 1, b + 1 }); }
        const int bottom = SX + (SX + 1) * (SY - 2);
        for (int i = 0; i < SX; i++) idx.insert(idx.end(), { bottom + i, bottom + i + 1, bottom + i + 1 + SX });
        meshes.push_back({ std::make_shared<Mesh>(dev, v, idx, uv, n), 0 });
        updateCompositeBounds();
        loadSuccess = true;
    }
};

// What CustomModel (CustomModel.cpp:135-145) is after import: explicit (mesh, materialIndex) pairs
class MeshObject : public DisplayObject {
public:
    MeshObject(Device& dev, std::vector<std::pair<std::shared_ptr<Mesh>, int>> m, std::vector<std::shared_ptr<Material>> mats, vec3 position = {},
               quat rotation = {}, vec3 scl = { 1, 1, 1 }) : DisplayObject(dev, position, rotation, scl)
    {
        meshes = std::move(m); materials = std::move(mats);
        isComposite = meshes.size() > 1;
        if (!meshes.empty()) { updateCompositeBounds(); loadSuccess = true; }
    }
};

// ---------------------------------------------------------------- CustomModel (CustomModel.hpp, CustomModel.cpp:47-194)
// A model file turned into (Mesh, materialIndex) pairs and Materials by the native flatteners of
// rt_assets.hpp (COLLADA, OBJ/MTL, binary FBX) in place of assimp.  A file that cannot be read gives
// getLoadSuccess() == false; textures are looked up next to the asset, or next to its "source"
// directory (:110-112); a texture that fails to load leaves the material's constant (the bool
// convention of setOutputTexture, :76-101).  This host mirror decodes PNG only.
class CustomModel : public MeshObject {
public:
    CustomModel(Device& dev, const std::string& path, vec3 position = {}, quat rotation = {}, vec3 scl = { 1, 1, 1 })
        : MeshObject(dev, {}, {}, position, rotation, scl), path_(path)
    {
        auto it = dev.modelMap.find(path);
        if (it != dev.modelMap.end()) if (auto live = std::static_pointer_cast<Data>(it->second.lock())) {
            data_ = live;   // copyHostLLData (CustomModel.cpp:10-19): the same Mesh and Material objects, nothing parsed or uploaded
            meshes = data_->meshes; materials = data_->materials;
            isComposite = meshes.size() > 1;
            if (!meshes.empty()) { updateCompositeBounds(); loadSuccess = true; }
            return;
        }
        data_ = std::make_shared<Data>();
        dev.modelMap[path] = data_;
        dev.modelParses++;
        assets::FlatAsset a;
        if (!assets::loadAsset(path, a)) return;
        const std::string dir = a.textureDirectory();
        for (auto& fm : a.materials) {
            auto m = std::make_shared<Material>(dev, fm.name.c_str());
            for (auto& t : fm.textures) {
                if (t.second.empty()) continue;
                const std::string p = dir + "/" + t.second;
                if (t.first == "albedo") m->setAlbedoTexture(p.c_str());
                else if (t.first == "normal") m->setNormalTexture(p.c_str());
                else if (t.first == "roughness") m->setRoughnessTexture(p.c_str());
                else if (t.first == "metal") m->setMetalTexture(p.c_str());
                else if (t.first == "ao") m->setAmbientOcclusionTexture(p.c_str());
                else if (t.first == "emission") m->setEmissionTexture(p.c_str());
            }
            m->sendToGPU();
            materials.push_back(m);
        }
        for (auto& fm : a.meshes) {
            if (fm.indices.empty()) continue;
            meshes.emplace_back(std::make_shared<Mesh>(dev, fm.positions, fm.indices, fm.uvs, fm.normals), fm.material);
        }
        data_->meshes = meshes; data_->materials = materials;
        isComposite = meshes.size() > 1;
        if (!meshes.empty()) { updateCompositeBounds(); loadSuccess = true; }
    }
    std::string getPath() const { return path_; }
private:
    struct Data { std::vector<std::pair<std::shared_ptr<Mesh>, int>> meshes; std::vector<std::shared_ptr<Material>> materials; };
    std::string path_;
    std::shared_ptr<Data> data_;
};

// ---------------------------------------------------------------- Scene (Scene.hpp:41-176, Scene.cpp:3-29); iteration = insertion order
class Scene {
public:
    explicit Scene(Device& dev) : dev_(dev) {}
    bool addToScene(std::string name, std::shared_ptr<DisplayObject> object)
    {
        for (auto& p : objects_) if (p.first == name) return false;
        objects_.push_back({ std::move(name), std::move(object) });
        return true;
    }
    bool removeFromSceen(const std::string& name) // sic
    {
        for (auto it = objects_.begin(); it != objects_.end(); ++it) if (it->first == name) { objects_.erase(it); return true; }
        return false;
    }
    std::vector<std::pair<std::string, std::shared_ptr<DisplayObject>>>& getObjects() { return objects_; }
    void sendToGPU() { for (auto& p : objects_) p.second->sendToGPU(); }
    // getGPUObjectDataVector: instance table = (model, inverse(model)) per object -> rt_scene_commit.  The reference builds
    // that table every frame; here the call is made when something changed (scene->dirty()) -- the object
    // list, a transform, or any mesh / object created or destroyed on the context since this scene's last commit -- and the
    // library then refits or rebuilds (rt_scene_commit_kind).  force = true commits regardless.
    void commit(bool force = false)
    {
        std::vector<rt_id> ids; std::vector<float> model, inv;
        for (auto& p : objects_) {
            ids.push_back(p.second->id());
            const mat4 m = p.second->getModelMatrix(), im = inverse(m);
            model.insert(model.end(), &m.c[0][0], &m.c[0][0] + 16); inv.insert(inv.end(), &im.c[0][0], &im.c[0][0] + 16);
        }
        const bool same = !force && dev_.lastCommitter == this && lastEpoch_ == dev_.uploadEpoch && ids == lastIds_ &&
                          model.size() == lastModel_.size() &&
                          (model.empty() || memcmp(model.data(), lastModel_.data(), model.size() * sizeof(float)) == 0);
        if (same) return;
        dev_.check(rt_scene_commit(dev_.ctx(), ids.data(), model.data(), inv.data(), (int)ids.size()));
        dev_.sceneCommits++;
        dev_.lastCommitter = this; lastEpoch_ = dev_.uploadEpoch; lastIds_ = std::move(ids); lastModel_ = std::move(model);
    }
private:
    Device& dev_;
    std::vector<rt_id> lastIds_; std::vector<float> lastModel_; unsigned long long lastEpoch_ = ~0ull;
    std::vector<std::pair<std::string, std::shared_ptr<DisplayObject>>> objects_;
};

// ---------------------------------------------------------------- Camera (Camera.hpp:16-132)
class Camera {
public:
    Camera(Device& dev, vec3 position, quat rotation, float fov, float exposure) : dev_(dev), Position(position), Rotation(rotation), verticalFOV(fov), exposureValue(exposure) {}
    void update() { if ((int)preTransformRays.size() != width * height * 4) calcRays(); }
    // Camera::calcRays (Camera.hpp:46-70) without the upload
    static void computeRays(int width, int height, float verticalFOV, std::vector<float>& out)
    {
        const float rad = (verticalFOV / 2) * 0.01745329251994329576923690768489f;
        const float virtualHeight = (float)std::tan((double)rad) * 2; // pinned: tangent in binary64, rounded once
        const float virtualWidth = virtualHeight * width / height;
        const float wStep = virtualWidth / width, hStep = virtualHeight / height;
        const float xOffset = wStep * (width / 2), yOffset = hStep * (height / 2);
        out.assign((size_t)width * height * 4, 0.0f);
        for (int j = 0; j < height; j++) for (int i = 0; i < width; i++) {
            const float r = (float)((i * wStep) - xOffset), u = (float)((j * hStep) - yOffset);
            const float tx = r * r, ty = u * u, tz = -1.0f * -1.0f, tw = 0.0f * 0.0f;
            const float inv = 1.0f / std::sqrt((tx + ty) + (tz + tw)); // glm::normalize(vec4)
            float* o = &out[((size_t)j * width + i) * 4];
            o[0] = r * inv; o[1] = u * inv; o[2] = -1.0f * inv; o[3] = 0.0f * inv;
        }
    }
    void calcRays()
    {
        computeRays(width, height, verticalFOV, preTransformRays);
        dev_.check(rt_camera_set_rays(dev_.ctx(), preTransformRays.data(), width, height));
    }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    float getExposure() const { return exposureValue; }
    vec3 getPosition() const { return Position; }
    quat getRotation() const { return Rotation; }
    void setPosition(vec3 p) { Position = p; }
    void setRotation(quat q) { Rotation = q; }
    void setHeight(int h) { height = h; }
    void setWidth(int w) { width = w; }
    const std::vector<float>& rays() const { return preTransformRays; }
    int getPixelCount() const { return width * height; }
    // Camera::getRayDirection (Camera.hpp:105): Rotation * preTransformRays[y * width + x], glm's quat * vec3
    // (detail/type_quat.inl:343-350): v + ((cross(q.xyz, v) * q.w) + cross(q.xyz, cross(q.xyz, v))) * 2
    vec3 getRayDirection(int x, int y) const
    {
        const float* r = &preTransformRays[((size_t)y * width + x) * 4];
        const vec3 v{ r[0], r[1], r[2] }, q{ Rotation.x, Rotation.y, Rotation.z };
        auto cross = [](vec3 a, vec3 b) { return vec3{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; };
        const vec3 uv = cross(q, v), uuv = cross(q, uv);
        return { v.x + ((uv.x * Rotation.w) + uuv.x) * 2.0f, v.y + ((uv.y * Rotation.w) + uuv.y) * 2.0f, v.z + ((uv.z * Rotation.w) + uuv.z) * 2.0f };
    }
private:
    Device& dev_;
    vec3 Position; quat Rotation;
    std::vector<float> preTransformRays;
    int width = 0, height = 0;
    float verticalFOV, exposureValue;
};

// ---------------------------------------------------------------- output: headless framebuffer (replaces RenderWindow.cpp:215-234, 389-421)
class FrameBuffer {
public:
    FrameBuffer(Device& dev, int w, int h) : dev_(dev), w_(w), h_(h) { dev_.check(rt_framebuffer_create(dev_.ctx(), w, h, &id_)); }
    ~FrameBuffer() { if (id_ >= 0) rt_framebuffer_destroy(dev_.ctx(), id_); }
    FrameBuffer(const FrameBuffer&) = delete;
    rt_id id() const { return id_; }
    int width() const { return w_; }
    int height() const { return h_; }
    std::vector<float> read() const { std::vector<float> v((size_t)w_ * h_ * 4); dev_.check(rt_framebuffer_read(dev_.ctx(), id_, v.data())); return v; }
    // the texels RenderWindow uploads to its RGBA32F texture (RenderWindow.cpp:172-176, 282-283): pass data() to glTexSubImage2D
    std::vector<float> present() const { std::vector<float> v((size_t)w_ * h_ * 4); dev_.check(rt_present(dev_.ctx(), id_, RT_PRESENT_RGBA32F, 0, v.data())); return v; }
    // takeScreenshot (RenderWindow.cpp:389-421): the reference reads the sRGB-encoded back buffer and flips it; row 0 of
    // the render buffer is the bottom of the picture.  Converted and flipped on the GPU (rt_present).
    std::vector<unsigned char> screenshot() const { std::vector<unsigned char> px((size_t)w_ * h_ * 4); dev_.check(rt_present(dev_.ctx(), id_, RT_PRESENT_SRGB8, 1, px.data())); return px; }
    bool savePNG(const std::string& path) const { const std::vector<unsigned char> px = screenshot(); return png::encode(path, px.data(), (unsigned)w_, (unsigned)h_); }
private:
    Device& dev_;
    int w_, h_;
    rt_id id_ = -1;
};

// ---------------------------------------------------------------- Raytracer / GPURaytracer (Raytracer.hpp:18-63, GPURaytracer.cu:24-74)
class Raytracer {
public:
    void setBounceCount(int count) { bounceCount = count; }
    void setMaxDistance(float distance) { maxDistance = distance; }
    void setAOIntensity(float intensity) { aoIntensity = intensity; }
    void setProgressiveRendering(bool progressive) { progressiveRendering = progressive; }
    int getBounceCount() const { return bounceCount; }
    float getMaxDistance() const { return maxDistance; }
    float getAOIntensity() const { return aoIntensity; }
    bool getProgressiveRendering() const { return progressiveRendering; }
    int getSampleCount() const { return (int)sampleCount; }
    void setSampleCount(int count) { sampleCount = (unsigned)count; }
    void setAntiAliasingEnabled(bool enabled) { antiAliasingEnabled = enabled; }
    bool getAntiAliasingEnabled() const { return antiAliasingEnabled; }
    void setSkyLight(float pitch, float yaw, vec4 light, vec4 sky)
    {
        lightColor = light; skyColor = sky;
        const float cp = (float)std::cos((double)pitch), sp = (float)std::sin((double)pitch), cy = (float)std::cos((double)yaw), sy = (float)std::sin((double)yaw);
        lightDirection = { cp * cy, sp, cp * sy, 0.0f };
    }
    void setFrameCount(unsigned f) { frameCount = f; }
protected:
    int bounceCount = 3; float maxDistance = 100.0f; unsigned frameCount = 0, progressiveFrameCount = 0;
    float aoIntensity = 0.01f; bool progressiveRendering = false, antiAliasingEnabled = true; unsigned sampleCount = 1;
    vec4 lightDirection{ 1, 0, 0, 0 }, lightColor{ 1, 1, 1, 100 }, skyColor{ 0.5f, 0.5f, 1.0f, 0.3f }; // GUI defaults, RenderWindow.cpp:332-335
};

class GPURaytracer : public Raytracer {
public:
    explicit GPURaytracer(Device& dev) : dev_(dev) {}
    // the reference rebuilds its instance table every frame (Scene.cpp:3-29); pass recommit = false for a static scene
    void raytrace(Scene& s, Camera& c, FrameBuffer& o, bool recommit = true)
    {
        if (recommit) s.commit();
        c.update();
        rt_camera_params cam; const vec3 p = c.getPosition(); const quat q = c.getRotation();
        cam.origin[0] = p.x; cam.origin[1] = p.y; cam.origin[2] = p.z; cam.origin[3] = 1.0f;
        cam.rotation[0] = q.x; cam.rotation[1] = q.y; cam.rotation[2] = q.z; cam.rotation[3] = q.w;
        cam.exposure = c.getExposure(); cam.width = c.getWidth(); cam.height = c.getHeight();
        rt_renderer_params r{ bounceCount, maxDistance, aoIntensity, frameCount, progressiveFrameCount, antiAliasingEnabled ? 1 : 0, sampleCount };
        rt_sky_params sky;
        const vec4* src[3] = { &lightDirection, &lightColor, &skyColor }; float* dst[3] = { sky.direction, sky.lightColor, sky.skyColor };
        for (int i = 0; i < 3; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; dst[i][3] = src[i]->w; }
        dev_.check(rt_render(dev_.ctx(), &cam, &r, &sky, rt_rect{ 0, 0, c.getWidth(), c.getHeight() }, o.id()));
        dev_.check(rt_sync(dev_.ctx()));
        frameCount++;
        if (progressiveRendering) progressiveFrameCount++; else progressiveFrameCount = 0;
    }
private:
    Device& dev_;
};

// ---------------------------------------------------------------- CPURaytracer (processor/CPU/CPURaytracer.hpp / .cpp), SURVEY 8f-4
// The reference's legacy renderer behind its "Use GPU" toggle: single-threaded, diffuse only (no sky, no BRDF), nearest
// texel, ONE random seed walked through the whole frame in pixel order (CPURaytracer.hpp:44), only the first mesh of an
// object (DisplayObject.hpp:49), an object pre-cull against the box spanned by the model-transformed min and max corners
// (:52-56; not an axis-aligned bound once the object is rotated -- reproduced).  Host code: nothing here touches the GPU.
// The transcendentals are the C library's float functions (logf, cosf, expf), as the reference's <cmath> calls resolve; the
// reference ran MSVC's, so the last bit of a bounce direction is not pinned by anything (parity unpinned).
class CPURaytracer : public Raytracer {
public:
    // CPURaytracer::raytrace (:6-40): W*H*4 floats, alpha 1
    std::vector<float> raytrace(Scene& scene, Camera& camera)
    {
        camera.update();
        auto& objects = scene.getObjects();
        std::vector<float> output((size_t)camera.getPixelCount() * 4, 1.0f);
        const vec3 origin = camera.getPosition();
        const float exposure = camera.getExposure();
        for (int y = 0; y < camera.getHeight(); y++)
            for (int x = 0; x < camera.getWidth(); x++) {
                CRay ray{};
                ray.origin = origin; ray.direction = camera.getRayDirection(x, y); ray.bounceCount = bounceCount;
                const vec3 c = singleTrace(ray, objects);
                const size_t i = (size_t)y * camera.getWidth() + x;
                output[4 * i] = 1.0f - std::exp(-c.x * exposure);   // glm::exp(vec3) = std::exp per float component
                output[4 * i + 1] = 1.0f - std::exp(-c.y * exposure);
                output[4 * i + 2] = 1.0f - std::exp(-c.z * exposure);
            }
        return output;
    }
    void setSeed(unsigned s) { randomSeed = s; }
private:
    struct CHit { vec3 hitPosition; float distance; float u, v; Material* material; vec3 t, b, n; vec3 normal; };
    struct CRay { vec3 origin, direction; int bounceCount; bool didHit; CHit hitInfo; };
    using Objects = std::vector<std::pair<std::string, std::shared_ptr<DisplayObject>>>;
    static vec3 mulPoint(const mat4& m, vec3 v, float w) // glm mat4 * vec4 (detail/type_mat4x4.inl): the mul/add order of its four column products
    {
        const float m0x = m.c[0][0] * v.x, m0y = m.c[0][1] * v.x, m0z = m.c[0][2] * v.x;
        const float m1x = m.c[1][0] * v.y, m1y = m.c[1][1] * v.y, m1z = m.c[1][2] * v.y;
        const float m2x = m.c[2][0] * v.z, m2y = m.c[2][1] * v.z, m2z = m.c[2][2] * v.z;
        const float m3x = m.c[3][0] * w, m3y = m.c[3][1] * w, m3z = m.c[3][2] * w;
        return { (m0x + m1x) + (m2x + m3x), (m0y + m1y) + (m2y + m3y), (m0z + m1z) + (m2z + m3z) };
    }
    static vec3 normalize3(vec3 v) { const float s = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return { v.x * s, v.y * s, v.z * s }; } // v * inversesqrt(dot(v, v))
    static vec3 cross3(vec3 a, vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
    static float comp(vec3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

    vec3 singleTrace(CRay& ray, const Objects& objects) // :42-84
    {
        vec3 incoming{ 0, 0, 0 }, rayColor{ 1, 1, 1 };
        for (int i = 0; i < bounceCount; i++) {
            for (auto& op : objects) {
                DisplayObject* object = op.second.get();
                const mat4 model = object->getModelMatrix();
                const float* mn = object->getMesh()->getMinBound(); const float* mx = object->getMesh()->getMaxBound();
                if (!intersectsBoundingBox(ray, mulPoint(model, { mn[0], mn[1], mn[2] }, 1.0f), mulPoint(model, { mx[0], mx[1], mx[2] }, 1.0f))) continue;
                getIntersectionPoint(ray, object, model);
            }
            if (!ray.didHit) break;
            Material* m = ray.hitInfo.material;
            const float u = ray.hitInfo.u, v = ray.hitInfo.v;
            const vec3 nl = m->getNormal(u, v); // setMaterialData (:199-207): normal = tbn * map normal
            ray.hitInfo.normal = { ray.hitInfo.t.x * nl.x + ray.hitInfo.b.x * nl.y + ray.hitInfo.n.x * nl.z,
                                   ray.hitInfo.t.y * nl.x + ray.hitInfo.b.y * nl.y + ray.hitInfo.n.y * nl.z,
                                   ray.hitInfo.t.z * nl.x + ray.hitInfo.b.z * nl.y + ray.hitInfo.n.z * nl.z };
            const vec4 al = m->getAlbedo(u, v);
            if (i == 0) { const vec3 ao = m->getAmbientOcclusion(u, v); incoming.x += ao.x * al.x * aoIntensity; incoming.y += ao.y * al.y * aoIntensity; incoming.z += ao.z * al.z * aoIntensity; }
            const vec3 em = m->getEmissionColor(u, v); const float es = m->getEmissionStrength();
            incoming.x += rayColor.x * em.x * es; incoming.y += rayColor.y * em.y * es; incoming.z += rayColor.z * em.z * es;
            rayColor.x *= al.x; rayColor.y *= al.y; rayColor.z *= al.z;
            if (i < ray.bounceCount - 1) {
                ray.origin = ray.hitInfo.hitPosition;
                ray.direction = randomUnitVectorInHemisphere(randomSeed, ray.hitInfo.normal);
                ray.didHit = false;
            }
        }
        return incoming;
    }
    bool intersectsBoundingBox(const CRay& ray, vec3 minBound, vec3 maxBound) const // :86-105
    {
        float tmin = 0.0f, tmax = ray.didHit ? ray.hitInfo.distance : INFINITY;
        for (int d = 0; d < 3; d++) {
            const float dir = comp(ray.direction, d), inv = 1.0f / dir, o = comp(ray.origin, d);
            const bool sign = std::signbit(dir);
            const float bmin = sign ? comp(maxBound, d) : comp(minBound, d), bmax = sign ? comp(minBound, d) : comp(maxBound, d);
            const float dmin = (bmin - o) * inv, dmax = (bmax - o) * inv;
            tmin = dmin > tmin ? dmin : tmin;   // the class's own max / min (CPURaytracer.hpp:63-64)
            tmax = dmax < tmax ? dmax : tmax;
        }
        return tmin < tmax;
    }
    void getIntersectionPoint(CRay& ray, DisplayObject* object, const mat4& model) // :107-197
    {
        Mesh* mesh = object->getMesh();
        const auto& vertices = mesh->getVertices(); const auto& indices = mesh->getIndices(); const auto& uvs = mesh->getUVCoords();
        int closest = -1; vec3 bary{};
        float minDistance = ray.didHit ? ray.hitInfo.distance : INFINITY;
        auto vert = [&](int i) { return vec3{ vertices[(size_t)i * 3], vertices[(size_t)i * 3 + 1], vertices[(size_t)i * 3 + 2] }; };
        for (int i = 0; i < mesh->getTriangleCount(); i++) {
            const vec3 v0 = mulPoint(model, vert(indices[3 * i]), 1.0f), v1 = mulPoint(model, vert(indices[3 * i + 1]), 1.0f), v2 = mulPoint(model, vert(indices[3 * i + 2]), 1.0f);
            vec3 bc{};
            const float dist = distToTriangle(ray, v0, v1, v2, bc);
            if (dist < minDistance) { minDistance = dist; closest = i; bary = bc; }
        }
        if (closest == -1) return;
        const int i0 = indices[3 * closest], i1 = indices[3 * closest + 1], i2 = indices[3 * closest + 2];
        const vec3 v0 = vert(i0), v1 = vert(i1), v2 = vert(i2);
        auto uv = [&](int i, int c) { return uvs.empty() ? 0.0f : uvs[(size_t)i * 2 + c]; };
        const float du1 = uv(i1, 0) - uv(i0, 0), dv1 = uv(i1, 1) - uv(i0, 1), du2 = uv(i2, 0) - uv(i0, 0), dv2 = uv(i2, 1) - uv(i0, 1);
        const vec3 e1{ v1.x - v0.x, v1.y - v0.y, v1.z - v0.z }, e2{ v2.x - v0.x, v2.y - v0.y, v2.z - v0.z };
        const float f = 1.0f / (du1 * dv2 - du2 * dv1);
        const vec3 tg{ f * (dv2 * e1.x - dv1 * e2.x), f * (dv2 * e1.y - dv1 * e2.y), f * (dv2 * e1.z - dv1 * e2.z) };
        const vec3 bt{ f * (-du2 * e1.x + du1 * e2.x), f * (-du2 * e1.y + du1 * e2.y), f * (-du2 * e1.z + du1 * e2.z) };
        const vec3 T = normalize3(mulPoint(model, tg, 0.0f)), B = normalize3(mulPoint(model, bt, 0.0f));
        const vec3 N = cross3(T, B);
        vec3 ip{ (bary.x * v0.x + bary.y * v1.x) + bary.z * v2.x, (bary.x * v0.y + bary.y * v1.y) + bary.z * v2.y, (bary.x * v0.z + bary.y * v1.z) + bary.z * v2.z };
        ray.didHit = true;
        ray.hitInfo.hitPosition = mulPoint(model, ip, 1.0f);
        ray.hitInfo.distance = minDistance;
        ray.hitInfo.u = (bary.x * uv(i0, 0) + bary.y * uv(i1, 0)) + bary.z * uv(i2, 0);
        ray.hitInfo.v = (bary.x * uv(i0, 1) + bary.y * uv(i1, 1)) + bary.z * uv(i2, 1);
        ray.hitInfo.material = object->getMaterial();
        ray.hitInfo.t = T; ray.hitInfo.b = B; ray.hitInfo.n = N;
    }
    float distToTriangle(const CRay& ray, vec3 v0, vec3 v1, vec3 v2, vec3& bc) const // :211-296; INFINITY = no hit
    {
        vec3 a{ v0.x - ray.origin.x, v0.y - ray.origin.y, v0.z - ray.origin.z }, b{ v1.x - ray.origin.x, v1.y - ray.origin.y, v1.z - ray.origin.z },
            c{ v2.x - ray.origin.x, v2.y - ray.origin.y, v2.z - ray.origin.z };
        const float ax = std::fabs(ray.direction.x), ay = std::fabs(ray.direction.y), az = std::fabs(ray.direction.z);
        const int kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2), kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
        const float dx = comp(ray.direction, kx), dy = comp(ray.direction, ky), dz = comp(ray.direction, kz);
        float p0x = comp(a, kx), p0y = comp(a, ky), p0z = comp(a, kz), p1x = comp(b, kx), p1y = comp(b, ky), p1z = comp(b, kz),
              p2x = comp(c, kx), p2y = comp(c, ky), p2z = comp(c, kz);
        const float sz = 1.f / dz, sx = -dx * sz, sy = -dy * sz;
        p0x += sx * p0z; p0y += sy * p0z; p1x += sx * p1z; p1y += sy * p1z; p2x += sx * p2z; p2y += sy * p2z;
        float e0 = p1x * p2y - p1y * p2x, e1 = p2x * p0y - p2y * p0x, e2 = p0x * p1y - p0y * p1x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            e0 = (float)((double)p2y * (double)p1x - (double)p2x * (double)p1y);
            e1 = (float)((double)p0y * (double)p2x - (double)p0x * (double)p2y);
            e2 = (float)((double)p1y * (double)p0x - (double)p1x * (double)p0y);
        }
        if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return INFINITY;
        const float det = e0 + e1 + e2;
        if (det == 0) return INFINITY;
        p0z *= sz; p1z *= sz; p2z *= sz;
        const float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
        if (det < 0 && (tScaled >= 0 || tScaled < maxDistance * det)) return INFINITY;
        else if (det > 0 && (tScaled <= 0 || tScaled > maxDistance * det)) return INFINITY;
        const float invDet = 1 / det;
        bc = { e0 * invDet, e1 * invDet, e2 * invDet };
        return tScaled * invDet;
    }
    static float randomValue(unsigned& seed) // :298-304
    {
        seed = seed * 747796405u + 2891336453u;
        unsigned r = ((seed >> ((seed >> 28) + 4)) ^ seed) * 277803737u;
        r = (r >> 22) ^ r;
        return (float)(r / 4294967295.0);
    }
    static float randomNormal(unsigned& seed) // :306-312: theta through a binary64 product; log / sqrt / cos are the float overloads of <cmath>
    {
        const float theta = (float)(2 * 3.1415926 * randomValue(seed));
        const float r = randomValue(seed);
        const float rho = std::sqrt(-2 * std::log(r));
        return rho * std::cos(theta);
    }
    static vec3 randomUnitVectorInHemisphere(unsigned& seed, vec3 n) // :315-332
    {
        const float x = randomNormal(seed), y = randomNormal(seed), z = randomNormal(seed);
        const vec3 u = normalize3({ x, y, z });
        return (u.x * n.x + u.y * n.y + u.z * n.z) > 0.0f ? u : vec3{ -u.x, -u.y, -u.z };
    }
    unsigned randomSeed = 0;
};

// ---------------------------------------------------------------- one host thread driving N contexts (SURVEY 8b "Threading", 8e)
// The image-tile split of one frame in C++: every context (one per GPU; several on one GPU work too) holds its own
// copy of the scene, rank r renders the interleaved row bands r, r + N, ... of the frame with ONE asynchronous
// rt_render_bands call on its own stream, the thread then waits for all of them and composites on the host.
// (Between GPUs of a node the composite is an RCCL all_gather instead: raytracer_amd/multigpu.py, bench.py.)
class MultiGPURaytracer : public Raytracer {
public:
    struct Rank { Device* dev; Scene* scene; Camera* camera; FrameBuffer* fb; };
    explicit MultiGPURaytracer(std::vector<Rank> ranks, int bandRows = 16) : ranks_(std::move(ranks)), bandRows_(bandRows) {}
    // out: W*H*4 floats, row 0 = bottom, as rt_framebuffer_read gives them
    void raytrace(std::vector<float>& out, bool recommit = true)
    {
        const int n = (int)ranks_.size();
        const int W = ranks_[0].camera->getWidth(), H = ranks_[0].camera->getHeight();
        for (int r = 0; r < n; r++) { // enqueue everywhere first: the contexts' streams run side by side
            Rank& k = ranks_[r];
            if (recommit) k.scene->commit();
            k.camera->update();
            rt_camera_params cam; const vec3 p = k.camera->getPosition(); const quat q = k.camera->getRotation();
            cam.origin[0] = p.x; cam.origin[1] = p.y; cam.origin[2] = p.z; cam.origin[3] = 1.0f;
            cam.rotation[0] = q.x; cam.rotation[1] = q.y; cam.rotation[2] = q.z; cam.rotation[3] = q.w;
            cam.exposure = k.camera->getExposure(); cam.width = W; cam.height = H;
            rt_renderer_params rp{ bounceCount, maxDistance, aoIntensity, frameCount, progressiveFrameCount, antiAliasingEnabled ? 1 : 0, sampleCount };
            rt_sky_params sky;
            const vec4* src[3] = { &lightDirection, &lightColor, &skyColor }; float* dst[3] = { sky.direction, sky.lightColor, sky.skyColor };
            for (int i = 0; i < 3; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; dst[i][3] = src[i]->w; }
            k.dev->check(rt_render_bands(k.dev->ctx(), &cam, &rp, &sky, rt_rect{ 0, 0, W, H }, bandRows_, r, n, k.fb->id()));
        }
        out.assign((size_t)W * H * 4, 0.0f);
        for (int r = 0; r < n; r++) { // wait + composite: band b belongs to rank b % n
            const std::vector<float> part = ranks_[r].fb->read();
            for (int y0 = r * bandRows_; y0 < H; y0 += n * bandRows_) {
                const int y1 = std::min(H, y0 + bandRows_);
                memcpy(&out[(size_t)y0 * W * 4], &part[(size_t)y0 * W * 4], (size_t)(y1 - y0) * W * 16);
            }
        }
        frameCount++;
        if (progressiveRendering) progressiveFrameCount++; else progressiveFrameCount = 0;
    }
private:
    std::vector<Rank> ranks_;
    int bandRows_;
};

} // namespace rth
