// rt_headless.cpp -- example and test driver for the C++ host mirror (rt_host.hpp).
//   rt_headless math                          CPU only: prints the bits of the GLM restatements
//   rt_headless render OUT.png OUT.f32 W H SPP DEPTH   builds BASELINE config 1 through the class API
//                                             (Scene/Cube/Sphere/Material/Camera/GPURaytracer), renders it on
//                                             GPU 0, writes a PNG and the raw float4 framebuffer
#include "rt_host.hpp"

#include <cstdlib>

using namespace rth;

static void dump(const char* name, const float* f, size_t n)
{
    printf("%s", name);
    for (size_t i = 0; i < n; i++) { uint32_t u; memcpy(&u, &f[i], 4); printf(" %08x", u); }
    printf("\n");
}

int main(int argc, char** argv)
{
    if (argc >= 2 && std::string(argv[1]) == "math") {
        const vec3 pos{ 1.5f, -2.0f, 0.25f }, scl{ 2.0f, 0.5f, 1.25f };
        const quat q{ 0.0f, 0.38268343f, 0.0f, 0.92387953f };
        mat4 m = identity(); m = translate(m, pos); m = mul(m, mat4_cast(q)); m = scale(m, scl);
        const mat4 inv = inverse(m);
        dump("model", &m.c[0][0], 16);
        dump("inverse", &inv.c[0][0], 16);
        std::vector<float> rays;
        Camera::computeRays(7, 5, 45.0f, rays);
        dump("rays", rays.data(), rays.size());
        return 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "asset") {
        // flatten a model file and print what came out: per mesh the sizes and a CRC of every array
        // (tests/test_host_cpp.py compares them with raytracer_amd/assets.py); no device involved
        assets::FlatAsset a;
        if (!assets::loadAsset(argv[2], a)) { printf("load failed\n"); return 3; }
        printf("meshes %zu materials %zu triangles %zu texdir %s\n", a.meshes.size(), a.materials.size(), a.triangleCount(), a.textureDirectory().c_str());
        auto crc = [](const void* p, size_t n) { return (unsigned long)crc32(0L, (const Bytef*)p, (uInt)n); };
        for (auto& m : a.meshes)
            printf("mesh %zu %zu %d %lu %lu %lu %lu\n", m.positions.size() / 3, m.indices.size() / 3, m.material, crc(m.positions.data(), m.positions.size() * 4),
                   crc(m.indices.data(), m.indices.size() * 4), crc(m.uvs.data(), m.uvs.size() * 4), crc(m.normals.data(), m.normals.size() * 4));
        for (auto& m : a.materials) {
            printf("material %s", m.name.c_str());
            for (auto& t : m.textures) printf(" %s=%s", t.first.c_str(), t.second.c_str());
            printf("\n");
        }
        return 0;
    }
    if (argc >= 8 && std::string(argv[1]) == "render") {
        try {
            const int W = atoi(argv[4]), H = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
            Device dev(0);
            // RenderWindow's defaults (RenderWindow.cpp:152,167-168): Default material, camera at (0,0,10), FOV 45
            auto def = std::make_shared<Material>(dev, "Default", vec4{ 0.5f, 0.5f, 0.5f, 1.0f }, 0.5f, 0.0f);
            def->sendToGPU();
            Scene scene(dev);
            auto cube = std::make_shared<Cube>(dev);
            cube->setMaterial(def);
            auto sphere = std::make_shared<Sphere>(dev, vec3{ 1.5f, 0.0f, 0.0f });
            sphere->setMaterial(def);
            if (!cube->getLoadSuccess() || !sphere->getLoadSuccess()) return 3;
            cube->sendToGPU(); sphere->sendToGPU();
            scene.addToScene("cube", cube);
            scene.addToScene("sphere", sphere);
            if (scene.addToScene("cube", sphere)) return 4; // duplicate names are refused (Scene.hpp:45-53)
            Camera cam(dev, vec3{ 0.6f, 0.3f, 4.0f }, quat{}, 45.0f, 1.0f);
            cam.setWidth(W); cam.setHeight(H);
            GPURaytracer rt(dev);
            rt.setBounceCount(depth); rt.setSampleCount(spp); rt.setFrameCount(3);
            rt.setSkyLight(0.0f, 0.0f, vec4{ 1, 1, 1, 100 }, vec4{ 0.5f, 0.5f, 1.0f, 0.3f });
            FrameBuffer fb(dev, W, H);
            rt.raytrace(scene, cam, fb);
            if (!fb.savePNG(argv[2])) return 5;
            const std::vector<float> px = fb.read();
            FILE* f = fopen(argv[3], "wb");
            if (!f || fwrite(px.data(), 4, px.size(), f) != px.size()) return 6;
            fclose(f);
            // PNG round trip through the host mirror's own decoder + the Texture path
            std::vector<unsigned char> back; int w2, h2, n2;
            if (!png::decode(argv[2], back, w2, h2, n2) || w2 != W || h2 != H || n2 != 4) return 7;
            Texture t(dev, argv[2]);
            if (!t.getLoadSuccess() || t.getWidth() != W) return 8;
            Material bad(dev, "x");
            if (bad.setAlbedoTexture("/nonexistent.png")) return 9; // failed loads report false (Material.cu:113-130)
            printf("ok %dx%d\n", W, H);
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 2; }
        return 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "decode") {
        // the host mirror's PNG / JPEG decoders on a file: extents, channels and a CRC of the decoded bytes (CPU only)
        std::vector<unsigned char> pix; int w = 0, h = 0, n = 0;
        if (!png::decode(argv[2], pix, w, h, n) && !jpeg::decode(argv[2], pix, w, h, n)) { printf("refused\n"); return 0; }
        printf("%d %d %d %lu\n", w, h, n, (unsigned long)crc32(0L, pix.data(), (uInt)pix.size()));
        return 0;
    }
    auto build_c1 = [](Device& dev, Scene& scene, std::shared_ptr<Material>& def) {
        def = std::make_shared<Material>(dev, "Default", vec4{ 0.5f, 0.5f, 0.5f, 1.0f }, 0.5f, 0.0f);
        def->sendToGPU();
        auto cube = std::make_shared<Cube>(dev);
        cube->setMaterial(def);
        auto sphere = std::make_shared<Sphere>(dev, vec3{ 1.5f, 0.0f, 0.0f });
        sphere->setMaterial(def);
        cube->sendToGPU(); sphere->sendToGPU();
        scene.addToScene("cube", cube);
        scene.addToScene("sphere", sphere);
    };
    if (argc >= 6 && std::string(argv[1]) == "frames") {
        // GPURaytracer::raytrace's counter rule (GPURaytracer.cu:69-73) over seven frames -- progressive rendering on for
        // three, off for two, on for two -- each frame appended to OUT.f32; then path-keyed sharing of Textures and CustomModels
        try {
            const int W = atoi(argv[3]), H = atoi(argv[4]);
            Device dev(0);
            Scene scene(dev); std::shared_ptr<Material> def;
            build_c1(dev, scene, def);
            Camera cam(dev, vec3{ 0.6f, 0.3f, 4.0f }, quat{}, 45.0f, 1.0f);
            cam.setWidth(W); cam.setHeight(H);
            GPURaytracer rt(dev);
            rt.setBounceCount(3); rt.setSampleCount(2);
            FrameBuffer fb(dev, W, H);
            FILE* f = fopen(argv[2], "wb");
            if (!f) return 6;
            const bool plan[7] = { true, true, true, false, false, true, true };
            for (bool prog : plan) {
                rt.setProgressiveRendering(prog);
                rt.raytrace(scene, cam, fb);
                const std::vector<float> px = fb.read();
                if (fwrite(px.data(), 4, px.size(), f) != px.size()) return 6;
            }
            fclose(f);
            // the instance table is committed when something changed, not per frame: seven frames of a static scene are one
            // rt_scene_commit; a moved object is one more, and a refit of the tree (rt_scene_commit_kind 2); a mesh created
            // on the context in between (another object's) makes the next commit a full one again
            if (dev.sceneCommits != 1) return 14;
            scene.getObjects()[1].second->setPosition(vec3{ 1.4f, 0.1f, 0.0f });
            rt.raytrace(scene, cam, fb);
            if (dev.sceneCommits != 2 || rt_scene_commit_kind(dev.ctx()) != 2) return 15;
            rt.raytrace(scene, cam, fb);
            if (dev.sceneCommits != 2) return 16;
            {
                auto extra = std::make_shared<Cube>(dev);
                extra->setMaterial(def); extra->sendToGPU();
                rt.raytrace(scene, cam, fb);
                if (dev.sceneCommits != 3 || rt_scene_commit_kind(dev.ctx()) != 1) return 17;
            }
            // sharing: two Textures of one path upload once; two CustomModels of one path parse and upload once
            fb.savePNG(argv[5]);
            {
                Texture a(dev, argv[5]), b(dev, argv[5]);
                if (!a.getLoadSuccess() || a.id() != b.id() || dev.textureUploads != 1) return 10;
            }
            { Texture c(dev, argv[5]); if (dev.textureUploads != 2) return 11; } // the last owner gone: loaded again
            if (argc >= 7) {
                CustomModel m1(dev, argv[6]), m2(dev, argv[6], vec3{ 1, 0, 0 });
                if (!m1.getLoadSuccess() || !m2.getLoadSuccess() || dev.modelParses != 1) return 12;
                if (m1.getMeshes()[0].first->id() != m2.getMeshes()[0].first->id()) return 13;
            }
            printf("ok frames\n");
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 2; }
        return 0;
    }
    if (argc >= 5 && std::string(argv[1]) == "legacy") {
        // the reference's legacy CPURaytracer (its "Use GPU" toggle off) on a small textured scene: cube with an albedo map,
        // rotated and scaled sphere, emissive cube; rendered on the host, written as float4
        try {
            const int W = atoi(argv[3]), H = atoi(argv[4]);
            Device dev(0);
            std::vector<unsigned char> tex(16 * 16 * 3);
            for (int i = 0; i < 16 * 16; i++) { tex[i * 3] = (unsigned char)(i * 7 + 30); tex[i * 3 + 1] = (unsigned char)(255 - i); tex[i * 3 + 2] = (unsigned char)((i * 13) & 255); }
            auto a = std::make_shared<Material>(dev, "a", vec4{ 0.8f, 0.3f, 0.2f, 1.0f }, 0.4f, 0.1f);
            a->setAlbedoTexture(std::make_shared<Texture>(dev, tex.data(), 16, 16, 3));
            auto b = std::make_shared<Material>(dev, "b", vec4{ 0.2f, 0.6f, 0.9f, 1.0f }, 0.7f, 0.0f, vec3{ 0.05f, 0.1f, 0.02f }, 1.0f);
            auto e = std::make_shared<Material>(dev, "e", vec4{ 0.9f, 0.9f, 0.9f, 1.0f }, 0.5f, 0.0f, vec3{ 1.0f, 0.8f, 0.5f }, 2.0f);
            Scene scene(dev);
            auto cube = std::make_shared<Cube>(dev); cube->setPosition({ -0.7f, 0.0f, 0.0f }); cube->setRotation({ 0.0f, 0.1f, 0.0f, 0.995f }); cube->setScale({ 1.3f, 1.6f, 1.1f }); cube->setMaterial(a);
            auto sphere = std::make_shared<Sphere>(dev, vec3{ 0.9f, 0.1f, 0.3f }, quat{ 0.1f, 0.2f, 0.0f, 0.97f }, vec3{ 1.4f, 1.0f, 1.2f }); sphere->setMaterial(b);
            auto lamp = std::make_shared<Cube>(dev); lamp->setPosition({ 0.2f, 1.6f, 0.5f }); lamp->setScale({ 3.0f, 0.2f, 3.0f }); lamp->setMaterial(e);
            scene.addToScene("cube", cube); scene.addToScene("sphere", sphere); scene.addToScene("lamp", lamp);
            Camera cam(dev, vec3{ 0.1f, 0.4f, 5.0f }, quat{ -0.05f, 0.02f, 0.0f, 0.998f }, 45.0f, 1.2f);
            cam.setWidth(W); cam.setHeight(H);
            CPURaytracer rt;
            rt.setBounceCount(3); rt.setAOIntensity(0.05f);
            const std::vector<float> px = rt.raytrace(scene, cam);
            FILE* f = fopen(argv[2], "wb");
            if (!f || fwrite(px.data(), 4, px.size(), f) != px.size()) return 6;
            fclose(f);
            printf("ok legacy\n");
        } catch (const std::exception& ex) { fprintf(stderr, "error: %s\n", ex.what()); return 2; }
        return 0;
    }
    if (argc >= 7 && std::string(argv[1]) == "multi") {
        // one thread, N contexts (all on GPU 0 here): the interleaved bands of one frame, composited on the host
        try {
            const int W = atoi(argv[3]), H = atoi(argv[4]), spp = atoi(argv[5]), n = atoi(argv[6]);
            std::vector<std::unique_ptr<Device>> devs; std::vector<std::unique_ptr<Scene>> scenes; std::vector<std::unique_ptr<Camera>> cams;
            std::vector<std::unique_ptr<FrameBuffer>> fbs; std::vector<std::shared_ptr<Material>> mats((size_t)n);
            std::vector<MultiGPURaytracer::Rank> ranks;
            for (int r = 0; r < n; r++) {
                devs.emplace_back(new Device(0));
                scenes.emplace_back(new Scene(*devs[r]));
                build_c1(*devs[r], *scenes[r], mats[r]);
                cams.emplace_back(new Camera(*devs[r], vec3{ 0.6f, 0.3f, 4.0f }, quat{}, 45.0f, 1.0f));
                cams[r]->setWidth(W); cams[r]->setHeight(H);
                fbs.emplace_back(new FrameBuffer(*devs[r], W, H));
                ranks.push_back({ devs[r].get(), scenes[r].get(), cams[r].get(), fbs[r].get() });
            }
            MultiGPURaytracer rt(ranks, 8);
            rt.setBounceCount(4); rt.setSampleCount(spp); rt.setFrameCount(3);
            std::vector<float> out;
            rt.raytrace(out);
            FILE* f = fopen(argv[2], "wb");
            if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) return 6;
            fclose(f);
            printf("ok multi %d\n", n);
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 2; }
        return 0;
    }
    fprintf(stderr, "usage: rt_headless math | asset FILE | decode FILE.png | render OUT.png OUT.f32 W H SPP DEPTH | frames OUT.f32 W H TMP.png [MODEL] | multi OUT.f32 W H SPP N\n");
    return 1;
}
