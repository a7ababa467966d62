// headless_host.cpp — a C++ host in the shape of the reference's Application (source/Application.cpp:28-128,661-674):
// create resources and entities, RayTracer::initiate, createOctTree, call render() once per frame, save the result.
// It links the host layer (okh::RayTracer, the mirror of source/Graphics/RayTracer.h) and the HIP device layer
// (libokrt.so) through nothing but include/okrt.h — the literal "Application and Scene stay" drop-in, without a window.
//
//   headless_host <width> <height> <frames> <out.png | out.ppm> [device-builder]
//
// Scene: the TestScene spheres of config 1 plus one small mesh (an octahedron with a checker texture), so that the
// sphere, mesh, texture and accumulation paths are all live. tests/test_gpu_cpp_host.py builds the same scene through the
// Python binding and compares the images byte for byte.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <string>

#include "okrt.h"
#include "png_writer.h"
#include "raytracer_host.h"

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s width height frames out.png|out.ppm [device-builder=0|1]\n", argv[0]);
        return 2;
    }
    const uint32_t W = (uint32_t)std::atoi(argv[1]), H = (uint32_t)std::atoi(argv[2]), frames = (uint32_t)std::atoi(argv[3]);
    const bool deviceBuilder = argc > 5 && std::atoi(argv[5]) != 0;

    okh::ResourceManager resources;
    // octahedron around the origin, radius 2: 8 triangles, de-indexed like Importer.cpp:15-56 leaves them
    const float P[6][3] = {{2, 0, 0}, {-2, 0, 0}, {0, 2, 0}, {0, -2, 0}, {0, 0, 2}, {0, 0, -2}};
    const int F[8][3] = {{0, 2, 4}, {2, 1, 4}, {1, 3, 4}, {3, 0, 4}, {2, 0, 5}, {1, 2, 5}, {3, 1, 5}, {0, 3, 5}};
    std::vector<float> pos, info;
    for (auto& f : F) {
        float n[3] = {0, 0, 0};
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) n[k] += P[f[v]][k] * (1.f / 3.f);
        for (int v = 0; v < 3; ++v) {
            for (int k = 0; k < 3; ++k) pos.push_back(P[f[v]][k]);
            const float vi[11] = {n[0], n[1], n[2], v == 1 ? 1.f : 0.f, v == 2 ? 1.f : 0.f, 1, 0, 0, 0, 1, 0};  // normal, uv, tangent, bitangent
            info.insert(info.end(), vi, vi + 11);
        }
    }
    const uint32_t mesh = resources.addMesh(pos.data(), info.data(), 8);
    std::vector<uint8_t> tex(16 * 16 * 4);
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) {
            const uint8_t c = ((x / 4 + y / 4) & 1) ? 230 : 40;
            uint8_t* t = &tex[(y * 16 + x) * 4];
            t[0] = c; t[1] = (uint8_t)(255 - c); t[2] = 128; t[3] = 255;
        }
    const uint32_t texId = resources.addTexture(tex.data(), 16, 16);

    okh::Scene scene;
    struct S { float p[3], r, albedo[3], emit[3], power; };
    const S spheres[4] = {{{-81.3f, -1109.9f, 18.0f}, 1109.8f, {1.f, 179 / 255.f, 128 / 255.f}, {1, 1, 1}, 0.f},
                          {{-4.0f, 0.4f, 20.0f}, 3.2f, {1, 1, 1}, {1, 1, 1}, 0.f},
                          {{105.5f, 71.6f, 118.0f}, 81.0f, {1, 1, 1}, {223 / 255.f, 168 / 255.f, 81 / 255.f}, 4.95f},
                          {{106.0f, 700.6f, -290.1f}, 680.0f, {0, 0, 0}, {1, 1, 1}, 0.f}};
    for (const S& s : spheres) {
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = {s.p[0], s.p[1], s.p[2]};
        scene.get(e).hasSphere = true;
        okh::SphereComponent& c = scene.get(e).sphere;
        c.radius = s.r;
        for (int k = 0; k < 3; ++k) { c.material.albedo[k] = s.albedo[k]; c.material.emissionColour[k] = s.emit[k]; }
        c.material.emissionPower = s.power;
    }
    {
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = {3.0f, 3.5f, 17.0f};
        scene.get(e).transform.rotation = {20.f, 35.f, 0.f};
        scene.get(e).hasMesh = true;
        scene.get(e).mesh.meshID = mesh;
        scene.get(e).mesh.material.albedo[0] = scene.get(e).mesh.material.albedo[1] = scene.get(e).mesh.material.albedo[2] = 1.f;
        scene.get(e).mesh.material.albedoTextureId = texId;
        scene.get(e).mesh.material.roughness = 0.4f;
    }
    {
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = {1.2f, 4.1f, 10.1f};
        scene.get(e).transform.rotation = {19.4f, -10.9f, 0.f};
        scene.get(e).hasCamera = true;
    }

    okh::RayTracer rt(okrt_get_api(), 0);
    if (!rt.initiate(W, H, resources) ) { std::fprintf(stderr, "initiate: %s\n", rt.lastError().c_str()); return 1; }
    rt.setScene(scene);
    if (!rt.createOctTree(scene, 0, 50)) { std::fprintf(stderr, "createOctTree: %s\n", rt.lastError().c_str()); return 1; }
    if (deviceBuilder && okrt_set_option(rt.deviceContext(), OKRT_OPT_BUILDER, OKRT_BUILDER_DEVICE) != OKRT_OK) return 1;
    for (uint32_t f = 0; f < frames; ++f)  // Application::run: one render() per frame (the device layer batches them)
        if (!rt.render()) { std::fprintf(stderr, "render: %s\n", rt.lastError().c_str()); return 1; }

    std::vector<uint8_t> rgba((size_t)W * H * 4);
    if (okrt_read_result_rgba8(rt.deviceContext(), rgba.data()) != OKRT_OK) {  // Application::saveScreenshot, :661-674
        std::fprintf(stderr, "read_result: %s\n", okrt_last_error(rt.deviceContext()));
        return 1;
    }
    const std::string outPath = argv[4];
    if (outPath.size() > 4 && outPath.compare(outPath.size() - 4, 4, ".png") == 0) {  // stbi_write_png("render.png", W, H, 4, data, W * 4)
        std::string err;
        if (!okh::writePng(outPath, W, H, rgba.data(), &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    } else {
        FILE* out = std::fopen(argv[4], "wb");
        if (!out) return 1;
        std::fprintf(out, "P6\n%u %u\n255\n", W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i) std::fwrite(&rgba[4 * i], 1, 3, out);
        std::fclose(out);
    }
    okrt_build_info bi;
    okrt_get_build_info(rt.deviceContext(), &bi);
    std::printf("%ux%u, %u frames (numAccumulationFrames %u), structure: %u nodes / %u triangles built by %s\n", W, H, frames,
                rt.getNumAccumulationFrames(), bi.num_nodes, bi.num_triangles, bi.builder ? "the device" : "the host");
    return 0;
}
