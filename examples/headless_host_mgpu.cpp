// headless_host_mgpu.cpp — the tile split of config 5 driven by ONE C++ process over every visible GPU, through nothing
// but include/okrt.h: one okrt context per device, each owning the interleaved 8-row stripes of its rank
// (okrt_set_tile), the same scene tables uploaded to each, F frames rendered with no per-frame exchange, then ONE
// gather of the float4 accumulation tiles to the lightest rank's device over RCCL / xGMI (okrt_mgpu_gather_accum_all) and render.png from
// the gathered frame. With one visible GPU it degenerates to the single-GPU path (the gather is a device copy).
//
//   headless_host_mgpu <width> <height> <frames> <out.png> [max-gpus]
//
// The scene is a small procedural one (a displaced floor under a ring of blobs, one emitter) built with the host layer;
// a real host would upload its own tables the same way.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "okrt.h"
#include "png_writer.h"
#include "raytracer_host.h"

namespace {

// a UV sphere of radius r around c: de-indexed triangles + TriangleInfo (normal, uv, tangent, bitangent per vertex)
void addBlob(std::vector<float>& pos, std::vector<float>& info, const float c[3], float r, int seg) {
    auto vert = [&](int i, int j, float* p, float* vi) {
        const float th = 6.2831853f * (float)i / (float)(2 * seg), ph = 3.1415927f * (float)j / (float)seg;
        const float n[3] = {std::sin(ph) * std::cos(th), std::cos(ph), std::sin(ph) * std::sin(th)};
        for (int k = 0; k < 3; ++k) p[k] = c[k] + r * n[k];
        const float v[11] = {n[0], n[1], n[2], (float)i / (float)(2 * seg), (float)j / (float)seg, -std::sin(th), 0.f, std::cos(th), 0.f, 1.f, 0.f};
        for (int k = 0; k < 11; ++k) vi[k] = v[k];
    };
    for (int i = 0; i < 2 * seg; ++i)
        for (int j = 0; j < seg; ++j) {
            const int q[4][2] = {{i, j}, {i + 1, j}, {i + 1, j + 1}, {i, j + 1}};
            const int tri[2][3] = {{0, 1, 2}, {0, 2, 3}};
            for (auto& t : tri)
                for (int v = 0; v < 3; ++v) {
                    float p[3], vi[11];
                    vert(q[t[v]][0], q[t[v]][1], p, vi);
                    pos.insert(pos.end(), p, p + 3);
                    info.insert(info.end(), vi, vi + 11);
                }
        }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s width height frames out.png [max-gpus]\n", argv[0]);
        return 2;
    }
    const uint32_t W = (uint32_t)std::atoi(argv[1]), H = (uint32_t)std::atoi(argv[2]), frames = (uint32_t)std::atoi(argv[3]);
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { std::fprintf(stderr, "no GPU visible\n"); return 1; }
    const uint32_t n = (uint32_t)std::min(visible, argc > 5 ? std::max(std::atoi(argv[5]), 1) : visible);
    const uint32_t kStripeRows = 8;

    okh::ResourceManager resources;
    okh::Scene scene;
    for (int b = 0; b < 7; ++b) {
        std::vector<float> pos, info;
        const float a = 0.8975979f * (float)b, c[3] = {0.f, 0.f, 0.f};
        addBlob(pos, info, c, b == 6 ? 1.0f : 1.6f, 10);
        const uint32_t mesh = resources.addMesh(pos.data(), info.data(), (uint32_t)(pos.size() / 9));
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = b == 6 ? okh::Vec3{0.f, 5.5f, 0.f} : okh::Vec3{6.f * std::cos(a), 0.f, 6.f * std::sin(a)};
        scene.get(e).hasMesh = true;
        okh::Material& m = scene.get(e).mesh.material;
        scene.get(e).mesh.meshID = mesh;
        m.albedo[0] = 0.3f + 0.1f * (float)b; m.albedo[1] = 0.8f - 0.1f * (float)b; m.albedo[2] = 0.5f;
        m.roughness = 0.15f * (float)b;
        if (b == 6) { m.emissionPower = 6.f; m.emissionColour[0] = 1.f; m.emissionColour[1] = 0.85f; m.emissionColour[2] = 0.6f; }
    }
    {
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = {0.f, -1001.6f, 0.f};
        scene.get(e).hasSphere = true;
        scene.get(e).sphere.radius = 1000.f;
        scene.get(e).sphere.material.albedo[0] = scene.get(e).sphere.material.albedo[1] = scene.get(e).sphere.material.albedo[2] = 0.7f;
    }
    {
        const uint32_t e = scene.createEntity();
        scene.get(e).transform.position = {0.f, 4.f, -16.f};
        scene.get(e).transform.rotation = {12.f, 0.f, 0.f};
        scene.get(e).hasCamera = true;
    }

    std::vector<std::unique_ptr<okh::RayTracer>> rts;
    std::vector<okrt_ctx*> ctxs;
    for (uint32_t r = 0; r < n; ++r) {  // scene tables replicated on every GPU
        rts.emplace_back(new okh::RayTracer(okrt_get_api(), (int)r));
        okh::RayTracer& rt = *rts.back();
        if (!rt.initiate(W, H, resources)) { std::fprintf(stderr, "initiate (gpu %u): %s\n", r, rt.lastError().c_str()); return 1; }
        rt.setScene(scene);
        if (!rt.createOctTree(scene, 2, 50)) { std::fprintf(stderr, "createOctTree: %s\n", rt.lastError().c_str()); return 1; }
        if (okrt_set_tile(rt.deviceContext(), kStripeRows, r, n) != OKRT_OK) { std::fprintf(stderr, "set_tile: %s\n", okrt_last_error(rt.deviceContext())); return 1; }
        ctxs.push_back(rt.deviceContext());
    }
    std::vector<okrt_comm*> comms(n, nullptr);
    if (okrt_comm_create_all(ctxs.data(), n, comms.data()) != OKRT_OK) { std::fprintf(stderr, "comm_create_all: %s\n", okrt_comm_last_error(nullptr)); return 1; }

    // every rank traces its stripes of `frames` frames; launches are asynchronous, so the GPUs run concurrently
    for (uint32_t r = 0; r < n; ++r)
        if (!rts[r]->renderFrames(frames)) { std::fprintf(stderr, "render (gpu %u): %s\n", r, rts[r]->lastError().c_str()); return 1; }
    // the root of the gather: the rank that owns the fewest rows (the last one among equals) — it is done with its own frames
    // first, so the receives and the de-interleave start earliest there
    uint32_t root = 0, fewest = 0xFFFFFFFFu;
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t rows = 0;
        okrt_tile_rows(ctxs[r], &rows);
        if (rows <= fewest) fewest = rows, root = r;
    }
    if (okrt_mgpu_gather_accum_all(ctxs.data(), comms.data(), n, root, nullptr) != OKRT_OK) {
        std::fprintf(stderr, "gather: %s\n", okrt_last_error(ctxs[root]));
        return 1;
    }
    std::vector<float> acc((size_t)W * H * 4);
    if (okrt_mgpu_read_gathered(ctxs[root], comms[root], acc.data()) != OKRT_OK) { std::fprintf(stderr, "read_gathered: %s\n", okrt_last_error(ctxs[root])); return 1; }
    float exchangeMs = 0.f;
    okrt_comm_exchange_ms(comms[root], &exchangeMs);
    std::fprintf(stderr, "gather onto rank %u (%u rows): %.3f ms on its stream\n", root, fewest, exchangeMs);
    for (uint32_t r = 0; r < n; ++r)
        if (r != root) okrt_synchronize(ctxs[r]);

    // resolve like the shader: saturate(acc / N) -> UNORM8 (RaytracerCS.hlsl:526-536)
    std::vector<uint8_t> rgba((size_t)W * H * 4);
    double sum = 0.0;
    for (size_t i = 0; i < (size_t)W * H; ++i)
        for (int k = 0; k < 4; ++k) {
            float v = acc[4 * i + k] / (float)frames;
            v = v != v ? 0.f : std::fmin(std::fmax(v, 0.f), 1.f);
            rgba[4 * i + k] = (uint8_t)(v * 255.f + 0.5f);
            if (k < 3) sum += v;
        }
    std::string err;
    if (!okh::writePng(argv[4], W, H, rgba.data(), &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    bool alphaOk = true;
    for (size_t i = 0; i < (size_t)W * H; ++i) alphaOk = alphaOk && acc[4 * i + 3] == (float)frames;  // every pixel got every frame
    std::printf("%ux%u, %u frames on %u GPU(s), %u-row stripes, gathered to GPU 0: mean %.4f, alpha %s\n", W, H, frames, n, kStripeRows, sum / (3.0 * W * H),
                alphaOk ? "ok" : "MISMATCH");
    for (okrt_comm* c : comms) okrt_comm_destroy(c);
    return alphaOk ? 0 : 1;
}
