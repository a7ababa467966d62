// raytracer_host.cpp — see raytracer_host.h. Citations are to the reference checkout.
#include "raytracer_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <stack>
#include <thread>

namespace okh {

// ---- ResourceManager --------------------------------------------------------------------------
uint32_t ResourceManager::addMesh(const float* positions, const float* info, uint32_t numTris) {
    Mesh m;
    m.trianglesPos.resize(numTris);
    m.trianglesInfo.resize(numTris);
    if (numTris) std::memcpy(m.trianglesPos.data(), positions, (size_t)numTris * sizeof(Triangle));
    if (info && numTris) std::memcpy(m.trianglesInfo.data(), info, (size_t)numTris * sizeof(TriangleInfo));
    // Importer.cpp:52-53: the box grows over every (scaled) vertex position
    for (uint32_t t = 0; t < numTris; ++t)
        for (int k = 0; k < 3; ++k) {
            m.boundingBox.min = vmin(m.trianglesPos[t].position[k], m.boundingBox.min);
            m.boundingBox.max = vmax(m.trianglesPos[t].position[k], m.boundingBox.max);
        }
    m_meshes.push_back(std::move(m));
    return (uint32_t)m_meshes.size() - 1;
}
uint32_t ResourceManager::addTexture(const uint8_t* rgba8, uint32_t width, uint32_t height) {
    Texture t;
    t.width = width;
    t.height = height;
    t.rgba8.assign(rgba8, rgba8 + (size_t)width * height * 4);
    m_textures.push_back(std::move(t));
    return (uint32_t)m_textures.size() - 1;
}

// ---- RayTracer ----------------------------------------------------------------------------------
RayTracer::RayTracer(const okrt_api* api, int device) : m_api(api) {
    std::memset(&m_renderData, 0, sizeof(m_renderData));
    m_renderData.accumulationEnabled = 1u;  // RayTracer.h:131
    m_renderData.debugMaxCount = 500;       // RayTracer.h:155
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(m_renderData.cameraInverseProjectionMatrix, I, 64);
    std::memcpy(m_renderData.cameraInverseViewMatrix, I, 64);
    if (!api || api->abi_version != OKRT_ABI_VERSION) {
        m_error = "okrt api table missing or ABI version mismatch";
        return;
    }
    okrt_status s = api->create(device, &m_ctx);
    if (s != OKRT_OK) {
        m_error = std::string("okrt create failed (status ") + std::to_string((int)s) + "): " + api->last_error(nullptr);
        m_ctx = nullptr;
    }
}
RayTracer::~RayTracer() { shutdown(); }

void RayTracer::shutdown() {
    if (m_ctx) m_api->destroy(m_ctx);
    m_ctx = nullptr;
    m_pScene = nullptr;
    m_pResourceManager = nullptr;
}

bool RayTracer::check(okrt_status s, const char* what) {
    if (s == OKRT_OK) return true;
    m_error = std::string(what) + ": " + (m_ctx ? m_api->last_error(m_ctx) : "no context");
    return false;
}

bool RayTracer::initiate(uint32_t width, uint32_t height, const ResourceManager& resourceManager, const uint8_t* envCross,
                         uint32_t envW, uint32_t envH) {
    if (!m_ctx) return false;
    m_pResourceManager = &resourceManager;
    m_renderData.textureDims[0] = width;
    m_renderData.textureDims[1] = height;
    if (!check(m_api->resize(m_ctx, width, height), "resize")) return false;  // result + accumulation (:73)
    if (!loadTextureData()) return false;
    if (!loadEnvironmentMap(envCross, envW, envH)) return false;
    return loadMeshAndBvhData(30, 5);  // :97
}

bool RayTracer::onResize(uint32_t width, uint32_t height) {
    m_renderData.textureDims[0] = width;
    m_renderData.textureDims[1] = height;
    if (!check(m_api->resize(m_ctx, width, height), "resize")) return false;
    resetAccumulation();
    return true;
}

void RayTracer::resetAccumulation() {
    m_renderData.numAccumulationFrames = 0u;
    if (m_ctx) m_api->reset_accumulation(m_ctx);
}

bool RayTracer::loadMeshAndBvhData(uint32_t maxDepth, uint32_t maxLeafTriangles) {
    const std::vector<Mesh>& meshes = m_pResourceManager->meshes();
    const uint32_t numMeshes = (uint32_t)meshes.size();
    if (!numMeshes) return true;  // :132-133

    // Trees are independent per mesh: build them on all cores, then concatenate in mesh order
    // exactly like the reference's serial loop (:158-207).
    std::vector<std::vector<BvhNode>> trees(numMeshes);
    {
        std::atomic<uint32_t> next{0};
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), numMeshes));
        auto work = [&]() {
            BvhBuilder builder(maxLeafTriangles, maxDepth);
            for (;;) {
                uint32_t i = next.fetch_add(1);
                if (i >= numMeshes) break;
                builder.buildTree(meshes[i].trianglesPos.data(), (uint32_t)meshes[i].trianglesPos.size(), meshes[i].boundingBox);
                trees[i] = builder.getTree();
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }

    m_meshDescs.resize(numMeshes);
    uint32_t numTotalTriangles = 0u;
    for (const Mesh& m : meshes) numTotalTriangles += (uint32_t)m.trianglesPos.size();
    m_gpuTrianglePositions.clear();
    m_gpuTriangleInfo.clear();
    m_gpuTrianglePositions.reserve(numTotalTriangles);
    m_gpuTriangleInfo.reserve(numTotalTriangles);
    m_bvhTreeNodes.clear();

    uint32_t triBufferCurStartIdx = 0;
    for (uint32_t i = 0; i < numMeshes; i++) {
        const Mesh& mesh = meshes[i];
        const std::vector<BvhNode>& nodes = trees[i];
        const uint32_t numNodes = (uint32_t)nodes.size();
        const uint32_t gpuNodesPrevSize = (uint32_t)m_bvhTreeNodes.size();
        m_bvhTreeNodes.resize(gpuNodesPrevSize + numNodes);
        uint32_t localTriStart = 0u;
        for (uint32_t k = 0; k < numNodes; k++) {
            GPUNode& gpuNode = m_bvhTreeNodes[gpuNodesPrevSize + k];
            const BvhNode& bvhNode = nodes[k];
            std::memcpy(gpuNode.bbMin, &bvhNode.boundingBox.min, 12);
            std::memcpy(gpuNode.bbMax, &bvhNode.boundingBox.max, 12);
            gpuNode.firstChildIdx = bvhNode.firstChildIdx == INVALID_UINT ? INVALID_UINT : bvhNode.firstChildIdx + gpuNodesPrevSize;  // :181
            if (!bvhNode.isLeaf()) continue;
            const uint32_t numTriIndicies = (uint32_t)bvhNode.triIndicies.size();
            gpuNode.triStart = triBufferCurStartIdx + localTriStart;
            gpuNode.triEnd = gpuNode.triStart + numTriIndicies;
            localTriStart += numTriIndicies;
            for (uint32_t j = 0; j < numTriIndicies; j++) {  // triangles are emitted in LEAF order (:191-198)
                m_gpuTrianglePositions.push_back(mesh.trianglesPos[bvhNode.triIndicies[j]]);
                m_gpuTriangleInfo.push_back(mesh.trianglesInfo[bvhNode.triIndicies[j]]);
            }
        }
        m_meshDescs[i].numBvhNodes = numNodes;
        m_meshDescs[i].bvhTreeStartIdx = gpuNodesPrevSize;
        m_meshDescs[i].startIdx = triBufferCurStartIdx;
        m_meshDescs[i].endIdx = triBufferCurStartIdx + (uint32_t)mesh.trianglesPos.size();
        triBufferCurStartIdx += (uint32_t)mesh.trianglesPos.size();
    }
    // m_trianglePositions / m_triangleInfo / m_bvhTree .initiate (:209-211)
    return check(m_api->set_table(m_ctx, OKRT_TABLE_TRI_POS, m_gpuTrianglePositions.data(), sizeof(Triangle), numTotalTriangles), "t0") &&
           check(m_api->set_table(m_ctx, OKRT_TABLE_TRI_INFO, m_gpuTriangleInfo.data(), sizeof(TriangleInfo), numTotalTriangles), "t1") &&
           check(m_api->set_table(m_ctx, OKRT_TABLE_BVH, m_bvhTreeNodes.data(), sizeof(GPUNode), (uint32_t)m_bvhTreeNodes.size()), "t2");
}

bool RayTracer::render() {
    if (!m_ctx || !m_pScene) {
        m_error = "render: no device context or scene";
        return false;
    }
    calculateProjectionData();
    if (!updateBuffers()) return false;
    // the result-target clear, SRV/UAV binding and Dispatch of :226-259 are all inside okrt_render
    return check(m_api->render(m_ctx, &m_renderData), "render");
}

bool RayTracer::renderFrames(uint32_t frames) {
    if (!frames) return true;
    if (!m_ctx || !m_pScene) {
        m_error = "render: no device context or scene";
        return false;
    }
    calculateProjectionData();
    if (!updateBuffers()) return false;  // numAccumulationFrames now holds the FIRST frame's index
    if (!check(m_api->render_frames(m_ctx, &m_renderData, frames), "render_frames")) return false;
    m_renderData.numAccumulationFrames += m_renderData.accumulationEnabled * (frames - 1);
    return true;
}

void RayTracer::calculateProjectionData() {
    int camId = m_pScene ? m_pScene->firstCamera() : -1;
    if (camId < 0) return;
    const Entity& cam = m_pScene->get((uint32_t)camId);
    const Camera& camData = cam.camera;
    const Transform& camTra = cam.transform;

    const float W = (float)m_renderData.textureDims[0], H = (float)m_renderData.textureDims[1];
    const float aspectRatio = W / H;
    const float sideB = camData.nearZ;
    const float vAngleRadians = radians(camData.fov * 0.5f);
    const float sideA = std::tan(vAngleRadians) * sideB;
    m_renderData.viewPlaneDims[0] = sideA * 2.f;
    m_renderData.viewPlaneDims[1] = m_renderData.viewPlaneDims[0] / aspectRatio;

    m_renderData.cameraPosition[0] = camTra.position.x;
    m_renderData.cameraPosition[1] = camTra.position.y;
    m_renderData.cameraPosition[2] = camTra.position.z;
    m_renderData.cameraNearZ = camData.nearZ;

    Mat4 invProj = transpose(inverse(perspectiveFovLH(radians(camData.fov), W, H, camData.nearZ, camData.farZ)));
    std::memcpy(m_renderData.cameraInverseProjectionMatrix, invProj.c, 64);

    Vec3 camForward = camTra.getForwardVec(), camRight = camTra.getRightVec(), camUp = camTra.getUpVec();
    m_renderData.cameraRightDir[0] = camRight.x; m_renderData.cameraRightDir[1] = camRight.y; m_renderData.cameraRightDir[2] = camRight.z;
    m_renderData.cameraUpDir[0] = camUp.x; m_renderData.cameraUpDir[1] = camUp.y; m_renderData.cameraUpDir[2] = camUp.z;

    Mat4 invView = transpose(inverse(lookAtLH(camTra.position, camTra.position + camForward, Vec3(0.f, 1.f, 0.f))));
    std::memcpy(m_renderData.cameraInverseViewMatrix, invView.c, 64);
}

bool RayTracer::loadTextureData() {
    const std::vector<Texture>& textures = m_pResourceManager->textures();
    uint32_t numTextures = (uint32_t)textures.size();
    m_scaledTextures.clear();
    m_textureArraySize = 0;
    if (!numTextures) return true;  // :315-316
    uint32_t totArea = 0;
    for (const Texture& t : textures) totArea += t.width * t.height;
    // glm::sqrt on an integer quotient resolves to std::sqrt(double) (:324)
    uint32_t newSize = (uint32_t)std::sqrt((double)(totArea / textures.size()));
    if (!newSize) newSize = 1;
    m_textureArraySize = newSize;
    m_scaledTextures.resize((size_t)numTextures * newSize * newSize * 4);
    for (uint32_t i = 0; i < numTextures; ++i) {
        uint8_t* dst = m_scaledTextures.data() + (size_t)i * newSize * newSize * 4;
        const Texture& t = textures[i];
        if (t.width == newSize && t.height == newSize) {  // Okay::scaleTexture early-out (DX11.cpp:234-235)
            std::memcpy(dst, t.rgba8.data(), (size_t)newSize * newSize * 4);
        } else if (!check(m_api->scale_texture_rgba8(m_ctx, t.rgba8.data(), t.width, t.height, dst, newSize, newSize), "scaleTexture")) {
            return false;
        }
    }
    return check(m_api->set_textures(m_ctx, m_scaledTextures.data(), numTextures, newSize, newSize), "createTextureArray");
}

bool RayTracer::cutEnvironmentCross(const uint8_t* img, uint32_t imgW, uint32_t imgH, std::vector<uint8_t>& faces, uint32_t& face) {
    uint32_t width = imgW / 4u, height = imgH / 3u;  // :352-353
    face = 0;
    faces.clear();
    if (!img || !width || width != height) return false;  // a TextureCube needs square faces
    face = width;
    faces.resize((size_t)6 * width * height * 4);
    const uint32_t* px = (const uint32_t*)img;
    // +X (row 1, col 2)  -X (1,0)  +Y (0,1)  -Y (2,1)  +Z (1,1)  -Z (1,3)   (:377-399)
    const uint32_t cellRow[6] = {1, 1, 0, 2, 1, 1}, cellCol[6] = {2, 0, 1, 1, 1, 3};
    for (int f = 0; f < 6; ++f) {
        const uint32_t* cursor = px + (size_t)imgW * height * cellRow[f] + (size_t)width * cellCol[f];
        uint32_t* target = (uint32_t*)(faces.data() + (size_t)f * width * height * 4);
        for (uint32_t r = 0; r < height; ++r) {
            std::memcpy(target, cursor, (size_t)width * 4);
            target += width;
            cursor += imgW;
        }
    }
    return true;
}

bool RayTracer::loadEnvironmentMap(const uint8_t* img, uint32_t w, uint32_t h) {
    if (!img) return check(m_api->set_envmap(m_ctx, nullptr, 0), "envmap");  // :349-350: no image, nothing bound
    std::vector<uint8_t> faces;
    uint32_t face = 0;
    if (!cutEnvironmentCross(img, w, h, faces, face)) return check(m_api->set_envmap(m_ctx, nullptr, 0), "envmap");
    return check(m_api->set_envmap(m_ctx, faces.data(), face), "envmap");
}

bool RayTracer::updateBuffers() {
    // Light tables (:438-481). The reference's Map callback never advances its write pointer, so
    // only element 0 of each table is ever valid there (quirk Q17); whole arrays are written here.
    std::vector<GPU_DirectionalLight> dl;
    for (uint32_t id : m_pScene->view([](const Entity& e) { return e.hasDirLight; })) {
        const Entity& e = m_pScene->get(id);
        GPU_DirectionalLight g;
        g.light = e.dirLight;
        g.light.effectiveAngle = std::cos(radians(e.dirLight.effectiveAngle));
        Vec3 f = e.transform.getForwardVec();
        g.direction[0] = f.x; g.direction[1] = f.y; g.direction[2] = f.z;
        dl.push_back(g);
    }
    std::vector<GPU_PointLight> pl;
    for (uint32_t id : m_pScene->view([](const Entity& e) { return e.hasPointLight; })) {
        const Entity& e = m_pScene->get(id);
        GPU_PointLight g;
        g.light = e.pointLight;
        g.position[0] = e.transform.position.x; g.position[1] = e.transform.position.y; g.position[2] = e.transform.position.z;
        pl.push_back(g);
    }
    std::vector<GPU_SpotLight> sl;
    for (uint32_t id : m_pScene->view([](const Entity& e) { return e.hasSpotLight; })) {
        const Entity& e = m_pScene->get(id);
        GPU_SpotLight g;
        g.light = e.spotLight;
        g.light.maxAngle = std::cos(radians(e.spotLight.maxAngle));
        g.position[0] = e.transform.position.x; g.position[1] = e.transform.position.y; g.position[2] = e.transform.position.z;
        Vec3 f = e.transform.getForwardVec();
        g.direction[0] = f.x; g.direction[1] = f.y; g.direction[2] = f.z;
        sl.push_back(g);
    }
    // Spheres: the reference's upload is commented out and numSpheres forced to 0 (:757-772,781,
    // quirk Q18) although the kernel path is live; here position + Sphere are written as the
    // 84-byte records the shader declares (RayTracer.cpp:88, GPU-Structs.hlsli:86-93).
    std::vector<GPU_Sphere> sp;
    for (uint32_t id : m_pScene->view([](const Entity& e) { return e.hasSphere; })) {
        const Entity& e = m_pScene->get(id);
        GPU_Sphere g;
        g.position[0] = e.transform.position.x; g.position[1] = e.transform.position.y; g.position[2] = e.transform.position.z;
        g.material = e.sphere.material;
        g.radius = e.sphere.radius;
        sp.push_back(g);
    }
    if (!check(m_api->set_table(m_ctx, OKRT_TABLE_DIR_LIGHT, dl.data(), sizeof(GPU_DirectionalLight), (uint32_t)dl.size()), "t8")) return false;
    if (!check(m_api->set_table(m_ctx, OKRT_TABLE_POINT_LIGHT, pl.data(), sizeof(GPU_PointLight), (uint32_t)pl.size()), "t9")) return false;
    if (!check(m_api->set_table(m_ctx, OKRT_TABLE_SPOT_LIGHT, sl.data(), sizeof(GPU_SpotLight), (uint32_t)sl.size()), "t10")) return false;
    if (!check(m_api->set_table(m_ctx, OKRT_TABLE_SPHERE, sp.data(), sizeof(GPU_Sphere), (uint32_t)sp.size()), "t6")) return false;

    m_renderData.numDirLights = (uint32_t)dl.size();
    m_renderData.numPointLights = (uint32_t)pl.size();
    m_renderData.numSpotLights = (uint32_t)sl.size();
    m_renderData.numSpheres = (uint32_t)sp.size();
    m_renderData.numAccumulationFrames += m_renderData.accumulationEnabled;  // :489
    return true;
}

// ---- culling octree (reference RayTracer.cpp:503-788) ---------------------------------------------------------------
// What the reference computes, restated on flat arrays: the scene's mesh and sphere entities become `items` (entity +
// the box the reference files it under: the LOCAL mesh box, quirk Q22, or the sphere's world box); a cell holds indices
// into `items`; cells are split depth-first from a LIFO work list, which fixes the cell numbering the shader's
// firstChildIdx / numChildren rely on. The float arithmetic follows the reference operation for operation (the tables
// must be bit-identical: tests/test_host_layer.py); the data structures and control flow are this file's own.
namespace {

struct SceneItem { uint32_t entity; AABB box; };
struct OctCell {
    AABB bounds;
    std::vector<uint32_t> members;  // indices into the item array, in filing order
    uint32_t child[8] = {INVALID_UINT, INVALID_UINT, INVALID_UINT, INVALID_UINT, INVALID_UINT, INVALID_UINT, INVALID_UINT, INVALID_UINT};
    uint32_t depth = 0;
};

// Union of the members' boxes (refitOctTreeNode, :503-519; every item is a mesh or a sphere, so the reference's
// component test is always true here).
AABB fitToMembers(const std::vector<uint32_t>& members, const std::vector<SceneItem>& items) {
    AABB fit;
    for (uint32_t m : members) {
        fit.growTo(items[m].box.min);
        fit.growTo(items[m].box.max);
    }
    return fit;
}

// Octant k of a cell: the cell shrunk to half size about its centre, then shifted by +-half of the shrunk extent per
// axis. Sign table and operation order of :579-590,620-640 (sign bit pattern: x = !(k & 1), y = k & 2, z = !(k & 4)).
AABB octantBounds(const AABB& cell, uint32_t k) {
    const Vec3 centre = (cell.max + cell.min) * 0.5f;
    AABB half;
    half.max = (cell.max - centre) * 0.5f + centre;
    half.min = (cell.min - centre) * 0.5f + centre;
    const Vec3 shift = (half.max - half.min) * 0.5f;
    const float sign[3] = {(k & 1u) ? -1.f : 1.f, (k & 2u) ? 1.f : -1.f, (k & 4u) ? -1.f : 1.f};
    AABB out = half;
    for (int a = 0; a < 3; ++a) {
        out.max[a] += sign[a] * shift[a];
        out.min[a] += sign[a] * shift[a];
    }
    return out;
}

}  // namespace

bool RayTracer::createOctTree(const Scene& scene, uint32_t maxDepth, uint32_t maxLeafObjects) {
    m_pScene = &scene;
    std::vector<SceneItem> items;
    std::vector<OctCell> cells(1);

    // Root: bounds over the eight TRANSFORMED corners of every mesh box and over every sphere's box (:539-577). The
    // corners come from DirectX::BoundingBox::GetCorners fed with centre and (min - max) / 2, i.e. NEGATIVE extents: the
    // corner order below is g_BoxOffset's; with min/max accumulation the order only matters for NaNs.
    for (uint32_t id : scene.view([](const Entity& e) { return e.hasMesh; })) {
        const Entity& ent = scene.get(id);
        const AABB& local = m_pResourceManager->meshes()[ent.mesh.meshID].boundingBox;
        const Vec3 centre = (local.min + local.max) * 0.5f, extent = (local.min - local.max) * 0.5f;
        const Mat4 toWorld = ent.transform.calculateMatrix();
        for (uint32_t corner = 0; corner < 8; ++corner) {
            // g_BoxOffset: x = -,+,+,-  y = -,-,+,+ (repeating), z = + for the first four corners, - for the rest
            const float sx = ((corner & 3u) == 1u || (corner & 3u) == 2u) ? 1.f : -1.f, sy = (corner & 2u) ? 1.f : -1.f, sz = corner < 4u ? 1.f : -1.f;
            const Vec3 p{extent.x * sx + centre.x, extent.y * sy + centre.y, extent.z * sz + centre.z};
            float w[4];
            mulPoint(toWorld, p, 1.f, w);
            cells[0].bounds.growTo(Vec3(w[0], w[1], w[2]));
        }
        cells[0].members.push_back((uint32_t)items.size());
        items.push_back({id, local});  // filed under its LOCAL box (Q22, :562)
    }
    for (uint32_t id : scene.view([](const Entity& e) { return e.hasSphere; })) {
        const Entity& ent = scene.get(id);
        AABB box;
        box.min = Vec3(-ent.sphere.radius) + ent.transform.position;
        box.max = Vec3(ent.sphere.radius) + ent.transform.position;
        cells[0].bounds.growTo(box.min);
        cells[0].bounds.growTo(box.max);
        cells[0].members.push_back((uint32_t)items.size());
        items.push_back({id, box});
    }

    std::vector<uint32_t> work = {0u};      // LIFO: the cell created last is split first (:592-612)
    std::vector<uint32_t> candidates, taken;
    std::vector<uint8_t> consumed;
    while (!work.empty()) {
        const uint32_t ci = work.back();
        work.pop_back();
        if (cells[ci].depth >= maxDepth || cells[ci].members.size() <= maxLeafObjects) continue;

        // Only items clearly smaller than the cell may move down (area ratio < 0.75, :636-640); the count the
        // "everything went into one octant" test uses is the cell's population BEFORE the split (:635).
        const float cellArea = cells[ci].bounds.getArea();
        const size_t population = cells[ci].members.size();
        candidates.clear();
        for (uint32_t m : cells[ci].members)
            if (items[m].box.getArea() / cellArea < 0.75f) candidates.push_back(m);
        consumed.assign(candidates.size(), 0);

        for (uint32_t k = 0; k < 8u; ++k) {
            const AABB octant = octantBounds(cells[ci].bounds, k);
            // an item goes to the FIRST octant it overlaps; once claimed it is gone for the later octants, also when
            // the claim is then dropped because the octant swallowed the whole population (:656-675)
            taken.clear();
            for (size_t c = 0; c < candidates.size(); ++c)
                if (!consumed[c] && AABB::intersects(octant, items[candidates[c]].box)) {
                    consumed[c] = 1;
                    taken.push_back(candidates[c]);
                }
            if (taken.empty() || taken.size() == population) continue;

            // every claimed item takes ONE entry out of the cell: the first one filed under the same ENTITY (:672-682). For
            // an entity that carries a mesh and a sphere (two items, one id) that is the earlier-filed item, whichever
            // of the two was claimed — the reference's tables keep that quirk, so this does too
            std::vector<uint32_t>& stay = cells[ci].members;
            for (uint32_t t : taken) {
                const auto it = std::find_if(stay.begin(), stay.end(), [&](uint32_t m) { return items[m].entity == items[t].entity; });
                if (it != stay.end()) stay.erase(it);
            }
            const uint32_t childIdx = (uint32_t)cells.size();
            cells.emplace_back();
            cells[childIdx].members = taken;
            cells[childIdx].bounds = fitToMembers(taken, items);  // children shrink to their contents (:697)
            cells[childIdx].depth = cells[ci].depth + 1u;
            cells[ci].child[k] = childIdx;
            work.push_back(childIdx);
        }
    }

    // Flatten (:706-788): cells keep their numbers; a cell's children were created back to back, so (first, count)
    // describes them; the instance table lists every cell's meshes in filing order, cell after cell.
    m_octTreeNodes.assign(cells.size(), GPU_OctTreeNode{});
    m_gpuMeshes.clear();
    for (size_t i = 0; i < cells.size(); ++i) {
        GPU_OctTreeNode& out = m_octTreeNodes[i];
        std::memcpy(out.bbMin, &cells[i].bounds.min, 12);
        std::memcpy(out.bbMax, &cells[i].bounds.max, 12);
        for (uint32_t c : cells[i].child)
            if (c != INVALID_UINT) {
                if (!out.numChildren) out.firstChildIdx = c;
                out.numChildren++;
            }
        out.meshesStartIdx = (uint32_t)m_gpuMeshes.size();
        for (uint32_t m : cells[i].members) {
            const Entity& ent = scene.get(items[m].entity);
            if (!ent.hasMesh) continue;
            const MeshDesc& desc = m_meshDescs[ent.mesh.meshID];
            const AABB& local = m_pResourceManager->meshes()[ent.mesh.meshID].boundingBox;
            GPU_MeshComponent inst;
            const Mat4 rows = transpose(ent.transform.calculateMatrix());  // uploaded transposed, inverse of the transposed (:747-749)
            const Mat4 rowsInverse = inverse(rows);
            std::memcpy(inst.transformMatrix, rows.c, 64);
            std::memcpy(inst.inverseTransformMatrix, rowsInverse.c, 64);
            inst.triStart = desc.startIdx;
            inst.triEnd = desc.endIdx;
            std::memcpy(inst.bbMin, &local.min, 12);
            std::memcpy(inst.bbMax, &local.max, 12);
            inst.material = ent.mesh.material;
            inst.bvhNodeStartIdx = desc.bvhTreeStartIdx;
            m_gpuMeshes.push_back(inst);
        }
        out.meshesEndIdx = (uint32_t)m_gpuMeshes.size();
    }
    m_renderData.numMeshes = 0u;  // :780
    return check(m_api->set_table(m_ctx, OKRT_TABLE_MESH, m_gpuMeshes.data(), sizeof(GPU_MeshComponent), (uint32_t)m_gpuMeshes.size()), "t7") &&
           check(m_api->set_table(m_ctx, OKRT_TABLE_OCT, m_octTreeNodes.data(), sizeof(GPU_OctTreeNode), (uint32_t)m_octTreeNodes.size()), "t5");
}

}  // namespace okh
