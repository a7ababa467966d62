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

void RayTracer::refitOctTreeNode(OctTreeNode& node) {
    node.boundingBox = AABB();
    for (const EntityAABB& e : node.entities) {
        const Entity& ent = m_pScene->get(e.entity);
        if (ent.hasMesh || ent.hasSphere) {
            node.boundingBox.growTo(e.aabb.min);
            node.boundingBox.growTo(e.aabb.max);
        }
    }
}

bool RayTracer::createOctTree(const Scene& scene, uint32_t maxDepth, uint32_t maxLeafObjects) {
    m_pScene = &scene;
    std::vector<OctTreeNode> nodes;
    nodes.emplace_back();
    {
        OctTreeNode& root = nodes.back();
        for (uint32_t id : scene.view([](const Entity& e) { return e.hasMesh; })) {  // :539-563
            const Entity& ent = scene.get(id);
            const Mesh& mesh = m_pResourceManager->meshes()[ent.mesh.meshID];
            AABB meshBB = mesh.boundingBox;
            Vec3 center = (meshBB.min + meshBB.max) * 0.5f;
            Vec3 extents = (meshBB.min - meshBB.max) * 0.5f;  // negative extents, as in the reference (:546)
            // DirectX::BoundingBox::GetCorners: Center + Extents * offset, offsets in g_BoxOffset order
            static const float OFF[8][3] = {{-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}, {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}};
            Mat4 matrix = ent.transform.calculateMatrix();
            for (int v = 0; v < 8; ++v) {
                Vec3 corner{extents.x * OFF[v][0] + center.x, extents.y * OFF[v][1] + center.y, extents.z * OFF[v][2] + center.z};
                float p[4];
                mulPoint(matrix, corner, 1.f, p);
                root.boundingBox.growTo(Vec3(p[0], p[1], p[2]));
            }
            root.entities.push_back({id, meshBB});  // the LOCAL mesh box (quirk Q22, :562)
        }
        for (uint32_t id : scene.view([](const Entity& e) { return e.hasSphere; })) {  // :565-577
            const Entity& ent = scene.get(id);
            float r = ent.sphere.radius;
            AABB sphereBB;
            sphereBB.min = Vec3(-r) + ent.transform.position;
            sphereBB.max = Vec3(r) + ent.transform.position;
            root.boundingBox.growTo(sphereBB.min);
            root.boundingBox.growTo(sphereBB.max);
            root.entities.push_back({id, sphereBB});
        }
    }

    static const float CHILD_BBS_OFFSETS[8][3] = {{1, -1, 1}, {-1, -1, 1}, {1, 1, 1}, {-1, 1, 1}, {1, -1, -1}, {-1, -1, -1}, {1, 1, -1}, {-1, 1, -1}};  // :579-590
    struct NodeStack { uint32_t parent, node, depth; };
    std::stack<NodeStack> stack;
    std::vector<EntityAABB> childEntities, childEntitiesPool;
    stack.push({INVALID_UINT, 0u, 0u});

    while (!stack.empty()) {  // :611-701
        NodeStack sd = stack.top();
        stack.pop();
        if (sd.depth >= maxDepth || nodes[sd.node].entities.size() <= maxLeafObjects) continue;

        AABB defaultChildBB = nodes[sd.node].boundingBox;
        Vec3 bbCenter = (defaultChildBB.max + defaultChildBB.min) * 0.5f;
        defaultChildBB.max = defaultChildBB.max - bbCenter;
        defaultChildBB.min = defaultChildBB.min - bbCenter;
        defaultChildBB.min = defaultChildBB.min * 0.5f;
        defaultChildBB.max = defaultChildBB.max * 0.5f;
        defaultChildBB.max = defaultChildBB.max + bbCenter;
        defaultChildBB.min = defaultChildBB.min + bbCenter;

        childEntitiesPool.clear();
        float nodeBBArea = nodes[sd.node].boundingBox.getArea();
        uint32_t totalChildren = (uint32_t)nodes[sd.node].entities.size();
        for (int k = (int)nodes[sd.node].entities.size() - 1; k >= 0; k--)
            if (nodes[sd.node].entities[k].aabb.getArea() / nodeBBArea < 0.75f) childEntitiesPool.push_back(nodes[sd.node].entities[k]);

        Vec3 ext = (defaultChildBB.max - defaultChildBB.min) * 0.5f;
        for (uint32_t i = 0; i < 8u; i++) {
            AABB childBB = defaultChildBB;
            for (int k = 0; k < 3; k++) {
                childBB.max[k] += CHILD_BBS_OFFSETS[i][k] * ext[k];
                childBB.min[k] += CHILD_BBS_OFFSETS[i][k] * ext[k];
            }
            childEntities.clear();
            for (int k = (int)childEntitiesPool.size() - 1; k >= 0; k--) {
                if (!AABB::intersects(childBB, childEntitiesPool[k].aabb)) continue;
                childEntities.push_back(childEntitiesPool[k]);
                childEntitiesPool.erase(childEntitiesPool.begin() + k);
            }
            if (childEntities.empty()) continue;
            if (childEntities.size() == totalChildren) continue;  // (their pool entries stay consumed, as in the reference)
            for (const EntityAABB& ce : childEntities) {
                std::vector<EntityAABB>& pe = nodes[sd.node].entities;
                for (size_t j = 0; j < pe.size(); j++)
                    if (pe[j].entity == ce.entity) {
                        pe.erase(pe.begin() + j);
                        break;
                    }
            }
            nodes.emplace_back();
            uint32_t childIdx = (uint32_t)nodes.size() - 1;
            nodes[sd.node].children[i] = childIdx;
            nodes[childIdx].entities = childEntities;
            refitOctTreeNode(nodes[childIdx]);
            stack.push({sd.node, childIdx, sd.depth + 1u});
        }
    }
    return loadOctTree(nodes);
}

bool RayTracer::loadOctTree(const std::vector<OctTreeNode>& nodes) {
    m_octTreeNodes.assign(nodes.size(), GPU_OctTreeNode{});
    m_gpuMeshes.clear();
    for (uint32_t i = 0; i < (uint32_t)nodes.size(); i++) {
        const OctTreeNode& node = nodes[i];
        GPU_OctTreeNode& gpuNode = m_octTreeNodes[i];
        std::memcpy(gpuNode.bbMin, &node.boundingBox.min, 12);
        std::memcpy(gpuNode.bbMax, &node.boundingBox.max, 12);
        for (uint32_t k = 0; k < 8u; k++) {  // :720-729
            if (node.children[k] == INVALID_UINT) continue;
            if (gpuNode.firstChildIdx == INVALID_UINT) gpuNode.firstChildIdx = node.children[k];
            gpuNode.numChildren += 1;
        }
        uint32_t numMeshes = 0u;
        for (const EntityAABB& e : node.entities) {  // :732-755
            const Entity& ent = m_pScene->get(e.entity);
            if (!ent.hasMesh) continue;
            GPU_MeshComponent gpuMesh;
            gpuMesh.triStart = m_meshDescs[ent.mesh.meshID].startIdx;
            gpuMesh.triEnd = m_meshDescs[ent.mesh.meshID].endIdx;
            const AABB& bb = m_pResourceManager->meshes()[ent.mesh.meshID].boundingBox;
            std::memcpy(gpuMesh.bbMin, &bb.min, 12);
            std::memcpy(gpuMesh.bbMax, &bb.max, 12);
            Mat4 transformMatrix = transpose(ent.transform.calculateMatrix());  // :747-749
            Mat4 inv = inverse(transformMatrix);
            std::memcpy(gpuMesh.transformMatrix, transformMatrix.c, 64);
            std::memcpy(gpuMesh.inverseTransformMatrix, inv.c, 64);
            gpuMesh.material = ent.mesh.material;
            gpuMesh.bvhNodeStartIdx = m_meshDescs[ent.mesh.meshID].bvhTreeStartIdx;
            m_gpuMeshes.push_back(gpuMesh);
            numMeshes++;
        }
        gpuNode.meshesStartIdx = (uint32_t)m_gpuMeshes.size() - numMeshes;
        gpuNode.meshesEndIdx = gpuNode.meshesStartIdx + numMeshes;
    }
    m_renderData.numMeshes = 0u;  // :780
    return check(m_api->set_table(m_ctx, OKRT_TABLE_MESH, m_gpuMeshes.data(), sizeof(GPU_MeshComponent), (uint32_t)m_gpuMeshes.size()), "t7") &&
           check(m_api->set_table(m_ctx, OKRT_TABLE_OCT, m_octTreeNodes.data(), sizeof(GPU_OctTreeNode), (uint32_t)m_octTreeNodes.size()), "t5");
}

}  // namespace okh
