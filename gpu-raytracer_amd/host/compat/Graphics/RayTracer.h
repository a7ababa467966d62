// Graphics/RayTracer.h (compat) — the reference's RayTracer class (source/Graphics/RayTracer.h:71-116): same name, same
// public methods and argument types (the reference's own RenderTexture-shaped target, ResourceManager and entt Scene),
// implemented over include/okrt.h through the host layer (okh::RayTracer builds the tables bit for bit). The structs
// DebugRenderer-style callers read (MeshDesc, GPUNode, GPU_OctTreeNode) keep the reference's layouts.
#pragma once
#include <memory>
#include <string_view>
#include <vector>

#include "DirectX/DX11.h"
#include "DirectX/RenderTexture.h"
#include "GPUStorage.h"
#include "Scene/Components.h"
#include "Scene/Entity.h"
#include "Utilities.h"

#include "glm/glm.hpp"

class Scene;
class ResourceManager;
namespace okh {
class RayTracer;
class ResourceManager;
class Scene;
}  // namespace okh

// Table records other parts of the reference read through the getters below (DebugRenderer does); layouts as uploaded.
struct GPU_OctTreeNode {  // t5, 48 B (reference RayTracer.h:57-69)
    Okay::AABB boundingBox;
    uint32_t meshesStartIdx = Okay::INVALID_UINT, meshesEndIdx = Okay::INVALID_UINT;
    uint32_t spheresStartIdx = Okay::INVALID_UINT, spheresEndIdx = Okay::INVALID_UINT;
    uint32_t firstChildIdx = Okay::INVALID_UINT, numChildren = 0u;
};
struct GPUNode {  // t2, 36 B (reference RayTracer.h:23-29)
    Okay::AABB boundingBox;
    uint32_t triStart = Okay::INVALID_UINT, triEnd = Okay::INVALID_UINT, firstChildIdx = Okay::INVALID_UINT;
};
struct MeshDesc { uint32_t startIdx, endIdx, bvhTreeStartIdx, numBvhNodes; };  // per mesh asset (reference RayTracer.h:15-21)

class RayTracer {
public:
    enum DebugDisplayMode : uint32_t { None = 0, BBCheckCount = 1, TriCheckCount = 2 };

    // --- per frame (Application.cpp:117, 312-317)
    void render();
    void resetAccumulation();
    void toggleAccumulation(bool enable);
    uint32_t getNumAccumulationFrames() const;
    float& getDOFDistance();
    float& getDOFStrength();
    uint32_t& getDebugMaxCount();
    void setDebugMode(DebugDisplayMode mode);
    void onResize();       // the application resized the target: follow it (RayTracer.cpp:494-501)
    void reloadShaders();  // kernels are compiled ahead of time: nothing to reload

    // --- set-up (Application.cpp:48-49, 104, 408-411)
    RayTracer();
    RayTracer(const RenderTexture& target, const ResourceManager& resourceManager, std::string_view environmentMapPath = "");
    ~RayTracer();
    void initiate(const RenderTexture& target, const ResourceManager& resourceManager, std::string_view environmentMapPath = "");
    void shutdown();
    void setScene(const Scene& scene) { m_pScene = &scene; }
    void createOctTree(const Scene& scene, uint32_t maxDepth, uint32_t maxLeafObjects);
    void loadMeshAndBvhData(uint32_t maxDepth, uint32_t maxLeafTriangles);

    // --- what was built (DebugRenderer and the ImGui panels read these)
    const GPUStorage& getTrianglesInfo() const { return m_triangleInfo; }
    const GPUStorage& getTrianglesPos() const { return m_trianglePositions; }
    const std::vector<GPU_OctTreeNode>& getOctTreeNodes() const { return m_octTreeNodes; }
    const std::vector<GPUNode>& getBvhTreeNodes() const { return m_bvhTreeNodes; }
    const std::vector<MeshDesc>& getMeshDescriptors() const { return m_meshDescs; }
    uint32_t getGlobalNodeIdx(const MeshComponent& meshComp, uint32_t localNodeIdx) const;

    // --- compat only: the device context behind this object (read-backs, options), and the last failure ("" = none)
    okrt_ctx* deviceContext() const;
    const std::string& lastError() const { return m_error; }

private:
    void syncScene(const Scene& scene);  // entt registry -> host-layer entity list, view order preserved
    void mirrorTables();                 // host-layer tables -> the vectors / storages the getters expose

    std::unique_ptr<okh::RayTracer> m_impl;
    std::shared_ptr<okay_compat::PixelSource> m_pixels;  // the link the target texture reads its pixels through
    std::unique_ptr<okh::ResourceManager> m_resources;
    std::unique_ptr<okh::Scene> m_scene;
    const Scene* m_pScene;
    const ResourceManager* m_pResourceManager;
    const RenderTexture* m_pTargetTexture;
    GPUStorage m_trianglePositions, m_triangleInfo;
    std::vector<MeshDesc> m_meshDescs;
    std::vector<GPUNode> m_bvhTreeNodes;
    std::vector<GPU_OctTreeNode> m_octTreeNodes;
    float m_dofFallback[2];
    uint32_t m_debugMaxFallback;
    std::string m_error;
};
