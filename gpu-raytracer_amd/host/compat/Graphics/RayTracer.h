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

struct MeshDesc {  // RayTracer.h:15-21
    uint32_t startIdx;
    uint32_t endIdx;
    uint32_t bvhTreeStartIdx;
    uint32_t numBvhNodes;
};

struct GPUNode {  // RayTracer.h:23-29
    Okay::AABB boundingBox;
    uint32_t triStart = Okay::INVALID_UINT;
    uint32_t triEnd = Okay::INVALID_UINT;
    uint32_t firstChildIdx = Okay::INVALID_UINT;
};

struct GPU_OctTreeNode {  // RayTracer.h:57-69
    Okay::AABB boundingBox;
    uint32_t meshesStartIdx = Okay::INVALID_UINT;
    uint32_t meshesEndIdx = Okay::INVALID_UINT;
    uint32_t spheresStartIdx = Okay::INVALID_UINT;
    uint32_t spheresEndIdx = Okay::INVALID_UINT;
    uint32_t firstChildIdx = Okay::INVALID_UINT;
    uint32_t numChildren = 0u;
};

class RayTracer {
public:
    enum DebugDisplayMode : uint32_t {
        None = 0,
        BBCheckCount = 1,
        TriCheckCount = 2,
    };

public:
    RayTracer();
    RayTracer(const RenderTexture& target, const ResourceManager& resourceManager, std::string_view environmentMapPath = "");
    ~RayTracer();

    void shutdown();
    void initiate(const RenderTexture& target, const ResourceManager& resourceManager, std::string_view environmentMapPath = "");

    void loadMeshAndBvhData(uint32_t maxDepth, uint32_t maxLeafTriangles);
    void createOctTree(const Scene& scene, uint32_t maxDepth, uint32_t maxLeafObjects);

    inline const std::vector<MeshDesc>& getMeshDescriptors() const { return m_meshDescs; }
    inline const std::vector<GPUNode>& getBvhTreeNodes() const { return m_bvhTreeNodes; }
    inline const std::vector<GPU_OctTreeNode>& getOctTreeNodes() const { return m_octTreeNodes; }

    inline const GPUStorage& getTrianglesPos() const { return m_trianglePositions; }
    inline const GPUStorage& getTrianglesInfo() const { return m_triangleInfo; }

    uint32_t getGlobalNodeIdx(const MeshComponent& meshComp, uint32_t localNodeIdx) const;

    inline void setScene(const Scene& scene) { m_pScene = &scene; }
    void render();

    void toggleAccumulation(bool enable);
    void resetAccumulation();
    uint32_t getNumAccumulationFrames() const;

    void reloadShaders();  // kernels are compiled ahead of time: nothing to reload

    float& getDOFStrength();
    float& getDOFDistance();

    void onResize();

    void setDebugMode(DebugDisplayMode mode);
    uint32_t& getDebugMaxCount();

    // compat only: the device context behind this object (read-backs, options), and the last failure ("" = none)
    okrt_ctx* deviceContext() const;
    const std::string& lastError() const { return m_error; }

private:
    void syncScene(const Scene& scene);
    void mirrorTables();

    const Scene* m_pScene;
    const ResourceManager* m_pResourceManager;
    const RenderTexture* m_pTargetTexture;

    std::unique_ptr<okh::RayTracer> m_impl;
    std::unique_ptr<okh::ResourceManager> m_resources;
    std::unique_ptr<okh::Scene> m_scene;

    GPUStorage m_trianglePositions;
    GPUStorage m_triangleInfo;
    std::vector<GPUNode> m_bvhTreeNodes;
    std::vector<MeshDesc> m_meshDescs;
    std::vector<GPU_OctTreeNode> m_octTreeNodes;
    float m_dofFallback[2];
    uint32_t m_debugMaxFallback;
    std::string m_error;
};
