// raytracer_host.h — host side of the drop-in: the reference's `RayTracer` driver
// (source/Graphics/RayTracer.{h,cpp}) re-written over the okrt C ABI (include/okrt.h), with the
// small pieces of ResourceManager / Scene it needs (mesh + texture stores, entity components).
// Same method names and argument meaning as the reference so call sites read alike; every DX11
// call of the original is an okrt_* call here. Nothing of the reference's source is copied.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/okrt.h"
#include "bvh_builder.h"

namespace okh {

// ---- components (reference source/Scene/Components.h:26-139) ------------------------------------
struct Material {  // 68 B, identical to the GPU layout (GPU-Structs.hlsli:37-71)
    float albedo[3] = {0.f, 0.f, 0.f};
    uint32_t albedoTextureId = INVALID_UINT;
    float roughness = 1.f;
    uint32_t roughnessTextureId = INVALID_UINT;
    float metallic = 0.f;
    uint32_t metallicTextureId = INVALID_UINT;
    float specular = 1.f;
    uint32_t specularTextureId = INVALID_UINT;
    uint32_t normalMapIdx = INVALID_UINT;
    float emissionColour[3] = {1.f, 1.f, 1.f};
    float emissionPower = 0.f;
    float transparency = 0.f;
    float indexOfRefraction = 1.f;
};
static_assert(sizeof(Material) == OKRT_SIZEOF_MATERIAL, "Material layout");

struct DirectionalLight { float colour[3] = {1, 1, 1}; float intensity = 1.f; float effectiveAngle = 2.f; };
struct PointLight { float colour[3] = {1, 1, 1}; float intensity = 1.f; float radius = 1.f; };
struct SpotLight { float colour[3] = {1, 1, 1}; float intensity = 1.f; float radius = 1.f; float maxAngle = 90.f; };
struct SphereComponent { Material material; float radius = 4.f; };
struct MeshComponent { Material material; uint32_t meshID = 0u; };
struct Camera { float fov = 90.f, nearZ = 0.1f, farZ = 1000.f; };

// ---- assets (reference Mesh.h, Texture.h, ResourceManager.h) ------------------------------------
struct Mesh {
    std::vector<Triangle> trianglesPos;
    std::vector<TriangleInfo> trianglesInfo;
    AABB boundingBox;
};
struct Texture {
    std::vector<uint8_t> rgba8;
    uint32_t width = 0, height = 0;
};
class ResourceManager {
public:
    // positions: numTris*9 floats; info: numTris*33 floats (TriangleInfo layout) or nullptr (zeros).
    uint32_t addMesh(const float* positions, const float* info, uint32_t numTris);
    uint32_t addTexture(const uint8_t* rgba8, uint32_t width, uint32_t height);
    // an importer's own box for a mesh (Importer.cpp:52-53) instead of the one addMesh computed from the vertices
    void setMeshBoundingBox(uint32_t meshId, const AABB& box) { m_meshes[meshId].boundingBox = box; }
    const std::vector<Mesh>& meshes() const { return m_meshes; }
    const std::vector<Texture>& textures() const { return m_textures; }
private:
    std::vector<Mesh> m_meshes;
    std::vector<Texture> m_textures;
};

// ---- scene (the slice of the entt registry the RayTracer reads) ---------------------------------
struct Entity {
    Transform transform;
    bool hasMesh = false, hasSphere = false, hasDirLight = false, hasPointLight = false, hasSpotLight = false, hasCamera = false;
    MeshComponent mesh;
    SphereComponent sphere;
    DirectionalLight dirLight;
    PointLight pointLight;
    SpotLight spotLight;
    Camera camera;
};
class Scene {
public:
    uint32_t createEntity() { m_entities.emplace_back(); return (uint32_t)m_entities.size() - 1; }
    Entity& get(uint32_t id) { return m_entities[id]; }
    const Entity& get(uint32_t id) const { return m_entities[id]; }
    uint32_t size() const { return (uint32_t)m_entities.size(); }
    // entt 3.10 views walk a component pool's packed array back to front, i.e. in REVERSE creation
    // order (reference deps/include/Entt/entt.hpp, basic_sparse_set iterators); the octree's entity
    // order, and with it the instance-table order, inherits that (RayTracer.cpp:539-577).
    template <class Pred> std::vector<uint32_t> view(Pred has) const {
        std::vector<uint32_t> ids;
        for (uint32_t i = size(); i-- > 0;) if (has(m_entities[i])) ids.push_back(i);
        return ids;
    }
    int firstCamera() const { for (uint32_t i = size(); i-- > 0;) if (m_entities[i].hasCamera) return (int)i; return -1; }
private:
    std::vector<Entity> m_entities;
};

// ---- tables the driver builds (reference RayTracer.h:15-69) --------------------------------------
struct MeshDesc { uint32_t startIdx, endIdx, bvhTreeStartIdx, numBvhNodes; };
struct GPUNode {  // 36 B
    float bbMin[3], bbMax[3];
    uint32_t triStart = INVALID_UINT, triEnd = INVALID_UINT, firstChildIdx = INVALID_UINT;
};
struct GPU_OctTreeNode {  // 48 B
    float bbMin[3], bbMax[3];
    uint32_t meshesStartIdx = INVALID_UINT, meshesEndIdx = INVALID_UINT;
    uint32_t spheresStartIdx = INVALID_UINT, spheresEndIdx = INVALID_UINT;
    uint32_t firstChildIdx = INVALID_UINT, numChildren = 0u;
};
struct GPU_MeshComponent {  // 232 B
    float transformMatrix[16], inverseTransformMatrix[16];
    uint32_t triStart, triEnd;
    float bbMin[3], bbMax[3];
    Material material;
    uint32_t bvhNodeStartIdx;
};
struct GPU_Sphere { float position[3]; Material material; float radius; };                      // 84 B
struct GPU_DirectionalLight { DirectionalLight light; float direction[3]; };                     // 32 B
struct GPU_PointLight { PointLight light; float position[3]; };                                  // 32 B
struct GPU_SpotLight { SpotLight light; float position[3]; float direction[3]; };                // 48 B
static_assert(sizeof(GPUNode) == OKRT_SIZEOF_BVH_NODE && sizeof(GPU_OctTreeNode) == OKRT_SIZEOF_OCT_NODE, "layout");
static_assert(sizeof(GPU_MeshComponent) == OKRT_SIZEOF_MESH && sizeof(GPU_Sphere) == OKRT_SIZEOF_SPHERE, "layout");
static_assert(sizeof(GPU_DirectionalLight) == 32 && sizeof(GPU_PointLight) == 32 && sizeof(GPU_SpotLight) == 48, "layout");

class RayTracer {
public:
    enum DebugDisplayMode : uint32_t { None = 0, BBCheckCount = 1, TriCheckCount = 2 };  // RayTracer.h:74-79

    // `api` is the device layer (okrt_get_api() of libokrt.so); `device` the GPU ordinal.
    RayTracer(const okrt_api* api, int device);
    ~RayTracer();

    // RayTracer::initiate (RayTracer.cpp:60-118): target dims, texture array, environment map
    // (a 4x3 cross image, RGBA8; nullptr = none), BVHs with (30,5).
    bool initiate(uint32_t width, uint32_t height, const ResourceManager& resourceManager,
                  const uint8_t* envCrossRgba8 = nullptr, uint32_t envImgWidth = 0, uint32_t envImgHeight = 0);
    void shutdown();

    bool loadMeshAndBvhData(uint32_t maxDepth, uint32_t maxLeafTriangles);            // RayTracer.cpp:120-217
    bool createOctTree(const Scene& scene, uint32_t maxDepth, uint32_t maxLeafObjects);  // RayTracer.cpp:503-788 (build + flatten + upload)
    void setScene(const Scene& scene) { m_pScene = &scene; }
    bool render();                                                                    // RayTracer.cpp:219-260
    // `frames` consecutive render() calls with an unchanged scene and camera, handed to the device layer as one
    // okrt_render_frames (same framebuffers, bit for bit; not in the reference, whose main loop calls render() per frame)
    bool renderFrames(uint32_t frames);
    void toggleAccumulation(bool enable) { m_renderData.accumulationEnabled = enable ? 1u : 0u; resetAccumulation(); }
    void resetAccumulation();                                                         // RayTracer.h:203-209
    uint32_t getNumAccumulationFrames() const { return m_renderData.numAccumulationFrames; }
    float& getDOFStrength() { return m_renderData.dofStrength; }
    float& getDOFDistance() { return m_renderData.dofDistance; }
    bool onResize(uint32_t width, uint32_t height);                                   // RayTracer.cpp:494-501
    void setDebugMode(DebugDisplayMode mode) { m_renderData.debugMode = mode; }
    uint32_t& getDebugMaxCount() { return m_renderData.debugMaxCount; }

    const std::vector<MeshDesc>& getMeshDescriptors() const { return m_meshDescs; }
    const std::vector<GPUNode>& getBvhTreeNodes() const { return m_bvhTreeNodes; }
    const std::vector<GPU_OctTreeNode>& getOctTreeNodes() const { return m_octTreeNodes; }
    const std::vector<GPU_MeshComponent>& getGpuMeshes() const { return m_gpuMeshes; }
    const std::vector<Triangle>& getTrianglesPos() const { return m_gpuTrianglePositions; }
    const std::vector<TriangleInfo>& getTrianglesInfo() const { return m_gpuTriangleInfo; }
    uint32_t getGlobalNodeIdx(const MeshComponent& meshComp, uint32_t localNodeIdx) const {
        return m_meshDescs[meshComp.meshID].bvhTreeStartIdx + localNodeIdx;
    }

    // Not in the reference: direct access for the bench/test harness.
    okrt_ctx* deviceContext() { return m_ctx; }
    const okrt_render_data& renderData() const { return m_renderData; }
    okrt_render_data& renderDataMutable() { return m_renderData; }
    void calculateProjectionData();  // RayTracer.cpp:268-308 (public so tests can pin it)
    bool updateBuffers();            // RayTracer.cpp:434-492
    const std::string& lastError() const { return m_error; }
    // the six cube faces cut from a 4x3 cross image (RayTracer.cpp:344-401); exposed for tests
    static bool cutEnvironmentCross(const uint8_t* img, uint32_t imgW, uint32_t imgH, std::vector<uint8_t>& faces, uint32_t& face);
    uint32_t textureArraySize() const { return m_textureArraySize; }
    const std::vector<uint8_t>& scaledTextures() const { return m_scaledTextures; }

private:
    bool loadTextureData();                                                          // RayTracer.cpp:310-342
    bool loadEnvironmentMap(const uint8_t* img, uint32_t w, uint32_t h);             // RayTracer.cpp:344-432
    bool check(okrt_status s, const char* what);

    const okrt_api* m_api;
    okrt_ctx* m_ctx = nullptr;
    const Scene* m_pScene = nullptr;
    const ResourceManager* m_pResourceManager = nullptr;
    okrt_render_data m_renderData;
    std::vector<GPUNode> m_bvhTreeNodes;
    std::vector<MeshDesc> m_meshDescs;
    std::vector<GPU_MeshComponent> m_gpuMeshes;
    std::vector<GPU_OctTreeNode> m_octTreeNodes;
    std::vector<Triangle> m_gpuTrianglePositions;
    std::vector<TriangleInfo> m_gpuTriangleInfo;
    std::vector<uint8_t> m_scaledTextures;
    uint32_t m_textureArraySize = 0;
    std::string m_error;
};

}  // namespace okh
