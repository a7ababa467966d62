/*
 * ref_shim.cpp — ORACLE (test infrastructure). Thin C entry points around the REAL reference
 * sources, compiled where they lie under /root/reference (never copied):
 *   source/Graphics/BvhBuilder.cpp + BvhBuilder.h + Mesh.h, source/Utilities.h, source/SMath.h,
 *   source/Scene/Components.h, and the vendored glm under deps/include.
 * Output goes to oracle/_ref/libref_host.so (git-ignored). Used to
 *   (1) validate the product's BVH builder against the reference builder (tests, this container),
 *   (2) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py).
 * It exists only in this container; the GPU box uses the committed fixtures.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "Graphics/BvhBuilder.h"
#include "Scene/Components.h"
#include "glm/gtc/matrix_transform.hpp"

extern "C" {

// BvhBuilder(maxLeafTriangles, maxDepth).buildTree(mesh)  (BvhBuilder.cpp:7-36), flattened like
// RayTracer::loadMeshAndBvhData does for ONE mesh (RayTracer.cpp:158-207): nodes as
// {min[3],max[3],triStart,triEnd,firstChild} (36 B) and the leaf-order triangle permutation.
// Returns the node count; call with out_nodes == nullptr to size the buffers.
uint32_t ref_bvh_build(const float* tris9, uint32_t numTris, uint32_t maxLeafTriangles, uint32_t maxDepth,
                       void* out_nodes36, uint32_t nodeCapacity, uint32_t* out_triOrder) {
    MeshData md;
    md.positions.resize((size_t)numTris * 3);
    md.normals.resize((size_t)numTris * 3);
    md.uvs.resize((size_t)numTris * 3);
    md.tangents.resize((size_t)numTris * 3);
    md.bitangents.resize((size_t)numTris * 3);
    md.boundingBox = Okay::AABB();
    for (size_t i = 0; i < (size_t)numTris * 3; ++i) {
        glm::vec3 p(tris9[3 * i], tris9[3 * i + 1], tris9[3 * i + 2]);
        md.positions[i] = p;
        md.boundingBox.min = glm::min(p, md.boundingBox.min);  // Importer.cpp:52-53
        md.boundingBox.max = glm::max(p, md.boundingBox.max);
    }
    Mesh mesh(md, "oracle");
    BvhBuilder builder(maxLeafTriangles, maxDepth);
    builder.buildTree(mesh);
    const std::vector<BvhNode>& nodes = builder.getTree();
    uint32_t n = (uint32_t)nodes.size();
    if (!out_nodes36 || nodeCapacity < n) return n;
    struct GPUNode {
        float mn[3], mx[3];
        uint32_t triStart, triEnd, firstChildIdx;
    };
    static_assert(sizeof(GPUNode) == 36, "layout");
    GPUNode* out = (GPUNode*)out_nodes36;
    uint32_t localTriStart = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const BvhNode& b = nodes[k];
        GPUNode g;
        memcpy(g.mn, &b.boundingBox.min, 12);
        memcpy(g.mx, &b.boundingBox.max, 12);
        g.triStart = g.triEnd = Okay::INVALID_UINT;  // RayTracer.h:26-27
        g.firstChildIdx = b.firstChildIdx;
        if (b.isLeaf()) {
            g.triStart = localTriStart;
            g.triEnd = localTriStart + (uint32_t)b.triIndicies.size();
            for (uint32_t t : b.triIndicies) out_triOrder[localTriStart++] = t;
        }
        out[k] = g;
    }
    return n;
}

// Transform::calculateMatrix (Components.h:13-18) -> 16 floats in glm memory order.
void ref_transform_matrix(const float pos[3], const float rotDeg[3], const float scale[3], float out16[16]) {
    Transform t;
    t.position = glm::vec3(pos[0], pos[1], pos[2]);
    t.rotation = glm::vec3(rotDeg[0], rotDeg[1], rotDeg[2]);
    t.scale = glm::vec3(scale[0], scale[1], scale[2]);
    glm::mat4 m = t.calculateMatrix();
    memcpy(out16, &m[0][0], 64);
}

// The instance matrices exactly as loadOctTree uploads them (RayTracer.cpp:747-749).
void ref_instance_matrices(const float pos[3], const float rotDeg[3], const float scale[3], float outM[16], float outInv[16]) {
    Transform t;
    t.position = glm::vec3(pos[0], pos[1], pos[2]);
    t.rotation = glm::vec3(rotDeg[0], rotDeg[1], rotDeg[2]);
    t.scale = glm::vec3(scale[0], scale[1], scale[2]);
    glm::mat4 transformMatrix = glm::transpose(t.calculateMatrix());
    glm::mat4 inv = glm::inverse(transformMatrix);
    memcpy(outM, &transformMatrix[0][0], 64);
    memcpy(outInv, &inv[0][0], 64);
}

// RayTracer::calculateProjectionData (RayTracer.cpp:268-308) with the real glm:
// out = {viewPlaneDims[2], invProj[16], invView[16], camPos[3], nearZ, up[3], right[3], forward[3]}.
void ref_camera(const float pos[3], const float rotDeg[3], float fovDeg, float nearZ, float farZ, uint32_t W, uint32_t H,
                float out[48]) {
    Transform camTra;
    camTra.position = glm::vec3(pos[0], pos[1], pos[2]);
    camTra.rotation = glm::vec3(rotDeg[0], rotDeg[1], rotDeg[2]);
    const glm::vec2 windowDimsVec((float)W, (float)H);
    const float aspectRatio = windowDimsVec.x / windowDimsVec.y;
    const float sideB = nearZ;
    const float vAngleRadians = glm::radians(fovDeg * 0.5f);
    const float sideA = glm::tan(vAngleRadians) * sideB;
    float vpx = sideA * 2.f;
    float vpy = vpx / aspectRatio;
    glm::mat4 invProj = glm::transpose(glm::inverse(glm::perspectiveFovLH(glm::radians(fovDeg), windowDimsVec.x, windowDimsVec.y, nearZ, farZ)));
    glm::vec3 camForward = camTra.getForwardVec();
    glm::vec3 camRight = camTra.getRightVec();
    glm::vec3 camUp = camTra.getUpVec();
    glm::mat4 invView = glm::transpose(glm::inverse(glm::lookAtLH(camTra.position, camTra.position + camForward, glm::vec3(0.f, 1.f, 0.f))));
    float* o = out;
    *o++ = vpx; *o++ = vpy;
    memcpy(o, &invProj[0][0], 64); o += 16;
    memcpy(o, &invView[0][0], 64); o += 16;
    *o++ = pos[0]; *o++ = pos[1]; *o++ = pos[2];
    *o++ = nearZ;
    *o++ = camUp.x; *o++ = camUp.y; *o++ = camUp.z;
    *o++ = camRight.x; *o++ = camRight.y; *o++ = camRight.z;
    *o++ = camForward.x; *o++ = camForward.y; *o++ = camForward.z;
}

// Struct sizes of the host-side mirrors, for the layout fixture.
void ref_struct_sizes(uint32_t out[8]) {
    out[0] = sizeof(Okay::AABB);
    out[1] = sizeof(Okay::Triangle);
    out[2] = sizeof(Okay::TriangleInfo);
    out[3] = sizeof(Material);
    out[4] = sizeof(GPU_MeshComponent);
    out[5] = sizeof(Sphere) + sizeof(glm::vec3);  // RayTracer.cpp:88
    out[6] = sizeof(GPU_DirectionalLight) | (sizeof(GPU_PointLight) << 8) | (sizeof(GPU_SpotLight) << 16);
    out[7] = sizeof(Okay::VertexInfo);
}
}
