// Graphics/Importer.cpp (compat) — the reference's Importer interface (source/Graphics/Importer.h:5-21) for OBJ / MTL
// files, over the host layer's reader (host/obj_loader.cpp: triangulate, ConvertToLeftHanded, one mesh per object,
// de-indexed, CalcTangentSpace, Importer.cpp:15-56,77-120). The reference implements it with assimp, which ships only
// as Windows binaries; other formats (FBX) are not read here.
#include "Graphics/Importer.h"

#include <cstring>

#include "obj_loader.h"

namespace {

void toMeshData(const okh::ObjObject& o, MeshData& out) {
    const size_t numVerts = o.positions.size() * 3;
    out.positions.resize(numVerts);
    out.normals.resize(numVerts);
    out.uvs.resize(numVerts);
    out.tangents.resize(numVerts);
    out.bitangents.resize(numVerts);
    for (size_t t = 0; t < o.positions.size(); ++t)
        for (int k = 0; k < 3; ++k) {
            const size_t v = 3 * t + k;
            const okh::Vec3& p = o.positions[t].position[k];
            const okh::VertexInfo& i = o.info[t].vertexInfo[k];
            out.positions[v] = glm::vec3(p.x, p.y, p.z);
            out.normals[v] = glm::vec3(i.normal[0], i.normal[1], i.normal[2]);
            out.uvs[v] = glm::vec2(i.uv[0], i.uv[1]);
            out.tangents[v] = glm::vec3(i.tangent[0], i.tangent[1], i.tangent[2]);
            out.bitangents[v] = glm::vec3(i.bitangent[0], i.bitangent[1], i.bitangent[2]);
            out.boundingBox.growTo(out.positions[v]);  // Importer.cpp:52-53
        }
}

}  // namespace

namespace Importer {

bool loadObjects(std::string_view filePath, std::vector<ObjectDecriptionStr>& outObjects, float scale) {
    std::vector<okh::ObjObject> objects;
    std::string error;
    if (!okh::loadObjects(std::string(filePath), objects, scale, error)) return false;
    outObjects.resize(objects.size());
    for (size_t k = 0; k < objects.size(); ++k) {
        ObjectDecriptionStr& d = outObjects[k];
        d.name = objects[k].name;
        toMeshData(objects[k], d.meshData);
        d.albedoTexturePath = objects[k].albedoTexturePath;
        d.rougnessTexturePath = objects[k].roughnessTexturePath;
        d.metallicTexturePath = objects[k].metallicTexturePath;
        d.specularTexturePath = objects[k].specularTexturePath;
        d.normalTexturePath = objects[k].normalTexturePath;
    }
    return true;
}

bool loadMesh(std::string_view filePath, MeshData& outMeshData, std::string* pOutname) {
    std::vector<ObjectDecriptionStr> objects;
    if (!loadObjects(filePath, objects, 1.f) || objects.empty()) return false;  // Importer.cpp:15-56: the file's first mesh
    outMeshData = objects[0].meshData;
    if (pOutname) *pOutname = objects[0].name;
    return true;
}

}  // namespace Importer
