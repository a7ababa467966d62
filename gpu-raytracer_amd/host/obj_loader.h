// obj_loader.h — minimal Wavefront OBJ/MTL reader for the host layer ("next" row 1).
//
// Stands in for the reference's assimp import (source/Graphics/Importer.cpp:15-56,77-120:
// ReadFile(Triangulate | ConvertToLeftHanded | CalcTangentSpace), one Mesh per aiMesh, de-indexed to three
// vertices per triangle, DIFFUSE -> albedo, DIFFUSE_ROUGHNESS -> roughness, METALNESS -> metallic,
// SPECULAR -> specular, NORMALS-or-DISPLACEMENT -> normal map). assimp's source is NOT vendored in the reference
// (headers + Windows binaries only), so what it does inside is unpinned; this reader follows its documented
// post-processing steps (deps/include/assimp/postprocess.h:108-114,514-535,610-624):
//   one mesh per (object/group, material) in file order; polygons fan-triangulated;
//   MakeLeftHanded: z of positions / normals / tangents / bitangents negated;
//   FlipUVs: v -> 1 - v;  FlipWindingOrder: (a, b, c) -> (c, b, a);
//   CalcTangentSpace: per-face tangent frame from the uv deltas, Gram-Schmidt against the vertex normal;
//   a mesh without `vn` gets face normals (the reference would dereference a null mNormals there).
#pragma once
#include <string>
#include <vector>

#include "bvh_builder.h"

namespace okh {

struct ObjObject {  // Importer.h ObjectDecriptionStr
    std::string name;
    std::vector<Triangle> positions;
    std::vector<TriangleInfo> info;
    std::string albedoTexturePath, roughnessTexturePath, metallicTexturePath, specularTexturePath, normalTexturePath;
};

// Importer::loadObjects (Importer.cpp:77-120). `scale` multiplies positions (Importer.cpp:40).
bool loadObjects(const std::string& filePath, std::vector<ObjObject>& outObjects, float scale, std::string& error);

}  // namespace okh
