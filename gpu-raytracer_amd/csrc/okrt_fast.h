// okrt_fast.h — interface of the MI355X traversal structure (okrt_fast_build.cpp, okrt_fast_kernel.hip).
#pragma once
#include <string>

#include "okrt_kernels.h"

namespace okrt {

struct FastInputs {  // host copies of the uploaded reference tables
    const TriangleRec* triPos; uint32_t numTris;
    const BvhNodeRec* bvh; uint32_t numNodes;
    const OctNodeRec* oct; uint32_t numOct;
    const MeshRec* meshes; uint32_t numMeshes;
};
struct FastScene;  // opaque: device-resident derived structure

bool fastBuild(FastScene** scene, const FastInputs& in, hipStream_t stream, std::string& err);
void fastDestroy(FastScene* scene);
void launchFast(const FastScene& fs, const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum,
                uint32_t* result, unsigned long long* counters, hipStream_t stream);

}  // namespace okrt
