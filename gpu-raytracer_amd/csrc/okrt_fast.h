// okrt_fast.h — the MI355X traversal structure derived from the uploaded reference tables, and
// the wavefront pipeline that traces it (okrt_fast_build.cpp, okrt_fast_build_device.hip; okrt_fast_pipeline.hip + okrt_fast_trace.hip + okrt_fast_shade.hip).
//
// Layout in HBM (all 16-byte aligned, 128-byte node stride = one L2 line per node visit):
//   Node4Q   64 B  four child boxes, 8-bit planes on a per-node power-of-two grid, child[4] references.
//                  One visit = 4 x global_load_dwordx4 per lane.
//   Tri48    48 B  p2, E1 = p0-p2, E2 = p1-p2 (the reference's Moeller-Trumbore edge convention,
//                  GPU-Utilities.hlsli:124) + source triangle index (into t0/t1) + instance index.
//                  One test = 3 x global_load_dwordx4.
// Child reference: bit 31 set = leaf {count-1 : 4 bits, first triangle : 27 bits}; else node index;
// kEmptyRef = unused slot (its box is NaN, so every slab comparison fails; an inverted +-inf box
// would NOT do: min/max un-invert it and every ray hits it).
#pragma once
#include <string>
#include <vector>

#include "okrt_kernels.h"

namespace okrt {

// Node4Q: four child boxes quantised to 8 bits per plane on a per-node grid (origin + q * 2^e per axis).
// Boxes are conservative (floor/ceil on a padded grid), so the traversal visits a superset of what exact
// boxes would: the closest hit is unchanged. 64 B = half an L2 line = 4 x global_load_dwordx4 per visit
// (the fp32 Node4 of the first version needed 7): the traversal is bound by L1/TA address processing of
// divergent 16-byte gathers, so loads per visit is the lever.
struct alignas(16) Node4Q {
    float origin[3];
    uint32_t exps;       // IEEE exponent bytes ex | ey << 8 | ez << 16: grid step of axis a = 2^(e_a - 127)
    uint32_t child[4];
    uint32_t qlo[3];     // per axis: byte k = child k's lower plane
    uint32_t pad0;
    uint32_t qhi[3];     // per axis: byte k = child k's upper plane
    uint32_t pad1;
};
static_assert(sizeof(Node4Q) == 64, "Node4Q is half a 128-byte line");

struct alignas(16) Tri48 {
    float p2[3];
    uint32_t srcTri;    // index into t0 / t1 (TriangleInfo fetch)
    float e1[3];
    uint32_t instance;  // index into t7 (material + matrix)
    float e2[3];
    uint32_t pad;
};
static_assert(sizeof(Tri48) == 48, "Tri48");

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu;  // also "no more work" in the traversal loop
constexpr uint32_t kMaxLeafTris = 15;        // 4 bits hold count-1 (15 is reserved by kEmptyRef's pattern)

struct FastInputs {  // host copies of the uploaded reference tables
    const TriangleRec* triPos; uint32_t numTris;
    const BvhNodeRec* bvh; uint32_t numNodes;
    const OctNodeRec* oct; uint32_t numOct;
    const MeshRec* meshes; uint32_t numMeshes;
    bool geometryChanged = true;  // t0 / t2 were re-uploaded since the last build (else only instances / octree moved)
};

// Host-side result of the build (okrt_fast_build.cpp); uploaded by fastBuild.
struct FastHostScene {
    std::vector<Node4Q> nodes;  // nodes[0] is the root (always present, possibly all-empty)
    std::vector<Tri48> tris;
    uint32_t maxDepth = 0;     // deepest leaf, in Node4Q levels
    bool allIdentity = true;   // every instance matrix is the identity (bit-exact with the reference order)
    bool consistent = true;    // false: the collapse met an impossible table and could not place every subtree (the build is refused)
};
void buildFastHost(const FastInputs& in, FastHostScene& out);

// Grow-only device allocation.
template <class T>
struct DeviceArray {
    T* p = nullptr;
    size_t cap = 0;
    bool need(size_t n, std::string& err, bool headroom = false) {
        if (n <= cap) return true;
        release();
        if (headroom) n += n / 8;
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e != hipSuccess) {
            (void)hipGetLastError();  // not left behind for an unrelated later check
            p = nullptr;
            err = std::string("hipMalloc: ") + hipGetErrorString(e);
            return false;
        }
        cap = n;
        return true;
    }
    void release() {
        if (p) hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Device-side build (okrt_fast_build_device.hip): the same structure, made on the GPU from the tables in HBM.
struct FastDeviceTables { const TriangleRec* triPos; const MeshRec* meshes; };
struct FastDeviceResult {
    DeviceArray<uint4>& nodes;   // 4 x uint4 per Node4Q
    DeviceArray<float4>& tris;   // 3 x float4 per Tri48
    uint32_t numNodes = 0, numTris = 0, maxDepth = 0;
    bool allIdentity = true;
    float buildMs = 0.f;         // device timeline, first kernel to last
};
// Upper levels for the device builder (okrt_fast_build.cpp): binned SAH over n cluster boxes ({min[3], max[3]} each)
// weighted by their triangle counts. Children < n are clusters, others n + index into `out`; returns the root reference.
struct TopNode { float mn[3]; uint32_t left; float mx[3]; uint32_t right; uint32_t count; };
uint32_t buildTopSAH(const float* boxes6, const uint32_t* weights, uint32_t n, std::vector<TopNode>& out);

struct FastScene;  // device-resident structure + the wavefront queues
struct DeviceBuilder;
DeviceBuilder* deviceBuilderCreate();
void deviceBuilderDestroy(DeviceBuilder*);
bool buildFastDevice(DeviceBuilder& b, const FastInputs& in, const FastDeviceTables& dev, hipStream_t stream, FastDeviceResult& out, std::string& err);


// what the camera-segment probe of the first launch set after a build counted and chose (okrt_fast_pipeline.hip: probeCameraSegment)
struct FastCameraProbe {
    uint32_t tiles, chosePacket;
    float packetNodeStepsPerTile, packetTriTestsPerTile, rayNodeVisitsPerRay, rayTriTestsPerRay, packetCost, perRayCost;
};
void fastCameraProbe(const FastScene* scene, FastCameraProbe* out);
struct FastBuildInfo { uint32_t builder, numNodes, numTris, maxDepth; float deviceMs, wallMs; };
// builder: 0 = host binned SAH (okrt_fast_build.cpp), 1 = device PLOC (okrt_fast_build_device.hip)
bool fastBuild(FastScene** scene, const FastInputs& in, const FastDeviceTables& dev, uint32_t builder, hipStream_t stream, std::string& err);
void fastBuildInfo(const FastScene* scene, FastBuildInfo* info);
// known-answer probe: closest hit of n rays through the structure, one thread per ray (okrt_debug_kat)
bool fastKat(const FastScene* scene, const SceneView& s, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats,
             hipStream_t stream, std::string& err);
// known-answer probe through traceKernel itself (OKRT_KAT_CLOSEST_HIT_TRACE): the rays go onto a sharded queue, one launch of the
// persistent bounce-ray kernel traces them, the hit records come back in ray order
bool fastKatTrace(FastScene* scene, const SceneView& s, const okrt_render_data& rd, uint32_t n, const float* in, uint32_t inFloats, float* out,
                  uint32_t outFloats, hipStream_t stream, std::string& err);
// upper bound for the wavefront queues of one launch set (0 = none): larger sets are refused and the caller halves them
void fastSetQueueBudget(FastScene** scene, size_t bytes);
// test hook (OKRT_OPT_DEBUG_FAIL_ALLOC): the nth queue allocation from now on reports out-of-memory
void fastDebugFailAlloc(int nth);
// copies the structure back to the host (tests); either pointer may be null
bool fastReadStructure(const FastScene* scene, void* nodes, uint32_t nodeCapacity, void* tris, uint32_t triCapacity, hipStream_t stream, std::string& err);
void fastDestroy(FastScene* scene);
// One launch set through the wavefront pipeline. OutOfMemory (with err) if queue memory cannot be had.
// `counters` (optional) = full okrt_counters accounting; `rayTotals` (always) = the same buffer, of which
// only pixels/segments are bumped, by one atomic per trace launch. `events` (optional): 2*(max_bounces+1)+1 events recorded before/after every kernel, in launch order.
// `frames` >= 1 consecutive frames (numAccumulationFrames = rd's value, +1, +2, ...) are traced by ONE set of launches;
// their radiance lands in per-frame slots and is added to `accum` in frame order, so the result is bit-identical to
// `frames` single-frame calls (frames > 1 requires accumulationEnabled == 1).
enum class LaunchStatus { Ok, OutOfMemory /* queues do not fit (budget or allocator): retry with fewer frames */, Invalid, Device };
// Launch sets in flight together (OKRT_OPT_PIPELINE, up to kPipelineSlots): each uses its own queue slot and stream; what must stay ordered is the
// accumulation, so a pipelined set keeps its radiance in frame slots (also for one frame) and adds them in ONE kernel
// that waits for `after` (the previous set's accumulation) and signals `done`.
struct LaunchOrder {
    uint32_t slot = 0;         // which of the kPipelineSlots queue sets
    bool frameSlots = false;   // radiance slots + accumulateKernel even for a single frame
    hipEvent_t after = nullptr, done = nullptr;
};
LaunchStatus launchFast(FastScene& fs, const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum,
                        uint32_t* result, unsigned long long* counters, unsigned long long* rayTotals, hipStream_t stream, hipEvent_t* events,
                        uint32_t frames, const LaunchOrder& order, std::string& err);
// The device memory a launch set of `frames` frames will need on queue slot `slot`, allocated now (a deferred okrt_render
// reserves what its launch set will use, so that an allocator refusal belongs to the call that caused it).
LaunchStatus fastReserve(FastScene& fs, const SceneView& s, const TileView& t, uint32_t frames, uint32_t slot, bool frameSlots, hipStream_t stream, std::string& err);
constexpr uint32_t kMaxFramesPerLaunch = 64;
constexpr uint32_t kPipelineSlots = 4;  // launch sets that may be in flight together (queue slots = streams)
constexpr uint64_t kPipelineSplitRays = 12ull << 20;  // a larger set of more than kPipelineDefaultFrames frames is issued as sub-sets on the slots when it traces at most this many camera rays
constexpr uint32_t kPipelineMaxFrames = 32, kPipelineDefaultFrames = 4;  // OKRT_OPT_PIPELINE: launch sets up to n frames alternate between the two queue slots

}  // namespace okrt
