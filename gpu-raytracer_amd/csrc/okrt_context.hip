// okrt_context.hip — the C ABI of include/okrt.h over the HIP runtime: device memory for the
// reference's tables (t0..t10), framebuffers (u0/u1), validation, options, counters, timing.
// Replaces source/DirectX/DX11.cpp + source/Graphics/GPUStorage.* + RenderTexture for the compute
// path of the reference (citations in include/okrt.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "okrt_fast.h"
#include "okrt_kernels.h"
#include "okrt_launch_plan.h"

using namespace okrt;

namespace {

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

struct Table {
    std::vector<uint8_t> host;  // retained copy: validation + derivation of the fast structure
    DeviceBuffer dev;
    uint32_t count = 0, stride = 0;
};

}  // namespace

struct okrt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    Table tables[11];
    DeviceBuffer textures, env, shadeInfo;
    uint32_t numTextures = 0, texW = 0, texH = 0, envFace = 0;
    uint32_t W = 0, H = 0;
    TileView tile{};
    DeviceBuffer accumOwn, resultOwn;
    float4* accum = nullptr;
    uint32_t* result = nullptr;
    DeviceBuffer counters;
    bool countersOn = false, timingOn = false;
    int traversal = OKRT_TRAVERSAL_AUTO;  // see effectiveTraversal
    uint32_t instFlags = 0;               // OKRT_BUILD_FLAG_SCALED / _SHRUNK_INSTANCES of the uploaded t7 (recomputed at upload)
    int maxBounces = 1;
    // SceneView::lastSegmentDark = lastSegmentShortcut && materialsDark && no light in the frame's RenderData && counters off
    bool materialsDark = true;          // every Material in t6 / t7: emission colour * power == (+-0)^3 and |albedo| <= 1 (recomputed at upload)
    bool lastSegmentShortcut = true;    // OKRT_OPT_LAST_SEGMENT_SHORTCUT
    bool lastSegmentAnyHit = false;     // OKRT_OPT_LAST_SEGMENT_ANY_HIT
    bool tablesDirty = true;   // re-validate before the next render
    bool tablesValid = false;
    FastScene* fast = nullptr;  // derived structure (okrt_fast.h); rebuilt when geometry changes
    bool fastDirty = true;
    bool fastGeometryDirty = true;  // t0 / t2 changed since the last build (device builder's leaf-range cache)
    uint32_t builder = 0;           // OKRT_OPT_BUILDER
    // quirk Q3: the fast default renders shrunk instances differently from the reference; said once through
    // okrt_last_error unless the host has looked at okrt_get_build_info().flags itself
    bool buildFlagsRead = false, shrunkWarningDue = false;
    // Deferred okrt_render calls (OKRT_OPT_DEFER): consecutive frames of an unchanged scene wait here and are traced by
    // one launch set (renderImpl with frames > 1); anything that observes or changes state flushes them first.
    uint32_t pending = 0, deferMax = 16;
    bool deferReserved = false;  // the queues of a full deferred launch set are allocated (OOM belongs to the call that defers)
    uint32_t batchLimit = kMaxFramesPerLaunch;  // halved when the queues of a launch set do not fit in memory
    bool queueBudgetChosen = false;             // the host set OKRT_OPT_QUEUE_MEMORY_MB itself (else: a quarter of the free memory at okrt_resize)
    okrt_render_data pendingRd{};
    bool externalFramebuffers = false, externalStream = false;
    hipEvent_t evBegin = nullptr, evEnd = nullptr;
    // OKRT_OPT_PIPELINE: consecutive launch sets of a progressive accumulation take turns on up to four queue slots, each with
    // its own stream, so the next sets' bulk fills the GPU while the previous sets' last rays drain. Only the accumulation is
    // ordered between them (accumDone events). Every entry point but the render calls joins stream B back into `stream`.
    bool pipelineOn = true, accumChain = false;
    bool sideActive[kPipelineSlots] = {}, sideNeedsSync[kPipelineSlots] = {};  // per side stream: work since the last join / main-stream work since it last ran
    uint32_t pipelineMaxFrames = kPipelineDefaultFrames;  // launch sets up to this many frames take turns on the queue slots
    uint32_t pipelineDepth = kPipelineSlots;              // how many of the slots are used (OKRT_PIPELINE_DEPTH in the environment: 2 ... 4)
    hipStream_t side[kPipelineSlots] = {};  // side[0] is unused (slot 0 runs on `stream`)
    hipEvent_t accumDone[kPipelineSlots] = {}, evJoinA = nullptr, evJoinB = nullptr;
    uint32_t pipeNext = 0, pipeLast = 0;
    // A launch set on a SMALL tile (one rank's share of an N-way split) is issued as sub-sets that ride the pipeline slots:
    // its fixed cost — the last dependent chains of every kernel on an emptied GPU — is a fifth of its time there, and the
    // next sub-set's bulk fills the GPU meanwhile. Above splitRays rays per set the contention costs more than it returns.
    uint64_t splitRays = kPipelineSplitRays;
    uint32_t splitParts = 1;  // (round 5: with the camera tiles issued longest first the tail the sub-sets were hiding is gone: one set is faster; OKRT_PIPELINE_SPLIT_PARTS=2 restores them)
    // okrt_peek_result_rgba8: two device-side copies of the resolved picture, refreshed alternately at the end of every
    // launch set once a host has asked for one; the peek copies the newest finished one out on its own stream
    bool peekOn = false;
    DeviceBuffer front[2];
    hipEvent_t frontReady[2] = {nullptr, nullptr}, peekDone[2] = {nullptr, nullptr};
    bool frontValid[2] = {false, false}, peekRecorded[2] = {false, false};
    uint32_t frontFrames[2] = {0, 0}, frontNext = 0, launchedFrames = 0;
    hipStream_t copyStream = nullptr;
    // OKRT_OPT_TIMING: ring of per-render event lists (kernel boundaries, in launch order)
    struct TimedFrame { std::vector<hipEvent_t> ev; int used = 0; bool fast = false; uint32_t frames = 1; };
    std::vector<TimedFrame> ring;
    size_t ringNext = 0, ringCount = 0;
    std::string err;
};

static okrt_status renderImpl(okrt_ctx* c, const okrt_render_data* rd, uint32_t frames, bool subSet = false);  // below

namespace {

okrt_status fail(okrt_ctx* c, okrt_status s, const std::string& msg) {
    if (c) c->err = msg;
    return s;
}

// OKRT_TRAVERSAL_AUTO (the default): the fast pipeline wherever it shows the reference's picture; the reference-order kernel
// for a scene with a shrunk instance, where the reference drops geometry (quirk Q3) and the fast structure does not.
int effectiveTraversal(const okrt_ctx* c) {
    if (c->traversal != OKRT_TRAVERSAL_AUTO) return c->traversal;
    return (c->instFlags & OKRT_BUILD_FLAG_SHRUNK_INSTANCES) ? OKRT_TRAVERSAL_REFERENCE : OKRT_TRAVERSAL_FAST;
}

// Executes the deferred frames. Errors of the launch (queue memory) surface at the call that flushes.
// `frames` consecutive frames of the accumulation loop, as few launch sets as memory allows: a launch set that cannot
// get its queues (56+ bytes per ray and frame) is retried at half the size, and the limit is remembered.
LaunchPlanParams planParams(const okrt_ctx* c, bool batchable) {
    return LaunchPlanParams{c->tile.rows, c->W, c->batchLimit, c->pipelineMaxFrames, c->pipelineDepth, c->splitParts, c->splitRays,
                            c->pipelineOn && !c->externalStream && !c->externalFramebuffers && effectiveTraversal(c) == OKRT_TRAVERSAL_FAST, batchable};
}

okrt_status renderBatched(okrt_ctx* c, const okrt_render_data* rd, uint32_t frames, bool batchable) {
    okrt_render_data cur = *rd;
    LaunchCursor lc;
    while (frames) {
        const LaunchStep step = nextLaunch(planParams(c, batchable), lc, frames);  // okrt_launch_plan.h
        const uint32_t n = step.frames;
        const okrt_status s = renderImpl(c, &cur, n, step.subSet);
        if (s == OKRT_ERR_OOM && n > 1) {
            c->batchLimit = step.subSet ? n : n / 2;  // (a refused sub-set: whole sets of its size are issued in halves again)
            lc = LaunchCursor{};
            continue;
        }
        if (s != OKRT_OK) return s;
        if (cur.accumulationEnabled == 1) cur.numAccumulationFrames += n;  // the host's pre-increment per frame (RayTracer.cpp:489)
        frames -= n;
        launched(lc, n);
    }
    return OKRT_OK;
}

okrt_status flushPending(okrt_ctx* c) {
    if (!c->pending) return OKRT_OK;
    const uint32_t n = c->pending;
    c->pending = 0;
    return renderBatched(c, &c->pendingRd, n, true);
}
// Everything launched on the side streams so far is ordered before whatever follows on the context's main stream; the
// next launch on a side stream waits for the main stream in turn (uploads, resets, ... made there).
okrt_status joinStreams(okrt_ctx* c) {
    for (uint32_t k = 1; k < kPipelineSlots; ++k) {
        c->sideNeedsSync[k] = true;
        if (!c->sideActive[k]) continue;
        c->sideActive[k] = false;
        if (hipSetDevice(c->device) != hipSuccess || hipEventRecord(c->evJoinB, c->side[k]) != hipSuccess || hipStreamWaitEvent(c->stream, c->evJoinB, 0) != hipSuccess)
            return fail(c, OKRT_ERR_DEVICE, "pipeline: joining a side stream failed");
    }
    return OKRT_OK;
}
// every entry point that observes or changes state: deferred frames executed, both streams joined
#define OKRT_FLUSH(c)                                \
    do {                                             \
        okrt_status fs_ = flushPending(c);           \
        if (fs_ == OKRT_OK) fs_ = joinStreams(c);    \
        if (fs_ != OKRT_OK) return fs_;              \
    } while (0)
// the render calls themselves: launch sets may stay in flight on both streams
#define OKRT_FLUSH_KEEP_PIPELINE(c)                  \
    do {                                             \
        okrt_status fs_ = flushPending(c);           \
        if (fs_ != OKRT_OK) return fs_;              \
    } while (0)
#define OKRT_HIP(c, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(c, e_ == hipErrorOutOfMemory ? OKRT_ERR_OOM : OKRT_ERR_DEVICE,                       \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                  \
    } while (0)

okrt_status ensure(okrt_ctx* c, DeviceBuffer& b, size_t bytes) {
    if (b.bytes == bytes && (b.ptr || !bytes)) return OKRT_OK;
    if (b.ptr) {
        OKRT_HIP(c, hipStreamSynchronize(c->stream));
        OKRT_HIP(c, hipFree(b.ptr));
        b.ptr = nullptr;
        b.bytes = 0;
    }
    if (bytes) {
        OKRT_HIP(c, hipMalloc(&b.ptr, bytes));
        b.bytes = bytes;
    }
    return OKRT_OK;
}

uint32_t slotStride(okrt_table slot) {
    switch (slot) {
        case OKRT_TABLE_TRI_POS: return OKRT_SIZEOF_TRIANGLE;
        case OKRT_TABLE_TRI_INFO: return OKRT_SIZEOF_TRIANGLE_INFO;
        case OKRT_TABLE_BVH: return OKRT_SIZEOF_BVH_NODE;
        case OKRT_TABLE_OCT: return OKRT_SIZEOF_OCT_NODE;
        case OKRT_TABLE_SPHERE: return OKRT_SIZEOF_SPHERE;
        case OKRT_TABLE_MESH: return OKRT_SIZEOF_MESH;
        case OKRT_TABLE_DIR_LIGHT: return OKRT_SIZEOF_DIR_LIGHT;
        case OKRT_TABLE_POINT_LIGHT: return OKRT_SIZEOF_POINT_LIGHT;
        case OKRT_TABLE_SPOT_LIGHT: return OKRT_SIZEOF_SPOT_LIGHT;
        default: return 0;
    }
}

uint32_t ownedRows(uint32_t H, uint32_t stripeRows, uint32_t idx, uint32_t stride) {
    uint32_t rows = 0;
    for (uint32_t s = idx; (uint64_t)s * stripeRows < H; s += stride) rows += std::min<uint32_t>(stripeRows, H - s * stripeRows);
    return rows;
}

okrt_status allocFramebuffers(okrt_ctx* c) {
    size_t px = (size_t)c->tile.rows * c->W;
    okrt_status s;
    if ((s = ensure(c, c->accumOwn, px * sizeof(float4))) != OKRT_OK) return s;
    if ((s = ensure(c, c->resultOwn, px * 4)) != OKRT_OK) return s;
    c->accum = (float4*)c->accumOwn.ptr;
    c->result = (uint32_t*)c->resultOwn.ptr;
    c->externalFramebuffers = false;  // okrt_resize / okrt_set_tile rebind to the library's own buffers (okrt.h: re-bind after them)
    c->frontValid[0] = c->frontValid[1] = false;  // pictures of another frame size
    c->launchedFrames = 0;
    c->deferReserved = false;
    if (px) {
        OKRT_HIP(c, hipMemsetAsync(c->accum, 0, px * sizeof(float4), c->stream));
        OKRT_HIP(c, hipMemsetAsync(c->result, 0, px * 4, c->stream));
    }
    return OKRT_OK;
}

// Every index a kernel will follow is checked on the host first: an out-of-range or cyclic table
// would fault or hang the GPU. Children must come after their parent (true of every tree the
// reference's builders emit: BvhBuilder.cpp:178-186, RayTracer.cpp:690-693), which rules out cycles.
okrt_status validateTables(okrt_ctx* c) {
    const Table& tp = c->tables[OKRT_TABLE_TRI_POS];
    const Table& ti = c->tables[OKRT_TABLE_TRI_INFO];
    const Table& bv = c->tables[OKRT_TABLE_BVH];
    const Table& oc = c->tables[OKRT_TABLE_OCT];
    const Table& me = c->tables[OKRT_TABLE_MESH];
    const BvhNodeRec* nodes = (const BvhNodeRec*)bv.host.data();
    for (uint32_t i = 0; i < bv.count; ++i) {
        const BvhNodeRec& n = nodes[i];
        if (n.firstChild != kInvalid) {
            if (n.firstChild <= i || (uint64_t)n.firstChild + 1 >= bv.count) return fail(c, OKRT_ERR_RANGE, "BVH node " + std::to_string(i) + ": child index out of order or range");
        } else if (n.triStart > n.triEnd || n.triEnd > tp.count) {
            if (!(n.triStart == kInvalid && n.triEnd == kInvalid)) return fail(c, OKRT_ERR_RANGE, "BVH leaf " + std::to_string(i) + ": triangle range outside t0");
        }
    }
    if (tp.count && ti.count < tp.count) return fail(c, OKRT_ERR_RANGE, "t1 (TriangleInfo) is shorter than t0 (Triangle)");
    const MeshRec* meshes = (const MeshRec*)me.host.data();
    for (uint32_t i = 0; i < me.count; ++i)
        if (meshes[i].bvhNodeStartIdx >= bv.count) return fail(c, OKRT_ERR_RANGE, "mesh " + std::to_string(i) + ": bvhNodeStartIdx outside t2");
    const OctNodeRec* oct = (const OctNodeRec*)oc.host.data();
    for (uint32_t i = 0; i < oc.count; ++i) {
        const OctNodeRec& n = oct[i];
        if (n.meshesStart != n.meshesEnd && (n.meshesStart > n.meshesEnd || n.meshesEnd > me.count))
            return fail(c, OKRT_ERR_RANGE, "octree node " + std::to_string(i) + ": instance range outside t7");
        if (n.numChildren && (n.numChildren > 8 || n.firstChild <= i || (uint64_t)n.firstChild + n.numChildren > oc.count))
            return fail(c, OKRT_ERR_RANGE, "octree node " + std::to_string(i) + ": child range out of order or range");
    }
    return OKRT_OK;
}

// OKRT_BUILD_FLAG_*: singular values of the linear part of every instance matrix in t7 (eigenvalues of A^T A, closed
// form for a symmetric 3x3), against 1 with a tolerance that float rotations stay inside.
uint32_t instanceFlags(const okrt_ctx* c) {
    const Table& me = c->tables[OKRT_TABLE_MESH];
    const MeshRec* meshes = (const MeshRec*)me.host.data();
    uint32_t flags = 0;
    for (uint32_t i = 0; i < me.count; ++i) {
        const float* m = meshes[i].M;
        const double A[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
        double S[3][3];
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) S[r][q] = A[0][r] * A[0][q] + A[1][r] * A[1][q] + A[2][r] * A[2][q];
        const double p1 = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
        const double tr = (S[0][0] + S[1][1] + S[2][2]) / 3.0;
        double lo, hi;
        if (p1 < 1e-24) {
            lo = std::min(S[0][0], std::min(S[1][1], S[2][2]));
            hi = std::max(S[0][0], std::max(S[1][1], S[2][2]));
        } else {
            const double p2 = (S[0][0] - tr) * (S[0][0] - tr) + (S[1][1] - tr) * (S[1][1] - tr) + (S[2][2] - tr) * (S[2][2] - tr) + 2.0 * p1;
            const double p = std::sqrt(p2 / 6.0);
            double B[3][3];
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) B[r][q] = (S[r][q] - (r == q ? tr : 0.0)) / p;
            const double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                               B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
            const double phi = std::acos(std::min(1.0, std::max(-1.0, det / 2.0))) / 3.0;
            hi = tr + 2.0 * p * std::cos(phi);
            lo = tr + 2.0 * p * std::cos(phi + 2.0943951023931953);
        }
        if (!(lo > 1.0 - 1e-4 && hi < 1.0 + 1e-4)) flags |= OKRT_BUILD_FLAG_SCALED_INSTANCES;  // also catches NaN
        if (!(lo > 1.0 - 1e-4)) flags |= OKRT_BUILD_FLAG_SHRUNK_INSTANCES;
    }
    return flags;
}

// Is a hit on the last path segment worth nothing in this scene? (SceneView::lastSegmentDark has the argument.) Per material:
// the product the shader forms, emission colour * power (RaytracerCS.hlsl:152), is exactly zero in every component (NaN and
// inf * 0 fail the comparison), and |albedo| <= 1 so that the path throughput — a product of lerp(albedo, 1, metallicFactor)
// factors starting at 1 (:505) — can never leave [-1, 1]: 0 * throughput stays a signed zero. Sampled albedo lies in [0, 1].
bool materialIsDark(const Material& m) {
    for (int k = 0; k < 3; ++k) {
        const float e = m.emission[k] * m.emissionPower;
        if (!(e == 0.f) || !(std::fabs(m.albedo[k]) <= 1.f)) return false;
    }
    return true;
}
bool allMaterialsDark(const okrt_ctx* c) {
    const Table& sp = c->tables[OKRT_TABLE_SPHERE];
    const Table& me = c->tables[OKRT_TABLE_MESH];
    for (uint32_t i = 0; i < sp.count; ++i)
        if (!materialIsDark(((const SphereRec*)sp.host.data())[i].material)) return false;
    for (uint32_t i = 0; i < me.count; ++i)
        if (!materialIsDark(((const MeshRec*)me.host.data())[i].material)) return false;
    return true;
}
bool frameHasNoLights(const okrt_render_data& rd) { return (rd.numPointLights | rd.numSpotLights | rd.numDirLights) == 0u; }

SceneView makeView(const okrt_ctx* c, const okrt_render_data* rd = nullptr) {
    SceneView s{};
    s.triPos = (const TriangleRec*)c->tables[OKRT_TABLE_TRI_POS].dev.ptr;
    s.triInfo = (const TriangleInfoRec*)c->tables[OKRT_TABLE_TRI_INFO].dev.ptr;
    s.shadeInfo = (const float4*)c->shadeInfo.ptr;
    s.bvh = (const BvhNodeRec*)c->tables[OKRT_TABLE_BVH].dev.ptr;
    s.textures = (const uint8_t*)c->textures.ptr;
    s.env = (const uint8_t*)c->env.ptr;
    s.oct = (const OctNodeRec*)c->tables[OKRT_TABLE_OCT].dev.ptr;
    s.spheres = (const SphereRec*)c->tables[OKRT_TABLE_SPHERE].dev.ptr;
    s.meshes = (const MeshRec*)c->tables[OKRT_TABLE_MESH].dev.ptr;
    s.dirLights = (const DirLightRec*)c->tables[OKRT_TABLE_DIR_LIGHT].dev.ptr;
    s.pointLights = (const PointLightRec*)c->tables[OKRT_TABLE_POINT_LIGHT].dev.ptr;
    s.spotLights = (const SpotLightRec*)c->tables[OKRT_TABLE_SPOT_LIGHT].dev.ptr;
    s.numOct = c->tables[OKRT_TABLE_OCT].count;
    s.numTextures = c->numTextures;
    s.texW = c->texW;
    s.texH = c->texH;
    s.envFace = c->envFace;
    s.maxBounces = (uint32_t)c->maxBounces;
    // (with OKRT_OPT_COUNTERS every segment is shaded in full: tex_taps / mesh_hits count the reference's work)
    s.lastSegmentDark = (rd && c->lastSegmentShortcut && c->materialsDark && !c->countersOn && frameHasNoLights(*rd)) ? 1u : 0u;
    s.lastSegmentAnyHit = (s.lastSegmentDark && c->lastSegmentAnyHit) ? 1u : 0u;
    return s;
}

}  // namespace

namespace okrt {
okrt_status ctxFrame(okrt_ctx* c, CtxFrame* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->W || (!c->accum && c->tile.rows)) return fail(c, OKRT_ERR_STATE, "multi-GPU gather before okrt_resize");  // (a rank may own zero rows)
    *out = CtxFrame{c->device, c->stream, c->accum, c->tile};
    return OKRT_OK;
}
okrt_status ctxFail(okrt_ctx* c, okrt_status s, const std::string& msg) { return fail(c, s, msg); }
}  // namespace okrt

extern "C" {

int okrt_abi_version(void) { return OKRT_ABI_VERSION; }

static thread_local std::string g_createError;  // okrt_last_error(NULL): why the last okrt_create on this thread failed

okrt_status okrt_create(int device, okrt_ctx** out) {
    if (!out) return OKRT_ERR_INVALID_ARG;
    *out = nullptr;
    g_createError.clear();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n) {
        g_createError = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e)
                                        : "device " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)";
        return OKRT_ERR_DEVICE;
    }
    okrt_ctx* c = new okrt_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking)) != hipSuccess) {
        g_createError = std::string("hipSetDevice/hipStreamCreate: ") + hipGetErrorString(e);
        delete c;
        return OKRT_ERR_DEVICE;
    }
    c->stream = c->ownStream;
    hipEventCreate(&c->evBegin);
    hipEventCreate(&c->evEnd);
    if (const char* d = getenv("OKRT_PIPELINE_DEPTH")) c->pipelineDepth = std::min<uint32_t>(std::max(atoi(d), 2), kPipelineSlots);
    if (const char* d = getenv("OKRT_PIPELINE_SPLIT_RAYS")) c->splitRays = strtoull(d, nullptr, 10);
    if (const char* d = getenv("OKRT_PIPELINE_SPLIT_PARTS")) c->splitParts = std::min<uint32_t>(std::max(atoi(d), 1), kPipelineSlots);
    for (uint32_t k = 1; k < kPipelineSlots; ++k) {
        c->sideNeedsSync[k] = true;
        if (hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking) != hipSuccess) c->side[k] = nullptr, c->pipelineOn = false;
    }
    for (uint32_t k = 0; k < kPipelineSlots; ++k)
        if (hipEventCreateWithFlags(&c->accumDone[k], hipEventDisableTiming) != hipSuccess) c->pipelineOn = false;
    for (hipEvent_t* e : {&c->evJoinA, &c->evJoinB})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) c->pipelineOn = false;
    if ((e = hipMalloc(&c->counters.ptr, kNumCounters * sizeof(unsigned long long))) != hipSuccess) {
        g_createError = std::string("hipMalloc: ") + hipGetErrorString(e);
        delete c;
        return OKRT_ERR_OOM;
    }
    c->counters.bytes = kNumCounters * sizeof(unsigned long long);
    hipMemsetAsync(c->counters.ptr, 0, c->counters.bytes, c->stream);
    *out = c;
    return OKRT_OK;
}

okrt_status okrt_destroy(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    (void)flushPending(c);
    hipSetDevice(c->device);
    (void)joinStreams(c);
    hipStreamSynchronize(c->stream);
    for (uint32_t k = 1; k < kPipelineSlots; ++k)
        if (c->side[k]) {
            hipStreamSynchronize(c->side[k]);
            hipStreamDestroy(c->side[k]);
        }
    for (uint32_t k = 0; k < kPipelineSlots; ++k)
        if (c->accumDone[k]) hipEventDestroy(c->accumDone[k]);
    for (hipEvent_t e : {c->evJoinA, c->evJoinB})
        if (e) hipEventDestroy(e);
    for (Table& t : c->tables)
        if (t.dev.ptr) hipFree(t.dev.ptr);
    for (DeviceBuffer* b : {&c->textures, &c->env, &c->shadeInfo, &c->accumOwn, &c->resultOwn, &c->counters, &c->front[0], &c->front[1]})
        if (b->ptr) hipFree(b->ptr);
    if (c->copyStream) {
        hipStreamSynchronize(c->copyStream);
        hipStreamDestroy(c->copyStream);
    }
    for (hipEvent_t e : {c->frontReady[0], c->frontReady[1], c->peekDone[0], c->peekDone[1]})
        if (e) hipEventDestroy(e);
    fastDestroy(c->fast);
    for (hipEvent_t e : {c->evBegin, c->evEnd})
        if (e) hipEventDestroy(e);
    for (auto& f : c->ring)
        for (hipEvent_t e : f.ev) hipEventDestroy(e);
    if (c->ownStream) hipStreamDestroy(c->ownStream);
    delete c;
    return OKRT_OK;
}

const char* okrt_last_error(const okrt_ctx* c) { return c ? c->err.c_str() : g_createError.c_str(); }

okrt_status okrt_set_table(okrt_ctx* c, okrt_table slot, const void* data, uint32_t elem_bytes, uint32_t count) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    uint32_t stride = slotStride(slot);
    if (!stride) return fail(c, OKRT_ERR_INVALID_ARG, "set_table: unknown slot");
    if (elem_bytes != stride) return fail(c, OKRT_ERR_INVALID_ARG, "set_table: element stride " + std::to_string(elem_bytes) + " != reference stride " + std::to_string(stride));
    OKRT_HIP(c, hipSetDevice(c->device));
    Table& t = c->tables[slot];
    if (!data) count = 0;
    size_t bytes = (size_t)count * stride;
    // RayTracer::updateBuffers re-uploads the light / sphere tables every frame (RayTracer.cpp:434-492): identical
    // contents are not a change (and must not cut the deferred frames short)
    if (t.stride == stride && t.count == count && bytes <= (64u << 10) && t.host.size() == bytes && (!bytes || std::memcmp(t.host.data(), data, bytes) == 0))
        return OKRT_OK;
    OKRT_FLUSH(c);
    okrt_status s = ensure(c, t.dev, bytes);
    if (s != OKRT_OK) return s;
    t.host.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
    t.count = count;
    t.stride = stride;
    if (bytes) {
        // the previous contents may still be read by an in-flight frame on this stream: stream order
        OKRT_HIP(c, hipMemcpyAsync(t.dev.ptr, t.host.data(), bytes, hipMemcpyHostToDevice, c->stream));
    }
    if (slot == OKRT_TABLE_TRI_INFO) {  // derive the 64-byte shading records (okrt_device.h: SceneView::shadeInfo)
        std::vector<float> packed((size_t)count * 16, 0.f);
        const TriangleInfoRec* ti = (const TriangleInfoRec*)t.host.data();
        for (uint32_t i = 0; i < count; ++i) {
            float* o = &packed[(size_t)i * 16];
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 3; ++k) o[4 * v + k] = ti[i].v[v].normal[k];
            o[3] = ti[i].v[0].uv[0]; o[7] = ti[i].v[0].uv[1]; o[11] = ti[i].v[1].uv[0];
            o[12] = ti[i].v[1].uv[1]; o[13] = ti[i].v[2].uv[0]; o[14] = ti[i].v[2].uv[1];
        }
        okrt_status s2 = ensure(c, c->shadeInfo, packed.size() * sizeof(float));
        if (s2 != OKRT_OK) return s2;
        if (!packed.empty()) {
            OKRT_HIP(c, hipMemcpyAsync(c->shadeInfo.ptr, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
            OKRT_HIP(c, hipStreamSynchronize(c->stream));  // `packed` dies at return
        }
    }
    if (slot == OKRT_TABLE_SPHERE || slot == OKRT_TABLE_MESH) c->materialsDark = allMaterialsDark(c);
    if (slot == OKRT_TABLE_MESH) {
        c->instFlags = instanceFlags(c);
        c->shrunkWarningDue = (c->instFlags & OKRT_BUILD_FLAG_SHRUNK_INSTANCES) != 0;
    }
    if (slot == OKRT_TABLE_TRI_POS || slot == OKRT_TABLE_TRI_INFO || slot == OKRT_TABLE_BVH || slot == OKRT_TABLE_OCT || slot == OKRT_TABLE_MESH) {
        c->tablesDirty = true;
        c->fastDirty = true;
        if (slot == OKRT_TABLE_TRI_POS || slot == OKRT_TABLE_BVH) c->fastGeometryDirty = true;
    }
    return OKRT_OK;
}

// Row-major RGBA8 slices from the caller -> 4x4-texel tiles in `dst` (okrt_device.h: tiledOffset). The caller's buffer
// is copied by the time this returns.
static okrt_status uploadTiled(okrt_ctx* c, DeviceBuffer& dst, const uint8_t* rgba8, uint32_t slices, uint32_t width, uint32_t height) {
    const size_t linear = (size_t)slices * width * height * 4;
    const size_t tiled = (size_t)slices * ((width + 3) / 4) * ((height + 3) / 4) * 64;
    okrt_status s = ensure(c, dst, tiled);
    if (s != OKRT_OK) return s;
    void* staging = nullptr;
    OKRT_HIP(c, hipMalloc(&staging, linear));
    hipError_t e = hipMemcpyAsync(staging, rgba8, linear, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launchTileTextures(staging, dst.ptr, slices, width, height, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    hipFree(staging);
    if (e != hipSuccess) return fail(c, OKRT_ERR_DEVICE, std::string("texture upload: ") + hipGetErrorString(e));
    return OKRT_OK;
}

okrt_status okrt_set_textures(okrt_ctx* c, const uint8_t* rgba8, uint32_t count, uint32_t width, uint32_t height) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    if (!rgba8 || !count || !width || !height) {
        okrt_status s = ensure(c, c->textures, 0);
        c->numTextures = c->texW = c->texH = 0;
        return s;
    }
    okrt_status s = uploadTiled(c, c->textures, rgba8, count, width, height);
    if (s != OKRT_OK) return s;
    c->numTextures = count;
    c->texW = width;
    c->texH = height;
    return OKRT_OK;
}

okrt_status okrt_set_envmap(okrt_ctx* c, const uint8_t* faces, uint32_t face) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    if (!faces || !face) {
        okrt_status s = ensure(c, c->env, 0);
        c->envFace = 0;
        return s;
    }
    okrt_status s = uploadTiled(c, c->env, faces, 6, face, face);
    if (s != OKRT_OK) return s;
    c->envFace = face;
    return OKRT_OK;
}

okrt_status okrt_scale_texture_rgba8(okrt_ctx* c, const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh) {
    if (!c || !src || !dst || !sw || !sh || !dw || !dh) return fail(c, OKRT_ERR_INVALID_ARG, "scale_texture: bad argument");
    OKRT_HIP(c, hipSetDevice(c->device));
    void *dsrc = nullptr, *ddst = nullptr;
    size_t sb = (size_t)sw * sh * 4, db = (size_t)dw * dh * 4;
    OKRT_HIP(c, hipMalloc(&dsrc, sb));
    hipError_t e = hipMalloc(&ddst, db);
    if (e != hipSuccess) {
        hipFree(dsrc);
        return fail(c, OKRT_ERR_OOM, "scale_texture: hipMalloc");
    }
    e = hipMemcpyAsync(dsrc, src, sb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launchScaleTexture((const uint8_t*)dsrc, sw, sh, (uint8_t*)ddst, dw, dh, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dst, ddst, db, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dsrc);
    hipFree(ddst);
    if (e != hipSuccess) return fail(c, OKRT_ERR_DEVICE, std::string("scale_texture: ") + hipGetErrorString(e));
    return OKRT_OK;
}

okrt_status okrt_resize(okrt_ctx* c, uint32_t width, uint32_t height) {
    if (!c || !width || !height || width > 65536 || height > 65536) return fail(c, OKRT_ERR_INVALID_ARG, "resize: bad dimensions");
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    c->W = width;
    c->H = height;
    c->tile = TileView{width, height, height, height, 0, 1};
    okrt_status s = allocFramebuffers(c);
    if (s == OKRT_OK && !c->queueBudgetChosen) {
        // default bound for the ray queues of one launch set (okrt.h, OKRT_OPT_QUEUE_MEMORY_MB): a quarter of what is free now
        size_t freeB = 0, totalB = 0;
        if (const char* e = getenv("OKRT_QUEUE_MEMORY_ASSUME_FREE_MB")) freeB = (size_t)strtoull(e, nullptr, 10) << 20;  // test hook
        else if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = 0, (void)hipGetLastError();
        if (freeB) {
            fastSetQueueBudget(&c->fast, std::max<size_t>(freeB / 4, (size_t)1 << 20));
            c->batchLimit = kMaxFramesPerLaunch;
            c->deferReserved = false;
        }
    }
    return s;
}

okrt_status okrt_set_tile(okrt_ctx* c, uint32_t stripe_rows, uint32_t stripe_index, uint32_t stripe_stride) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->W) return fail(c, OKRT_ERR_STATE, "set_tile before resize");
    if (!stripe_rows || !stripe_stride || stripe_index >= stripe_stride) return fail(c, OKRT_ERR_INVALID_ARG, "set_tile: bad stripe");
    OKRT_HIP(c, hipSetDevice(c->device));
    c->tile = TileView{c->W, c->H, ownedRows(c->H, stripe_rows, stripe_index, stripe_stride), stripe_rows, stripe_index, stripe_stride};
    return allocFramebuffers(c);
}

okrt_status okrt_tile_rows(const okrt_ctx* c, uint32_t* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    *out = c->tile.rows;
    return OKRT_OK;
}

okrt_status okrt_bind_framebuffers(okrt_ctx* c, void* accum_dev, void* result_dev) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->W) return fail(c, OKRT_ERR_STATE, "bind_framebuffers before resize");
    c->accum = accum_dev ? (float4*)accum_dev : (float4*)c->accumOwn.ptr;
    c->result = result_dev ? (uint32_t*)result_dev : (uint32_t*)c->resultOwn.ptr;
    c->externalFramebuffers = accum_dev || result_dev;  // the caller reads them in stream order: no deferral
    return OKRT_OK;
}

okrt_status okrt_reset_accumulation(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->accum) return OKRT_OK;
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemsetAsync(c->accum, 0, (size_t)c->tile.rows * c->W * sizeof(float4), c->stream));
    return OKRT_OK;
}

static okrt_status checkRenderArgs(okrt_ctx* c, const okrt_render_data* rd) {
    if (!c->W) return fail(c, OKRT_ERR_STATE, "render before resize");
    if (rd->textureDims[0] != c->W || rd->textureDims[1] != c->H) return fail(c, OKRT_ERR_INVALID_ARG, "render: textureDims differ from the resize dimensions");
    if (rd->numSpheres > c->tables[OKRT_TABLE_SPHERE].count || rd->numDirLights > c->tables[OKRT_TABLE_DIR_LIGHT].count ||
        rd->numPointLights > c->tables[OKRT_TABLE_POINT_LIGHT].count || rd->numSpotLights > c->tables[OKRT_TABLE_SPOT_LIGHT].count)
        return fail(c, OKRT_ERR_RANGE, "render: a count in RenderData exceeds its uploaded table");
    return OKRT_OK;
}
static okrt_status ensureTables(okrt_ctx* c);
static okrt_status ensureFast(okrt_ctx* c);
static void warnShrunkInstancesOnce(okrt_ctx* c);

// One frame (RayTracer::render, RayTracer.cpp:219-260). With OKRT_OPT_DEFER (default) a frame that continues the
// progressive accumulation of an unchanged scene is not launched at once: it joins the pending frames and is traced
// together with them — by the next call that observes or changes anything, or when deferMax frames wait — with
// framebuffers bit-identical to launching every frame on its own (test_render_frames_is_bit_identical...). Arguments
// and tables are still validated here, so the call that passes bad data is the one that fails.
okrt_status okrt_render(okrt_ctx* c, const okrt_render_data* rd) {
    if (!c || !rd) return OKRT_ERR_INVALID_ARG;
    const bool defer = c->deferMax > 1 && !c->externalFramebuffers && !c->externalStream && effectiveTraversal(c) == OKRT_TRAVERSAL_FAST &&
                       rd->debugMode == 0 && rd->accumulationEnabled == 1;
    if (!defer) {
        OKRT_FLUSH_KEEP_PIPELINE(c);
        return renderImpl(c, rd, 1);
    }
    okrt_status s = checkRenderArgs(c, rd);
    if (s != OKRT_OK) return s;
    OKRT_HIP(c, hipSetDevice(c->device));
    if ((s = ensureTables(c)) != OKRT_OK || (s = ensureFast(c)) != OKRT_OK) return s;
    if (!c->deferReserved && c->tile.rows && c->pending >= 1) {
        // the host is looping (a second frame waits): what the deferred frames will need when they are finally launched is
        // allocated NOW, so that an allocator refusal is reported inside the loop, by a call that defers, and not by whichever
        // unrelated call happens to flush (smaller launch sets are tried first, like renderBatched does). A host that looks
        // at every frame never gets here and keeps single-frame queues.
        // (reserved the way the flush will launch it, renderBatched / renderImpl: a set that rides the pipeline slots — small, or
        // issued in sub-sets — keeps its radiance in frame slots and lands on whichever slot's turn it is, so every slot in use
        // gets its queues; a big set runs on slot 0)
        const uint32_t want = std::min(c->deferMax, c->batchLimit);
        uint32_t n = want;
        std::string msg;
        for (;;) {
            LaunchPlanParams pp = planParams(c, true);
            pp.batchLimit = n;
            LaunchCursor lc;
            const LaunchStep first = nextLaunch(pp, lc, n);
            const uint32_t sub = first.frames;
            const bool rides = ridesSlots(pp, first);
            LaunchStatus ls = LaunchStatus::Ok;
            for (uint32_t k = 0; k < (rides ? c->pipelineDepth : 1u) && ls == LaunchStatus::Ok; ++k)
                ls = fastReserve(*c->fast, makeView(c), c->tile, sub, k, rides || sub > 1, k ? c->side[k] : c->stream, msg);
            if (ls == LaunchStatus::Ok) break;
            if (ls != LaunchStatus::OutOfMemory || n == 1) return fail(c, ls == LaunchStatus::OutOfMemory ? OKRT_ERR_OOM : OKRT_ERR_DEVICE, "fast pipeline: " + msg);
            n /= 2;
        }
        if (n < want) c->batchLimit = n;  // (only a refusal lowers the limit: okrt_render_frames keeps its larger sets otherwise)
        c->deferReserved = true;
    }
    if (c->pending) {
        okrt_render_data expect = c->pendingRd;
        expect.numAccumulationFrames += c->pending;  // the host's pre-increment per frame (RayTracer.cpp:489)
        if (std::memcmp(&expect, rd, sizeof(expect)) != 0) OKRT_FLUSH_KEEP_PIPELINE(c);
    }
    if (!c->pending) c->pendingRd = *rd;
    if (++c->pending >= c->deferMax) return flushPending(c);
    return OKRT_OK;
}

okrt_status okrt_render_frames(okrt_ctx* c, const okrt_render_data* rd, uint32_t frames) {
    if (!c || !rd) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH_KEEP_PIPELINE(c);
    if (!frames) return OKRT_OK;
    const bool batchable = effectiveTraversal(c) == OKRT_TRAVERSAL_FAST && rd->debugMode == 0 && rd->accumulationEnabled == 1;
    return renderBatched(c, rd, frames, batchable);
}

static okrt_status ensureTables(okrt_ctx* c) {
    if (c->tablesDirty) {
        okrt_status s = validateTables(c);
        c->tablesValid = s == OKRT_OK;
        c->tablesDirty = false;
        if (s != OKRT_OK) return s;
    }
    if (!c->tablesValid) return fail(c, OKRT_ERR_RANGE, "render: the uploaded tables failed validation");
    return OKRT_OK;
}

// (Re)derives the traversal structure when a geometry / instance / octree table changed since the last build.
static okrt_status ensureFast(okrt_ctx* c) {
    if (!c->fastDirty) return OKRT_OK;
    std::string msg;
    FastInputs in;
    in.triPos = (const TriangleRec*)c->tables[OKRT_TABLE_TRI_POS].host.data();
    in.numTris = c->tables[OKRT_TABLE_TRI_POS].count;
    in.bvh = (const BvhNodeRec*)c->tables[OKRT_TABLE_BVH].host.data();
    in.numNodes = c->tables[OKRT_TABLE_BVH].count;
    in.oct = (const OctNodeRec*)c->tables[OKRT_TABLE_OCT].host.data();
    in.numOct = c->tables[OKRT_TABLE_OCT].count;
    in.meshes = (const MeshRec*)c->tables[OKRT_TABLE_MESH].host.data();
    in.numMeshes = c->tables[OKRT_TABLE_MESH].count;
    in.geometryChanged = c->fastGeometryDirty;
    FastDeviceTables dev{(const TriangleRec*)c->tables[OKRT_TABLE_TRI_POS].dev.ptr, (const MeshRec*)c->tables[OKRT_TABLE_MESH].dev.ptr};
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    if (!fastBuild(&c->fast, in, dev, c->builder, c->stream, msg)) return fail(c, OKRT_ERR_DEVICE, "fast structure: " + msg);
    c->fastDirty = false;
    c->fastGeometryDirty = false;
    c->deferReserved = false;  // (the spill slab follows the tree depth)
    return OKRT_OK;
}

// The status stays OKRT_OK; the text is there for a host that logs okrt_last_error after its first frame: the default traversal
// has chosen the (much slower) reference-order kernel for this scene, and why. Said once per upload of such an instance table,
// and not at all to a host that has read okrt_get_build_info().flags or chose a traversal itself.
static void warnShrunkInstancesOnce(okrt_ctx* c) {
    if (!c->shrunkWarningDue || c->traversal != OKRT_TRAVERSAL_AUTO) return;
    c->shrunkWarningDue = false;
    if (c->buildFlagsRead) return;
    c->err = "warning: an instance matrix shrinks its mesh (OKRT_BUILD_FLAG_SHRUNK_INSTANCES): the reference drops geometry there (its BVH cull "
             "compares a local-space box distance with the world-space best hit, RaytracerCS.hlsl:285 vs :311-318). The default traversal "
             "(OKRT_TRAVERSAL_AUTO) therefore renders this scene with the reference-order kernel, which reproduces the reference's picture "
             "exactly but is many times slower than the fast pipeline. Select OKRT_TRAVERSAL_FAST for the true world-space closest hit at full speed.";
}

okrt_status okrt_commit(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    okrt_status s = ensureTables(c);
    if (s != OKRT_OK) return s;
    return ensureFast(c);
}

okrt_status okrt_get_build_info(okrt_ctx* c, okrt_build_info* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    FastBuildInfo bi;
    fastBuildInfo(c->fast, &bi);
    out->builder = bi.builder;
    out->num_nodes = bi.numNodes;
    out->num_triangles = bi.numTris;
    out->max_depth = bi.maxDepth;
    out->device_ms = bi.deviceMs;
    out->wall_ms = bi.wallMs;
    FastCameraProbe probe;
    fastCameraProbe(c->fast, &probe);
    out->flags = c->instFlags | (c->materialsDark ? OKRT_BUILD_FLAG_DARK_MATERIALS : 0u) | (c->fast && !probe.chosePacket ? OKRT_BUILD_FLAG_CAMERA_PER_RAY : 0u);
    c->buildFlagsRead = true;
    return OKRT_OK;
}

okrt_status okrt_get_fast_structure(okrt_ctx* c, void* nodes64, uint32_t node_capacity, void* tris48, uint32_t tri_capacity) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    std::string msg;
    if (c->fastDirty || !fastReadStructure(c->fast, nodes64, node_capacity, tris48, tri_capacity, c->stream, msg))
        return fail(c, c->fastDirty ? OKRT_ERR_STATE : OKRT_ERR_INVALID_ARG, c->fastDirty ? "structure not built (okrt_commit)" : msg);
    return OKRT_OK;
}

static okrt_status renderImpl(okrt_ctx* c, const okrt_render_data* rd, uint32_t frames, bool subSet) {
    if (!c || !rd) return OKRT_ERR_INVALID_ARG;
    okrt_status as = checkRenderArgs(c, rd);
    if (as != OKRT_OK) return as;
    OKRT_HIP(c, hipSetDevice(c->device));
    okrt_status vs = ensureTables(c);
    if (vs != OKRT_OK) return vs;
    if (!c->tile.rows) return OKRT_OK;

    SceneView view = makeView(c, rd);
    unsigned long long* counters = c->countersOn ? (unsigned long long*)c->counters.ptr : nullptr;
    const bool referenceOrder = effectiveTraversal(c) == OKRT_TRAVERSAL_REFERENCE || rd->debugMode != 0;
    if (!referenceOrder) {
        okrt_status fs = ensureFast(c);
        if (fs != OKRT_OK) return fs;
    }
    warnShrunkInstancesOnce(c);
    // OKRT_OPT_TIMING: the launch set records into the ring's next slot; the slot is only claimed (ringNext / ringCount
    // advance) once the launches succeeded, so a refused or failed attempt leaves the ring as it was
    okrt_ctx::TimedFrame* tf = nullptr;
    if (c->timingOn && !c->ring.empty()) {
        tf = &c->ring[c->ringNext];
        tf->used = 0;  // its events are about to be re-recorded: not a valid measurement again until this launch set succeeded
        const size_t need = 2 * (size_t)(c->maxBounces + 1) + 1;
        while (tf->ev.size() < need) {
            hipEvent_t e;
            OKRT_HIP(c, hipEventCreate(&e));
            tf->ev.push_back(e);
        }
    }
    int used = 0;
    hipStream_t launchStream = c->stream;
    // Two launch sets in flight: a set that merely continues a progressive accumulation takes the queue slot and stream
    // the previous set did NOT use, so its camera rays fill the GPU while that set's last bounce rays drain; its one
    // accumulation kernel waits for the previous set's. Larger sets (OKRT_OPT_PIPELINE) stay on one slot: their fixed
    // cost is already amortised and two big persistent grids sharing the GPU cost more than the overlap returns.
    const bool pipelined = c->pipelineOn && !referenceOrder && !c->externalStream && !c->externalFramebuffers && rd->accumulationEnabled == 1 &&
                           (frames <= c->pipelineMaxFrames || subSet);
    // The debug heat maps count the REFERENCE traversal's box/triangle checks (RaytracerCS.hlsl:516-520),
    // so those frames always run the reference-order kernel.
    if (referenceOrder) {
        okrt_status js = joinStreams(c);
        if (js != OKRT_OK) return js;
        if (tf) OKRT_HIP(c, hipEventRecord(tf->ev[0], c->stream));
        launchReferenceOrder(view, *rd, c->tile, c->accum, c->result, counters, c->stream);
        if (tf) OKRT_HIP(c, hipEventRecord(tf->ev[1], c->stream));
        used = 2;
        c->accumChain = false;
    } else {
        std::string msg;
        LaunchOrder order;
        if (pipelined) {
            order.slot = c->pipeNext;
            order.frameSlots = true;
            order.after = c->accumChain ? c->accumDone[c->pipeLast] : nullptr;
            order.done = c->accumDone[order.slot];
            if (order.slot != 0u) launchStream = c->side[order.slot];
            // Uploads, resets, ... made on the main stream since the side streams last ran: ordered before whatever the side
            // streams do next, for ALL of them at once and BEFORE this set is enqueued. (Round 3 recorded the event when a
            // side stream was next used: by then a set on slot 0 = the main stream may already sit in front of it, and the
            // side stream's set waited for that whole set instead of running beside it.)
            bool anyNeedsSync = false;
            for (uint32_t k = 1; k < kPipelineSlots; ++k) anyNeedsSync |= c->sideNeedsSync[k] && c->side[k];
            if (anyNeedsSync) {
                OKRT_HIP(c, hipEventRecord(c->evJoinA, c->stream));
                for (uint32_t k = 1; k < kPipelineSlots; ++k)
                    if (c->sideNeedsSync[k] && c->side[k]) {
                        OKRT_HIP(c, hipStreamWaitEvent(c->side[k], c->evJoinA, 0));
                        c->sideNeedsSync[k] = false;
                    }
            }
        } else {  // accumulates from inside its kernels, on the main stream: after everything B still has in flight
            okrt_status js = joinStreams(c);
            if (js != OKRT_OK) return js;
        }
        const LaunchStatus ls = launchFast(*c->fast, view, *rd, c->tile, c->accum, c->result, counters, (unsigned long long*)c->counters.ptr, launchStream,
                                           tf ? tf->ev.data() : nullptr, frames, order, msg);
        if (ls != LaunchStatus::Ok)
            return fail(c, ls == LaunchStatus::OutOfMemory ? OKRT_ERR_OOM : ls == LaunchStatus::Invalid ? OKRT_ERR_INVALID_ARG : OKRT_ERR_DEVICE,
                        "fast pipeline: " + msg);
        used = 2 * (c->maxBounces + 1) + 1;
        if (pipelined) {
            if (order.slot != 0u) c->sideActive[order.slot] = true;
            c->pipeLast = order.slot;
            c->pipeNext = (order.slot + 1u) % c->pipelineDepth;
            c->accumChain = true;
        } else {
            c->accumChain = false;  // (stream order on the main stream covers it)
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, OKRT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
    if (tf) {
        tf->used = used;
        tf->fast = !referenceOrder;
        tf->frames = frames;
        c->ringNext = (c->ringNext + 1) % c->ring.size();
        c->ringCount = std::min(c->ringCount + 1, c->ring.size());
    }
    c->launchedFrames = rd->accumulationEnabled == 1 ? rd->numAccumulationFrames + frames - 1 : 1u;
    if (c->peekOn && !c->externalFramebuffers) {  // a copy of this launch set's resolved picture for okrt_peek_result_rgba8
        const uint32_t k = c->frontNext;
        const size_t bytes = (size_t)c->tile.rows * c->W * 4;
        if (ensure(c, c->front[k], bytes) == OKRT_OK && bytes) {
            // (a peek may still be copying the picture this buffer held two launch sets ago: device-side wait, no host stall)
            if (c->peekRecorded[k]) OKRT_HIP(c, hipStreamWaitEvent(launchStream, c->peekDone[k], 0));
            OKRT_HIP(c, hipMemcpyAsync(c->front[k].ptr, c->result, bytes, hipMemcpyDeviceToDevice, launchStream));
            OKRT_HIP(c, hipEventRecord(c->frontReady[k], launchStream));
            // the next set's accumulation (possibly on another stream) rewrites c->result: it must come after this copy too, not
            // only after this set's accumulation, or the peek could show a torn picture
            if (pipelined) OKRT_HIP(c, hipEventRecord(c->accumDone[c->pipeLast], launchStream));
            c->frontValid[k] = true;
            c->frontFrames[k] = c->launchedFrames;
            c->frontNext = k ^ 1u;
        }
    }
    return OKRT_OK;
}

okrt_status okrt_synchronize(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    return OKRT_OK;
}

okrt_status okrt_read_result_rgba8(okrt_ctx* c, uint8_t* dst) {
    if (!c || !dst) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->result) return fail(c, OKRT_ERR_STATE, "read before resize");
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemcpyAsync(dst, c->result, (size_t)c->tile.rows * c->W * 4, hipMemcpyDeviceToHost, c->stream));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    return OKRT_OK;
}

okrt_status okrt_frame_size(const okrt_ctx* c, uint32_t* out_w, uint32_t* out_h) {
    if (!c || !out_w || !out_h) return OKRT_ERR_INVALID_ARG;
    *out_w = c->W;
    *out_h = c->H;
    return OKRT_OK;
}

// The newest resolved picture without flushing deferred frames and without waiting for launch sets in flight (okrt.h).
okrt_status okrt_peek_result_rgba8(okrt_ctx* c, uint8_t* dst, uint32_t* out_frames) {
    if (!c || !dst) return OKRT_ERR_INVALID_ARG;
    if (!c->result) return fail(c, OKRT_ERR_STATE, "peek before resize");
    OKRT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->tile.rows * c->W * 4;
    if (out_frames) *out_frames = 0;
    if (!bytes) return OKRT_OK;
    if (!c->peekOn && !c->externalFramebuffers) {  // first request: from now on every launch set leaves a copy behind
        OKRT_HIP(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            OKRT_HIP(c, hipEventCreateWithFlags(&c->frontReady[k], hipEventDisableTiming));
            OKRT_HIP(c, hipEventCreateWithFlags(&c->peekDone[k], hipEventDisableTiming));
        }
        c->peekOn = true;
    }
    int pick = -1;
    for (int t = 0; t < 2 && pick < 0; ++t) {  // newest first
        const int k = (int)(c->frontNext ^ 1u) ^ t;
        if (c->peekOn && c->frontValid[k] && c->front[k].bytes == bytes && hipEventQuery(c->frontReady[k]) == hipSuccess) pick = k;
    }
    (void)hipGetLastError();  // hipEventQuery reports "not ready" through the error state
    if (pick < 0) {
        // nothing finished in a front buffer yet (first call, or right after a resize): the picture as of the last LAUNCHED
        // set, in stream order — still without executing deferred frames
        okrt_status js = joinStreams(c);
        if (js != OKRT_OK) return js;
        OKRT_HIP(c, hipMemcpyAsync(dst, c->result, bytes, hipMemcpyDeviceToHost, c->stream));
        OKRT_HIP(c, hipStreamSynchronize(c->stream));
        if (out_frames) *out_frames = c->launchedFrames;
        return OKRT_OK;
    }
    OKRT_HIP(c, hipMemcpyAsync(dst, c->front[pick].ptr, bytes, hipMemcpyDeviceToHost, c->copyStream));
    OKRT_HIP(c, hipEventRecord(c->peekDone[pick], c->copyStream));
    c->peekRecorded[pick] = true;
    OKRT_HIP(c, hipStreamSynchronize(c->copyStream));
    if (out_frames) *out_frames = c->frontFrames[pick];
    return OKRT_OK;
}

okrt_status okrt_read_accum_f32x4(okrt_ctx* c, float* dst) {
    if (!c || !dst) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->accum) return fail(c, OKRT_ERR_STATE, "read before resize");
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemcpyAsync(dst, c->accum, (size_t)c->tile.rows * c->W * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    return OKRT_OK;
}

okrt_status okrt_write_accum_f32x4(okrt_ctx* c, const float* src) {
    if (!c || !src) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->accum) return fail(c, OKRT_ERR_STATE, "write before resize");
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemcpyAsync(c->accum, src, (size_t)c->tile.rows * c->W * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    return OKRT_OK;
}

okrt_status okrt_set_option(okrt_ctx* c, okrt_option opt, int64_t value) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    switch (opt) {
        case OKRT_OPT_TRAVERSAL:
            if (value != OKRT_TRAVERSAL_REFERENCE && value != OKRT_TRAVERSAL_FAST && value != OKRT_TRAVERSAL_AUTO)
                return fail(c, OKRT_ERR_INVALID_ARG, "traversal: 0 (reference order), 1 (fast) or 2 (auto)");
            c->traversal = (int)value;
            return OKRT_OK;
        case OKRT_OPT_MAX_BOUNCES:
            if (value < 0 || value > 64) return fail(c, OKRT_ERR_INVALID_ARG, "max_bounces: 0..64");
            c->maxBounces = (int)value;
            c->deferReserved = false;  // (a third segment needs the second queue)
            return OKRT_OK;
        case OKRT_OPT_COUNTERS: c->countersOn = value != 0; return OKRT_OK;
        case OKRT_OPT_DEFER:
            if (value < 0 || value > (int64_t)kMaxFramesPerLaunch) return fail(c, OKRT_ERR_INVALID_ARG, "defer: 0 (off) .. 64 frames");
            c->deferMax = (uint32_t)value;
            c->deferReserved = false;
            return OKRT_OK;
        case OKRT_OPT_QUEUE_MEMORY_MB:
            if (value < 0) return fail(c, OKRT_ERR_INVALID_ARG, "queue memory: MiB >= 0");
            fastSetQueueBudget(&c->fast, (size_t)value << 20);
            c->queueBudgetChosen = true;
            c->batchLimit = kMaxFramesPerLaunch;  // probe again from the top
            c->deferReserved = false;
            return OKRT_OK;
        case OKRT_OPT_PIPELINE:
            if (value < 0 || value > (int64_t)kPipelineMaxFrames) return fail(c, OKRT_ERR_INVALID_ARG, "pipeline: 0 (off) .. 32 frames per launch set");
            c->pipelineOn = value != 0;
            for (uint32_t k = 1; k < kPipelineSlots; ++k) c->pipelineOn = c->pipelineOn && c->side[k] != nullptr;  // (as okrt_create decides it)
            for (uint32_t k = 0; k < kPipelineSlots; ++k) c->pipelineOn = c->pipelineOn && c->accumDone[k] != nullptr;
            c->pipelineOn = c->pipelineOn && c->evJoinA && c->evJoinB;
            if (value) c->pipelineMaxFrames = (uint32_t)value;
            c->deferReserved = false;  // (the reservation follows the launch shape: which slots a deferred set will ride)
            return OKRT_OK;
        case OKRT_OPT_LAST_SEGMENT_SHORTCUT: c->lastSegmentShortcut = value != 0; return OKRT_OK;
        case OKRT_OPT_LAST_SEGMENT_ANY_HIT: c->lastSegmentAnyHit = value != 0; return OKRT_OK;
        case OKRT_OPT_DEBUG_FAIL_ALLOC:
            if (value < 0 || value > 1000) return fail(c, OKRT_ERR_INVALID_ARG, "debug fail alloc: 0..1000");
            fastDebugFailAlloc((int)value);
            return OKRT_OK;
        case OKRT_OPT_BUILDER:
            if (value != 0 && value != 1) return fail(c, OKRT_ERR_INVALID_ARG, "builder: 0 (host SAH) or 1 (device PLOC)");
            if ((uint32_t)value != c->builder) {
                c->builder = (uint32_t)value;
                c->fastDirty = true;
                c->fastGeometryDirty = true;
            }
            return OKRT_OK;
        case OKRT_OPT_TIMING: {
            if (value < 0 || value > 4096) return fail(c, OKRT_ERR_INVALID_ARG, "timing: ring size 0..4096");
            OKRT_HIP(c, hipSetDevice(c->device));
            OKRT_HIP(c, hipStreamSynchronize(c->stream));
            c->timingOn = value != 0;
            c->ring.resize((size_t)value > c->ring.size() ? (size_t)value : c->ring.size());
            if (value) c->ring.resize((size_t)value);
            c->ringNext = c->ringCount = 0;
            return OKRT_OK;
        }
        default: return fail(c, OKRT_ERR_INVALID_ARG, "unknown option");
    }
}

okrt_status okrt_get_counters(okrt_ctx* c, okrt_counters* out) {
    if (!c || !out) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    static_assert(sizeof(okrt_counters) == kNumCounters * sizeof(uint64_t), "counter layout");
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemcpyAsync(out, c->counters.ptr, sizeof(okrt_counters), hipMemcpyDeviceToHost, c->stream));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    return OKRT_OK;
}

okrt_status okrt_reset_counters(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipMemsetAsync(c->counters.ptr, 0, c->counters.bytes, c->stream));
    return OKRT_OK;
}

okrt_status okrt_set_stream(okrt_ctx* c, void* hip_stream) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->ownStream;
    c->externalStream = hip_stream != nullptr;  // the caller orders its own work after ours on that stream: no deferral
    return OKRT_OK;
}

okrt_status okrt_timer_begin(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipEventRecord(c->evBegin, c->stream));
    return OKRT_OK;
}
okrt_status okrt_timer_end(okrt_ctx* c) {
    if (!c) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipEventRecord(c->evEnd, c->stream));
    return OKRT_OK;
}
okrt_status okrt_timer_elapsed_ms(okrt_ctx* c, float* out_ms) {
    if (!c || !out_ms) return OKRT_ERR_INVALID_ARG;
    OKRT_HIP(c, hipSetDevice(c->device));
    OKRT_HIP(c, hipEventSynchronize(c->evEnd));
    OKRT_HIP(c, hipEventElapsedTime(out_ms, c->evBegin, c->evEnd));
    return OKRT_OK;
}
static okrt_status frameStages(okrt_ctx* c, const okrt_ctx::TimedFrame& f, float out[3]) {
    out[0] = out[1] = out[2] = 0.f;
    if (f.used < 2) return OKRT_OK;
    OKRT_HIP(c, hipEventSynchronize(f.ev[f.used - 1]));
    float total = 0.f;
    OKRT_HIP(c, hipEventElapsedTime(&total, f.ev[0], f.ev[f.used - 1]));
    for (int k = 0; k + 1 < f.used; ++k) {
        float ms = 0.f;
        OKRT_HIP(c, hipEventElapsedTime(&ms, f.ev[k], f.ev[k + 1]));
        out[f.fast ? (k & 1) : 0] += ms;  // fast pipeline alternates trace, shade, trace, shade ...
    }
    out[2] = total - out[0] - out[1];
    return OKRT_OK;
}

okrt_status okrt_last_render_ms(okrt_ctx* c, float* out_ms) {
    if (!c || !out_ms) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    if (!c->timingOn || !c->ringCount) return fail(c, OKRT_ERR_STATE, "last_render_ms: enable OKRT_OPT_TIMING before rendering");
    OKRT_HIP(c, hipSetDevice(c->device));
    const okrt_ctx::TimedFrame& f = c->ring[(c->ringNext + c->ring.size() - 1) % c->ring.size()];
    float st[3];
    okrt_status s = frameStages(c, f, st);
    *out_ms = st[0] + st[1] + st[2];
    return s;
}

okrt_status okrt_get_stage_ms(okrt_ctx* c, float out_ms[3], uint32_t* out_frames) {
    if (!c || !out_ms || !out_frames) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    out_ms[0] = out_ms[1] = out_ms[2] = 0.f;
    *out_frames = 0;
    for (size_t i = 0; i < c->ringCount; ++i) {
        if (c->ring[i].used < 2) continue;  // a slot whose launch set was refused or failed
        float st[3];
        okrt_status s = frameStages(c, c->ring[i], st);
        if (s != OKRT_OK) return s;
        for (int k = 0; k < 3; ++k) out_ms[k] += st[k];
        *out_frames += c->ring[i].frames;
    }
    return OKRT_OK;
}

okrt_status okrt_get_interval_ms(okrt_ctx* c, float* out_ms, uint32_t capacity, uint32_t* out_count, uint32_t* out_launches, uint32_t* out_frames) {
    if (!c || !out_ms || !out_count || !out_launches || !out_frames) return OKRT_ERR_INVALID_ARG;
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    *out_count = *out_launches = *out_frames = 0;
    if (!c->ringCount) return OKRT_OK;
    const okrt_ctx::TimedFrame& newest = c->ring[(c->ringNext + c->ring.size() - 1) % c->ring.size()];
    if (newest.used < 2) return OKRT_OK;
    const int n = newest.used - 1;
    *out_count = (uint32_t)n;
    for (uint32_t k = 0; k < capacity && k < (uint32_t)n; ++k) out_ms[k] = 0.f;
    for (size_t i = 0; i < c->ringCount; ++i) {
        const okrt_ctx::TimedFrame& f = c->ring[i];
        if (f.used != newest.used || f.fast != newest.fast || f.frames != newest.frames) continue;  // only launch sets of the newest one's shape
        OKRT_HIP(c, hipEventSynchronize(f.ev[f.used - 1]));
        for (int k = 0; k < n; ++k) {
            float ms = 0.f;
            OKRT_HIP(c, hipEventElapsedTime(&ms, f.ev[k], f.ev[k + 1]));
            if ((uint32_t)k < capacity) out_ms[k] += ms;
        }
        *out_launches += 1;
        *out_frames += f.frames;
    }
    return OKRT_OK;
}

// Test hook (not in okrt.h; needs no GPU): the launch sets `frames` consecutive frames of a progressive accumulation are issued as on
// a context that owns tile_rows x width pixels — out[2k] = frames of set k, out[2k+1] = 1 when it rides the pipeline slots. Returns the
// number of sets (those beyond `cap` pairs are counted, not written). pipeline_max_frames = 0: pipelining off; split_rays = 0: the default.
uint32_t okrt_debug_launch_plan(uint32_t tile_rows, uint32_t width, uint32_t frames, uint32_t batch_limit, uint32_t pipeline_max_frames,
                                uint64_t split_rays, uint32_t* out, uint32_t cap) {
    const LaunchPlanParams p{tile_rows, width, std::max(1u, std::min(batch_limit, kMaxFramesPerLaunch)), pipeline_max_frames, kPipelineSlots, 2u,
                             split_rays ? split_rays : kPipelineSplitRays, pipeline_max_frames != 0, true};
    LaunchCursor lc;
    uint32_t sets = 0;
    while (frames) {
        const LaunchStep st = nextLaunch(p, lc, frames);
        if (out && sets < cap) {
            out[2 * sets] = st.frames;
            out[2 * sets + 1] = ridesSlots(p, st) ? 1u : 0u;
        }
        ++sets;
        frames -= st.frames;
        launched(lc, st.frames);
    }
    return sets;
}

okrt_status okrt_debug_kat(okrt_ctx* c, okrt_kat op, uint32_t n, const float* in, uint32_t in_floats, float* out, uint32_t out_floats) {
    if (!c || !in || !out || !in_floats || !out_floats || (int)op < 0 || (int)op > OKRT_KAT_CLOSEST_HIT_TRACE) return OKRT_ERR_INVALID_ARG;
    static const uint32_t needIn[] = {1, 1, 10, 15, 12, 7, 3, 3, 7, 7, 1, 2, 8, 8, 62, 63, 2, 7}, needOut[] = {2, 2, 1, 3, 1, 3, 3, 3, 6, 5, 4, 1, 4, 4, 7, 8, 4, 5};
    if (in_floats < needIn[op] || out_floats < needOut[op]) return fail(c, OKRT_ERR_INVALID_ARG, "kat: record too short for this probe");
    OKRT_FLUSH(c);
    OKRT_HIP(c, hipSetDevice(c->device));
    const bool fastProbe = op == OKRT_KAT_CLOSEST_HIT_FAST || op == OKRT_KAT_CLOSEST_HIT_TRACE;
    if (op == OKRT_KAT_CLOSEST_HIT || fastProbe || op == OKRT_KAT_SEGMENT_RAY) {  // walks the uploaded tables: they must be consistent
        okrt_status vs = ensureTables(c);
        if (vs == OKRT_OK && fastProbe) vs = ensureFast(c);
        if (vs != OKRT_OK) return vs;
    }
    if (!n) return OKRT_OK;
    float *dIn = nullptr, *dOut = nullptr;
    OKRT_HIP(c, hipMalloc((void**)&dIn, (size_t)n * in_floats * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dOut, (size_t)n * out_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dIn, in, (size_t)n * in_floats * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dOut, 0, (size_t)n * out_floats * sizeof(float), c->stream);
    if (e == hipSuccess) {
        std::string msg;
        if (fastProbe) {
            okrt_render_data rd{};
            std::memcpy(&rd.numSpheres, &in[6], sizeof(uint32_t));  // (the bounce kernel takes it from RenderData: uniform over the probe)
            rd.numSpheres = std::min(rd.numSpheres, c->tables[OKRT_TABLE_SPHERE].count);
            if (!(op == OKRT_KAT_CLOSEST_HIT_FAST ? fastKat(c->fast, makeView(c), n, dIn, in_floats, dOut, out_floats, c->stream, msg)
                                                  : fastKatTrace(c->fast, makeView(c), rd, n, dIn, in_floats, dOut, out_floats, c->stream, msg))) {
                hipFree(dIn);
                hipFree(dOut);
                return fail(c, OKRT_ERR_STATE, "kat: " + msg);
            }
        } else {
            launchKat(makeView(c), (uint32_t)op, n, dIn, in_floats, dOut, out_floats, c->stream);
        }
        e = hipMemcpyAsync(out, dOut, (size_t)n * out_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dIn);
    if (dOut) hipFree(dOut);
    if (e != hipSuccess) return fail(c, OKRT_ERR_DEVICE, std::string("kat: ") + hipGetErrorString(e));
    return OKRT_OK;
}

const okrt_api* okrt_get_api(void) {
    static const okrt_api api = {OKRT_ABI_VERSION,     okrt_create,        okrt_destroy,      okrt_last_error,
                                 okrt_set_table,       okrt_set_textures,  okrt_set_envmap,   okrt_scale_texture_rgba8,
                                 okrt_resize,          okrt_set_tile,      okrt_tile_rows,    okrt_reset_accumulation,
                                 okrt_render,          okrt_synchronize,   okrt_read_result_rgba8, okrt_read_accum_f32x4,
                                 okrt_set_option,      okrt_get_counters,  okrt_reset_counters, okrt_render_frames,
                                 okrt_write_accum_f32x4, okrt_bind_framebuffers, okrt_set_stream, okrt_commit,
                                 okrt_get_build_info,  okrt_get_fast_structure, okrt_timer_begin, okrt_timer_end,
                                 okrt_timer_elapsed_ms, okrt_last_render_ms, okrt_get_stage_ms, okrt_get_interval_ms,
                                 okrt_debug_kat,       okrt_comm_unique_id, okrt_comm_create, okrt_comm_create_all,
                                 okrt_comm_destroy,    okrt_comm_last_error, okrt_mgpu_gather_accum, okrt_mgpu_gather_accum_all,
                                 okrt_mgpu_read_gathered, okrt_mgpu_deinterleave,
                                 okrt_device_count,    okrt_comm_transport, okrt_mgpu_reduce_accum, okrt_mgpu_reduce_accum_all,
                                 okrt_peek_result_rgba8, okrt_frame_size,
                                 okrt_comm_exchange_ms};
    return &api;
}

}  // extern "C"
