// okrt_fast_pipeline.hip — the MI355X wavefront path tracer (OKRT_TRAVERSAL_FAST).
//
// One frame = a short pipeline of kernels over ray queues in HBM (per path segment):
//
//   generateKernel         camera rays for the owned pixels of F frames, in 8x8-tile order, into ray queue 0 (32 B per ray)
//   tracePacketKernel      camera rays: one wave walks the tree once for its 8x8 tile (wave-uniform node, scalar loads)
//   traceKernel            every other segment: persistent waves, one node visit or triangle test per lane and
//                          iteration ("if-if"), closest hit through the Node4Q/Tri48 structure, 20-byte hit record per ray
//   shadeKernel            material/texture fetch, lights, BSDF sample; finished paths go to the framebuffer (or their
//                          frame's radiance slot), survivors are compacted into the next ray queue (56 B per ray)
//   traceKernel / shadeKernel   ... one pair per further segment;  accumulateKernel when F > 1
//
// Traversal (what the profile is dominated by) is built for CDNA4:
//   * wave64, 256-thread blocks, persistent: each wave REFILLS its idle lanes when fewer than
//     kRefillBelow lanes are still traversing (ballot + prefix-popcount ranks), so divergence in path
//     length does not idle the SIMD. Rays come from kShards = 8 queue shards, one per XCD (a
//     contiguous band of the image for camera rays): a wave reserves kBatch rays at a time from its
//     own XCD's cursor (HW_REG_XCC_ID) and steals from the other shards when it runs dry. One shared
//     cursor word saturates at ~90 atomics/us on MI355X and WAS the bottleneck of the first version;
//     sharding + batching removes it and keeps each XCD's L2 on one region of the scene;
//   * per-lane traversal stack in LDS ([entry][thread], conflict-free), entries carry the child's
//     entry distance so stale entries are dropped without touching memory; deeper levels spill to
//     a per-thread HBM slab (sized from the tree depth at build time);
//   * one Node4Q visit = 4 x 16-byte loads (64 B, 8-bit child planes on a per-node grid, conservative);
//     children are visited nearest first (4-key sorting network), a Tri48 test = 3 x 16-byte loads; a lane does
//     one of the two per iteration and the loads of all lanes are issued together (see the loop body);
//   * same arithmetic as the reference for what decides the result: the ray re-normalised as the
//     shader does for each instance, Moeller-Trumbore with the
//     shader's edge convention, hit distance = length(dir*t) (RaytracerCS.hlsl:276,285,306-316),
//     so for identity instances (all benchmark scenes) the closest hit equals the reference's bit
//     for bit; only equal-distance ties may resolve to a different triangle.
// The kernels live in okrt_fast_trace.hip (traversal) and okrt_fast_shade.hip (generation, shading, accumulation);
// this file owns the device-resident state and the launch sequence.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "okrt_fast_internal.h"

namespace okrt {

using namespace fastdetail;

namespace {

// OKRT_OPT_DEBUG_FAIL_ALLOC: the n-th queue allocation from now on fails as the allocator would (test hook for the
// launch-set split; 0 = off)
// (contexts of one process may be driven from several threads — bench.py --gpus N enqueues every rank's launch set from its
// own thread — so the countdown is atomic and the arming belongs to the thread that is inside ensureQueues)
std::atomic<int> g_failAllocIn{0};
thread_local bool g_failArmed = false;  // only the allocations of a launch set count (launchFast arms it), not the structure build
struct FailAllocScope {
    FailAllocScope() { g_failArmed = true; }
    ~FailAllocScope() { g_failArmed = false; }
};

template <class T>
bool devAlloc(T** p, size_t n, std::string& err) {
    if (*p) hipFree(*p);
    *p = nullptr;
    if (!n) return true;
    const bool refuse = g_failArmed && g_failAllocIn.load(std::memory_order_relaxed) > 0 && g_failAllocIn.fetch_sub(1) == 1;
    hipError_t e = refuse ? hipErrorOutOfMemory : hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) {
        // a failed hipMalloc stays behind as the thread's "last error": taken out here, or the hipGetLastError() after the
        // NEXT (successful, smaller) launch set would report this one
        (void)hipGetLastError();
        *p = nullptr;
        err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

}  // namespace

constexpr uint32_t kMaxSegments = 66;                       // max_bounces <= 64
constexpr uint32_t kCursorBase = kMaxSegments * kShards * kCursorStride;
constexpr uint32_t kCountWords = kCursorBase + kMaxSegments * kShards * kCursorStride;
constexpr uint32_t kAllCountWords = kCountWords;

struct FastScene {
    DeviceArray<uint4> nodes;   // 4 x uint4 per Node4Q
    DeviceArray<float4> tris;   // 3 x float4 per Tri48
    uint32_t numNodes = 0, numTris = 0, maxDepth = 0;
    bool allIdentity = true;
    DeviceBuilder* builder = nullptr;  // device-side build scratch (okrt_fast_build_device.hip), made on first use
    FastBuildInfo info = {};
    // pipeline state of a launch set, sized lazily for the tile. kPipelineSlots slots: consecutive launch sets of a progressive
    // accumulation may be in flight together, each on its own stream (okrt_context.hip, OKRT_OPT_PIPELINE), each with its own
    // queues, counters, spill slab and RenderData copy.
    struct LaunchSlot {
        size_t capacity = 0, capacityQ0 = 0;  // items queue 1 + the hit records / queue 0 are allocated for
        RayQueue q[2] = {};
        HitBuffer hb = {};
        uint32_t* counts = nullptr;  // per segment b, shard s: count at (b*kShards + s)*kCursorStride, cursor at kCursorBase + the same
                                     // (every counter on its own 128-byte line: device-scope atomics serialise per line)
        uint2* overflow = nullptr;
        uint32_t overflowCap = 0;
        okrt_render_data* rdDev = nullptr;  // RenderData of the launch set in flight, for the fused camera kernel
        float4* radiance = nullptr;  // per-frame radiance slots when several frames share one launch set
        size_t radianceCap = 0;
        uint32_t* tileOrder = nullptr;  // camera tiles of this set, longest first (okrt_fast_shade.hip: orderTilesKernel)
        size_t tileOrderCap = 0;
    } slot[kPipelineSlots];
    uint32_t* tileCost = nullptr;  // steps every camera tile took in frame 0 of the most recent set (shared by the slots: a property of scene + camera)
    size_t tileCostCap = 0;
    uint32_t overflowCap = 0;  // entries per thread the spill slabs need for the current tree
    int traceBlocks = 0, cus = 0;
    // which algorithm traces the camera segment: decided once per structure from a probe of the first frame (probeCameraSegment)
    int cameraPacket = -1;      // -1 = not decided yet, 1 = one wave walks the tree for its 8x8 tile, 0 = per-ray kernel
    FastCameraProbe probe = {};
    size_t queueBudgetBytes = 0;  // OKRT_OPT_QUEUE_MEMORY_MB (0 = no limit)
};

bool fastBuild(FastScene** scene, const FastInputs& in, const FastDeviceTables& tables, uint32_t builder, hipStream_t stream, std::string& err) {
    if (!*scene) *scene = new FastScene();
    FastScene& fs = **scene;
    const auto t0 = std::chrono::steady_clock::now();
    float deviceMs = 0.f;
    if (builder == 1u) {
        if (!fs.builder) fs.builder = deviceBuilderCreate();
        FastDeviceResult r{fs.nodes, fs.tris};
        if (!buildFastDevice(*fs.builder, in, tables, stream, r, err)) return false;
        fs.numNodes = r.numNodes;
        fs.numTris = r.numTris;
        fs.maxDepth = r.maxDepth;
        fs.allIdentity = r.allIdentity;
        deviceMs = r.buildMs;
    } else {
        FastHostScene host;
        buildFastHost(in, host);
        if (!host.consistent) {
            err = "host builder: inconsistent collapse table (a subtree could not be placed)";
            return false;
        }
        fs.numNodes = (uint32_t)host.nodes.size();
        fs.numTris = (uint32_t)host.tris.size();
        fs.maxDepth = host.maxDepth;
        fs.allIdentity = host.allIdentity;
        if (!fs.nodes.need(host.nodes.size() * 4, err)) return false;
        if (!fs.tris.need(std::max<size_t>(host.tris.size(), 1) * 3, err)) return false;
        hipError_t e = hipMemcpyAsync(fs.nodes.p, host.nodes.data(), host.nodes.size() * sizeof(Node4Q), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && !host.tris.empty())
            e = hipMemcpyAsync(fs.tris.p, host.tris.data(), host.tris.size() * sizeof(Tri48), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);  // `host` dies at return
        if (e != hipSuccess) {
            err = std::string("upload: ") + hipGetErrorString(e);
            return false;
        }
    }
    if ((uint64_t)fs.numNodes * sizeof(Node4Q) > 0xFFFFFFFFull || (uint64_t)fs.numTris * sizeof(Tri48) > 0xFFFFFFFFull) {
        err = "traversal structure: a table of 4 GiB or more (the bounce-ray kernel addresses nodes and triangles with 32-bit byte offsets)";
        return false;
    }
    fs.info = {builder, fs.numNodes, fs.numTris, fs.maxDepth, deviceMs,
               std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count()};
    // spill slab: a visit pushes at most 3 entries, so 3*depth bounds the stack
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    fs.cus = cus;
    fs.traceBlocks = cus * (getenv("OKRT_FAST_BLOCKS_PER_CU") ? atoi(getenv("OKRT_FAST_BLOCKS_PER_CU")) : OKRT_TRACE_WAVES);
    fs.cameraPacket = -1;
    fs.probe = {};
    uint32_t need = 3 * fs.maxDepth + 4;
    fs.overflowCap = need > (uint32_t)kLdsStack ? need - kLdsStack : 1;  // (the slabs themselves are made by the launch set that needs them)
    return true;
}

bool fastKat(const FastScene* fs, const SceneView& s, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats, hipStream_t stream,
             std::string& err) {
    if (!fs || !fs->nodes.p) {
        err = "no structure built yet";
        return false;
    }
    if (4 * fs->maxDepth + 4 > 256u) {
        err = "tree too deep for the probe's private stack";
        return false;
    }
    launchFastKat(stream, fs->nodes.p, fs->tris.p, s, n, in, inFloats, out, outFloats);
    return true;
}

bool fastKatTrace(FastScene* fs, const SceneView& s, const okrt_render_data& rd, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats,
                  hipStream_t stream, std::string& err) {
    if (!fs || !fs->nodes.p) {
        err = "no structure built yet";
        return false;
    }
    // its own small buffers (a probe must not disturb the launch slots): a / b arrays of the queue, hit records, one segment's
    // shard counters and cursors, the spill slab, the two ray totals
    const uint32_t cap = (((n + kShards - 1) / kShards + 255) / 256) * 256;
    const size_t items = (size_t)cap * kShards;
    RayQueue q{};
    HitBuffer hb{};
    uint32_t* counts = nullptr;
    uint2* overflow = nullptr;
    unsigned long long* totals = nullptr;
    auto release = [&]() {
        for (void* p : {(void*)q.a, (void*)q.b, (void*)hb.h, (void*)hb.tri, (void*)counts, (void*)overflow, (void*)totals})
            if (p) hipFree(p);
    };
    const uint32_t countWords = 2 * kShards * kCursorStride;
    bool ok = devAlloc(&q.a, items, err) && devAlloc(&q.b, items, err) && devAlloc(&hb.h, items, err) && devAlloc(&hb.tri, items, err) &&
              devAlloc(&counts, countWords, err) && devAlloc(&overflow, (size_t)fs->traceBlocks * kTraceThreads * fs->overflowCap, err) && devAlloc(&totals, 2, err);
    hipError_t e = hipSuccess;
    if (ok) {
        e = hipMemsetAsync(counts, 0, countWords * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(hb.h, 0xFF, items * sizeof(float4), stream);
        if (e == hipSuccess) e = hipMemsetAsync(hb.tri, 0xFF, items * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemsetAsync(totals, 0, 2 * sizeof(unsigned long long), stream);
    }
    if (ok && e == hipSuccess) {
        launchKatQueue(stream, false, n, in, inFloats, out, outFloats, q, hb, cap, counts);
        const FastView fv{fs->nodes.p, fs->tris.p, overflow, fs->overflowCap, (uint32_t)(fs->numNodes * sizeof(Node4Q)), (uint32_t)(fs->numTris * sizeof(Tri48))};
        const ShardView sv{cap, 0u, kBatch, 1u, counts, counts + kShards * kCursorStride};
        launchTrace(false, 0, dim3(fs->traceBlocks), stream, fv, s, rd, TileView{}, q, hb, sv, 0u, nullptr, totals);
        launchKatQueue(stream, true, n, in, inFloats, out, outFloats, q, hb, cap, counts);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);  // the buffers are released below
    }
    release();
    if (ok && e != hipSuccess) {
        err = std::string("trace probe: ") + hipGetErrorString(e);
        ok = false;
    }
    return ok;
}

void fastDebugFailAlloc(int nth) { g_failAllocIn.store(nth); }

void fastSetQueueBudget(FastScene** scene, size_t bytes) {
    if (!*scene) *scene = new FastScene();
    (*scene)->queueBudgetBytes = bytes;
}

void fastBuildInfo(const FastScene* fs, FastBuildInfo* info) {
    if (fs) *info = fs->info;
    else std::memset(info, 0, sizeof(*info));
}

bool fastReadStructure(const FastScene* fs, void* nodes, uint32_t nodeCapacity, void* tris, uint32_t triCapacity, hipStream_t stream, std::string& err) {
    if (!fs) {
        err = "no structure built yet";
        return false;
    }
    if ((nodes && nodeCapacity < fs->numNodes) || (tris && triCapacity < fs->numTris)) {
        err = "capacity too small";
        return false;
    }
    hipError_t e = hipSuccess;
    if (nodes && fs->numNodes) e = hipMemcpyAsync(nodes, fs->nodes.p, (size_t)fs->numNodes * sizeof(Node4Q), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && tris && fs->numTris) e = hipMemcpyAsync(tris, fs->tris.p, (size_t)fs->numTris * sizeof(Tri48), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        err = std::string("read structure: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

void fastDestroy(FastScene* fs) {
    if (!fs) return;
    fs->nodes.release();
    fs->tris.release();
    if (fs->builder) deviceBuilderDestroy(fs->builder);
    for (FastScene::LaunchSlot& ls : fs->slot) {
        for (void* p : {(void*)ls.rdDev, (void*)ls.counts, (void*)ls.overflow, (void*)ls.hb.h, (void*)ls.hb.tri, (void*)ls.radiance, (void*)ls.tileOrder})
            if (p) hipFree(p);
        for (RayQueue& q : ls.q)
            for (void* p : {(void*)q.a, (void*)q.b, (void*)q.c, (void*)q.d})
                if (p) hipFree(p);
    }
    if (fs->tileCost) hipFree(fs->tileCost);
    delete fs;
}

// ---- camera segment: packet or per-ray, per scene ------------------------------------------------------------------------
// The packet kernel walks the UNION of its 64 rays' nodes and leaves once per wave; the per-ray kernel walks every ray's own
// path at 40-60 % lane utilisation and pays the generate / hit-record / shade round trip through HBM. Which is cheaper depends
// on how far a packet's rays spread over the scene's triangles (profiles/r05_packet_vs_per_ray.txt). With 8x8-pixel packets of
// one frame the packet won 1.5x on the Sponza-class scene and LOST 1.9x on 8 M sub-pixel triangles (a tile's union: 105 triangle
// tests against 5 per ray); with the frame-stacked packets a launch set is cut into (4x2 pixels x 8 frames: 28 tests) it wins on
// all five workloads, by 1.3x on that one. So the first launch set after a build PROBES: about eight 8-row stripes spread over the
// FULL frame (not the context's own tile: every rank of a split decides alike, and like a single-GPU context), EIGHT frames of
// them cut into stacked packets (generateKernel, stacked order), are traced by the counting variants of both kernels, and a cost
// model in VALU wave-instructions per packet of 64 rays — the unit both kernels are bound by (roofline.kernels[].valu_issue_frac)
// — picks one:
//     packet   = 130 per node step + 95 per triangle test                        (counted per wave)
//     per ray  = (125 per node visit + 95 per triangle test) / 0.5 lanes + 1500  (counted per lane; 1500 = the unfused segment's
//                generate, hit-record and separate shading traffic, from the same table)
// The per-ray kernel is taken when its cost is below 0.7 of the packet's: a margin, because the model leans towards the per-ray
// kernel where it matters (model / measured per-ray : packet on dragon8m, dragon, dragon80k, glass, sponza: 0.98 / 1.31,
// 1.85 / 1.73, 2.8 / 1.66, 4.5 / 1.49, 2.1 / 1.74). One choice per scene, made for the stacked shape (every launch set of eight
// frames or more is mostly that; okrt_render defers and batches 16 by default): switching kernels with the set size would let
// equal-distance ties fall differently from set to set. The counts are deterministic, so the choice is; OKRT_FAST_PACKET=0/1 in
// the environment overrides it. Closest hits agree bit for bit between the two kernels (same arithmetic per lane); only
// equal-distance ties may pick another triangle.
namespace {
bool cameraUsesPacket(const FastScene& fs) {
    static const int envPacket = getenv("OKRT_FAST_PACKET") ? atoi(getenv("OKRT_FAST_PACKET")) : -1;
    if (3 * fs.maxDepth + 4 > (uint32_t)kPacketStack) return false;
    if (envPacket >= 0) return envPacket != 0;
    return fs.cameraPacket != 0;  // (undecided = packet: what a launch set without a probe uses)
}

bool probeCameraSegment(FastScene& fs, const SceneView& s, const okrt_render_data& rd, hipStream_t stream, std::string& err) {
    fs.cameraPacket = 1;
    fs.probe = {};
    if (3 * fs.maxDepth + 4 > (uint32_t)kPacketStack || !fs.numTris) return true;  // (nothing to choose / nothing to trace)
    const uint32_t W = rd.textureDims[0], H = rd.textureDims[1];
    if (!W || !H) return true;
    const uint32_t stripes = (H + 7) / 8, stride = std::max(1u, stripes / 8u);
    TileView pt{W, H, 0, 8, stride / 2, stride};
    for (uint32_t k = pt.stripeIndex; k * 8 < H; k += stride) pt.rows += std::min(8u, H - k * 8);
    if (!pt.rows) return true;
    const uint32_t perFrame = (((W + 7) >> 3) * ((pt.rows + 7) >> 3)) << 6;
    constexpr uint32_t kProbeFrames = 8;  // eight frames of the stripes, as packets of 4x2 pixels x 8 frames: what a launch set is mostly cut into
    const uint32_t positions = perFrame * kProbeFrames;
    const uint32_t cap = (((positions + kShards - 1) / kShards + 255) / 256) * 256;
    const size_t items = (size_t)cap * kShards;
    RayQueue q{};
    HitBuffer hb{};
    uint32_t* counts = nullptr;
    uint2* overflow = nullptr;
    unsigned long long* ctr = nullptr;  // two okrt_counters-shaped blocks: packet, per ray
    auto release = [&]() {
        for (void* p : {(void*)q.a, (void*)q.b, (void*)hb.h, (void*)hb.tri, (void*)counts, (void*)overflow, (void*)ctr})
            if (p) hipFree(p);
    };
    const uint32_t countWords = 2 * kShards * kCursorStride;
    bool ok = devAlloc(&q.a, items, err) && devAlloc(&q.b, items, err) && devAlloc(&hb.h, items, err) && devAlloc(&hb.tri, items, err) &&
              devAlloc(&counts, countWords, err) && devAlloc(&overflow, (size_t)fs.traceBlocks * kTraceThreads * fs.overflowCap, err) &&
              devAlloc(&ctr, 2 * kNumCounters, err);
    unsigned long long host[2 * kNumCounters] = {};
    hipError_t e = hipSuccess;
    if (ok) {
        e = hipMemsetAsync(ctr, 0, sizeof(host), stream);
        if (e == hipSuccess) e = hipMemsetAsync(counts, 0, countWords * sizeof(uint32_t), stream);
    }
    if (ok && e == hipSuccess) {
        okrt_render_data prd = rd;
        const FastView fv{fs.nodes.p, fs.tris.p, overflow, fs.overflowCap, (uint32_t)(fs.numNodes * sizeof(Node4Q)), (uint32_t)(fs.numTris * sizeof(Tri48))};
        const ShardView sv{cap, positions, kBatch, 1u, nullptr, counts + kShards * kCursorStride};
        launchGenerate(dim3((unsigned)(items / 256)), stream, prd, pt, q, perFrame, kProbeFrames, 1u);
        launchPacket(true, false, dim3((unsigned)(fs.cus * 6)), stream, fs.nodes.p, fs.tris.p, s, prd, pt, q, hb, sv, kProbeFrames, ctr, ctr, FusedArgs{});
        e = hipMemsetAsync(counts, 0, countWords * sizeof(uint32_t), stream);  // the cursors again, for the second walk over the same rays
        launchTrace(true, 0, dim3(fs.traceBlocks), stream, fv, s, prd, pt, q, hb, sv, kProbeFrames, ctr + kNumCounters, ctr + kNumCounters);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host, ctr, sizeof(host), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    release();
    if (!ok) return false;
    if (e != hipSuccess) {
        err = std::string("camera-segment probe: ") + hipGetErrorString(e);
        return false;
    }
    const double tiles = positions / 64.0, rays = (double)pt.rows * W * kProbeFrames;  // ("tile" = one packet of 64 rays)
    FastCameraProbe& pr = fs.probe;
    pr.tiles = positions / 64;
    pr.packetNodeStepsPerTile = (float)(host[10] / 64.0 / tiles);
    pr.packetTriTestsPerTile = (float)(host[11] / 64.0 / tiles);
    pr.rayNodeVisitsPerRay = (float)(host[kNumCounters + 10] / rays);
    pr.rayTriTestsPerRay = (float)(host[kNumCounters + 11] / rays);
    pr.packetCost = 130.f * pr.packetNodeStepsPerTile + 95.f * pr.packetTriTestsPerTile;
    pr.perRayCost = (125.f * pr.rayNodeVisitsPerRay + 95.f * pr.rayTriTestsPerRay) / 0.5f + 1500.f;
    fs.cameraPacket = pr.perRayCost < 0.7f * pr.packetCost ? 0 : 1;  // (a margin: on the 8 M-triangle scene the model says 0.98, the clock 1.31)
    pr.chosePacket = (uint32_t)fs.cameraPacket;
    if (getenv("OKRT_FAST_PROBE_PRINT"))  // developer knob: the numbers behind the choice
        fprintf(stderr, "[okrt] camera-segment probe: %u tiles; packet %.2f node steps + %.2f triangle tests per tile = %.0f; per ray %.2f node visits + %.2f "
                        "triangle tests = %.0f -> %s\n", pr.tiles, pr.packetNodeStepsPerTile, pr.packetTriTestsPerTile, pr.packetCost, pr.rayNodeVisitsPerRay,
                pr.rayTriTestsPerRay, pr.perRayCost, fs.cameraPacket ? "packet" : "per ray");
    return true;
}
}  // namespace

void fastCameraProbe(const FastScene* fs, FastCameraProbe* out) {
    if (fs) *out = fs->probe, out->chosePacket = cameraUsesPacket(*fs) ? 1u : 0u;
    else std::memset(out, 0, sizeof(*out));
}

// Everything a launch set of `frames` frames needs in device memory, on queue slot `slot`: ray queues, hit records, radiance
// slots, counters, spill slab, RenderData copy. Called by launchFast, and ahead of time by fastReserve so that a deferred
// okrt_render learns about an allocator refusal when it is issued, not when its frames are finally launched.
namespace {
struct QueuePlan {
    uint32_t perFrame, total, cap, lastBounce;
    size_t items, pixels;
    float4* radiance;
};
LaunchStatus ensureQueues(FastScene& fs, FastScene::LaunchSlot& ls, const SceneView& s, const TileView& t, uint32_t frames, bool frameSlots, hipStream_t stream,
                          QueuePlan& qp, std::string& err) {
    qp.perFrame = (((t.W + 7) >> 3) * ((t.rows + 7) >> 3)) << 6;  // camera rays of one frame incl. tile padding
    if (frames < 1 || frames > kMaxFramesPerLaunch || (uint64_t)qp.perFrame * frames > 0x7FFFFFFFull) {
        err = "frames per launch out of range";
        return LaunchStatus::Invalid;
    }
    const FailAllocScope failScope;
    qp.total = qp.perFrame * frames;
    // a shard holds every 8th tile (by cost rank) with all its frames: ceil(tiles / 8) x frames x 64 positions, >= total / 8
    const uint32_t tilesPerFrame = qp.perFrame >> 6;
    qp.cap = ((((tilesPerFrame + kShards - 1) / kShards) * frames * 64u + 255u) / 256u) * 256u;
    qp.items = (size_t)qp.cap * kShards;
    const size_t items = qp.items;
    // Queue 1 (56 B per item) and the hit records (20 B) are always needed. Queue 0 only exists when something uses it:
    // the unfused camera segment (generate -> trace -> shade) or a third segment (queues ping-pong from b = 2 on); with
    // the fused camera kernel and NUM_BOUNCES = 1 — the reference's configuration — it is never allocated: 76 instead of
    // 132 bytes per ray and frame in flight. Per pixel and frame of a multi-frame set: a 16-byte radiance slot.
    static const bool envFused0 = !getenv("OKRT_FAST_FUSED") || atoi(getenv("OKRT_FAST_FUSED")) != 0;
    const bool needQ0 = !(cameraUsesPacket(fs) && envFused0) || s.maxBounces >= 2;
    const size_t needBytes = items * (needQ0 ? 2 * 56 + 20 : 56 + 20) + (frames > 1 || frameSlots ? (size_t)t.rows * t.W * frames * 16 : 0);
    if (fs.queueBudgetBytes && needBytes > fs.queueBudgetBytes && frames > 1) {
        err = "launch set exceeds the queue memory budget";
        return LaunchStatus::OutOfMemory;
    }
    if (items > ls.capacity) {
        hipStreamSynchronize(stream);
        ls.capacity = ls.capacityQ0 = 0;  // the old buffers are released first: a failed allocation must not leave a stale capacity behind
        for (RayQueue* q : {&ls.q[0], &ls.q[1]})
            for (void** p : {(void**)&q->a, (void**)&q->b, (void**)&q->c, (void**)&q->d}) {
                if (*p) hipFree(*p);
                *p = nullptr;
            }
        RayQueue& q = ls.q[1];
        if (!devAlloc(&q.a, items, err) || !devAlloc(&q.b, items, err) || !devAlloc(&q.c, items, err) || !devAlloc(&q.d, items, err)) return LaunchStatus::OutOfMemory;
        if (!devAlloc(&ls.hb.h, items, err) || !devAlloc(&ls.hb.tri, items, err)) return LaunchStatus::OutOfMemory;
        ls.capacity = items;
    }
    if (needQ0 && items > ls.capacityQ0) {
        hipStreamSynchronize(stream);
        ls.capacityQ0 = 0;
        RayQueue& q = ls.q[0];
        if (!devAlloc(&q.a, ls.capacity, err) || !devAlloc(&q.b, ls.capacity, err) || !devAlloc(&q.c, ls.capacity, err) || !devAlloc(&q.d, ls.capacity, err))
            return LaunchStatus::OutOfMemory;
        ls.capacityQ0 = ls.capacity;
    }
    qp.pixels = (size_t)t.rows * t.W;
    qp.radiance = nullptr;
    if (frames > 1 || frameSlots) {  // (a set that shares the GPU with its neighbours adds its radiance in a later, ordered step too)
        if (qp.pixels * frames > ls.radianceCap) {
            hipStreamSynchronize(stream);
            ls.radianceCap = 0;
            if (!devAlloc(&ls.radiance, qp.pixels * frames, err)) return LaunchStatus::OutOfMemory;
            ls.radianceCap = qp.pixels * frames;
        }
        qp.radiance = ls.radiance;
    }
    qp.lastBounce = s.maxBounces;
    if (qp.lastBounce + 2 > kMaxSegments) {
        err = "max_bounces too large for the fast pipeline";
        return LaunchStatus::Invalid;
    }
    if (tilesPerFrame > ls.tileOrderCap) {
        hipStreamSynchronize(stream);
        ls.tileOrderCap = 0;
        if (!devAlloc(&ls.tileOrder, 2 * (size_t)tilesPerFrame, err)) return LaunchStatus::OutOfMemory;  // the order + the sort's keys
        ls.tileOrderCap = tilesPerFrame;
    }
    if (tilesPerFrame != fs.tileCostCap) {  // another tile size: the recorded costs belong to other tiles (rare: resize / set_tile)
        hipDeviceSynchronize();               // (other slots' streams may still read the old array)
        fs.tileCostCap = 0;
        if (!devAlloc(&fs.tileCost, tilesPerFrame, err)) return LaunchStatus::OutOfMemory;
        if (hipMemsetAsync(fs.tileCost, 0, (size_t)tilesPerFrame * sizeof(uint32_t), stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            err = "hipMemsetAsync(tile costs)";
            return LaunchStatus::Device;
        }
        fs.tileCostCap = tilesPerFrame;
    }
    if (!ls.rdDev && !devAlloc(&ls.rdDev, 1, err)) return LaunchStatus::OutOfMemory;
    if (!ls.counts && !devAlloc(&ls.counts, kAllCountWords, err)) return LaunchStatus::OutOfMemory;
    if (!ls.overflow || ls.overflowCap != fs.overflowCap) {  // per-thread spill slab of this slot's trace kernels, sized for the current tree
        hipStreamSynchronize(stream);
        ls.overflowCap = 0;
        if (!devAlloc(&ls.overflow, (size_t)fs.traceBlocks * kTraceThreads * fs.overflowCap, err)) return LaunchStatus::OutOfMemory;
        ls.overflowCap = fs.overflowCap;
    }
    return LaunchStatus::Ok;
}
}  // namespace

LaunchStatus fastReserve(FastScene& fs, const SceneView& s, const TileView& t, uint32_t frames, uint32_t slot, bool frameSlots, hipStream_t stream, std::string& err) {
    QueuePlan qp;
    return ensureQueues(fs, fs.slot[slot % kPipelineSlots], s, t, frames, frameSlots, stream, qp, err);
}

LaunchStatus launchFast(FastScene& fs, const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum, uint32_t* result,
                        unsigned long long* counters, unsigned long long* rayTotals, hipStream_t stream, hipEvent_t* events, uint32_t frames,
                        const LaunchOrder& order, std::string& err) {
    FastScene::LaunchSlot& ls = fs.slot[order.slot % kPipelineSlots];
    if (fs.cameraPacket < 0 && !probeCameraSegment(fs, s, rd, stream, err)) return LaunchStatus::OutOfMemory;
    QueuePlan qp;
    const LaunchStatus qs = ensureQueues(fs, ls, s, t, frames, order.frameSlots, stream, qp, err);
    if (qs != LaunchStatus::Ok) return qs;
    const uint32_t perFrame = qp.perFrame, total = qp.total, cap = qp.cap, lastBounce = qp.lastBounce;
    const size_t items = qp.items, pixels = qp.pixels;
    float4* const radiance = qp.radiance;
    hipError_t he = hipSuccess;
    auto hipFailed = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return false;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return true;
    };
    if (hipFailed(hipMemsetAsync(ls.counts, 0, (size_t)kAllCountWords * sizeof(uint32_t), stream), "hipMemsetAsync(queue counters)")) return LaunchStatus::Device;
    FastView fv{fs.nodes.p, fs.tris.p, ls.overflow, ls.overflowCap, (uint32_t)(fs.numNodes * sizeof(Node4Q)), (uint32_t)(fs.numTris * sizeof(Tri48))};
    const dim3 tg(fs.traceBlocks);
    const dim3 sg((unsigned)(items / 256));
    const bool cnt = counters != nullptr;
    int ev = 0;
    auto mark = [&]() {
        if (events && he == hipSuccess) he = hipEventRecord(events[ev], stream);
        ev++;
    };
    static const uint32_t envBatch = getenv("OKRT_FAST_BATCH") ? (uint32_t)atoi(getenv("OKRT_FAST_BATCH")) : kBatch;
    static const uint32_t envAffinity = getenv("OKRT_FAST_AFFINITY") ? (uint32_t)atoi(getenv("OKRT_FAST_AFFINITY")) : 1u;
    // (the refill threshold and the two step-block thresholds are compile-time constants of the trace kernel:
    // -DOKRT_REFILL_BELOW / -DOKRT_NODE_MIN / -DOKRT_TRI_MIN with tools/build_variant_all.sh; three fewer SGPRs live in its loop)
    auto shards = [&](uint32_t b, bool camera) {
        return ShardView{cap, camera ? total : 0u, envBatch, envAffinity, camera ? nullptr : ls.counts + b * kShards * kCursorStride,
                         ls.counts + kCursorBase + b * kShards * kCursorStride};
    };
    mark();
    static const bool envFused = !getenv("OKRT_FAST_FUSED") || atoi(getenv("OKRT_FAST_FUSED")) != 0;
    const bool packet = cameraUsesPacket(fs);  // camera rays: one wave walks the tree for its tile
    static const int envPacketBlocks = getenv("OKRT_FAST_PACKET_BLOCKS") ? atoi(getenv("OKRT_FAST_PACKET_BLOCKS")) : 6;
    const dim3 pg((unsigned)(fs.cus * envPacketBlocks));  // (2 KB LDS: more blocks per CU fit, but the sweep is flat from 6 to 16)
    // the launch set's RenderData in device memory, written by a kernel from its by-value argument: stream-ordered after
    // the previous launch set's readers, and no pageable-memory staging on the host side of an asynchronous call
    launchSetRenderData(stream, rd, ls.rdDev);
    // The last segment of a scene in which a hit there adds nothing (SceneView::lastSegmentDark, radiance kept in frame slots):
    // the kernel that appends its rays settles the hits in advance, traceKernel stores hit-or-miss, missKernel shades the misses.
    const bool terminal = s.lastSegmentDark && lastBounce >= 1 && radiance != nullptr && !cnt;
    const int terminalMode = s.lastSegmentAnyHit ? 2 : 1;
    // camera tiles longest first (fused packet kernel only; OKRT_FAST_LPT=0 keeps the frame-major order)
    static const bool envLpt = !getenv("OKRT_FAST_LPT") || atoi(getenv("OKRT_FAST_LPT")) != 0;
    const bool lpt = envLpt && packet && envFused;
    // frame-stacked camera packets (tracePacketKernel; OKRT_FAST_STACK=0 keeps every packet one 8x8 tile of one frame)
    static const bool envStack = !getenv("OKRT_FAST_STACK") || atoi(getenv("OKRT_FAST_STACK")) != 0;
    const uint32_t stackDepth = (lpt && envStack && frames >= 4u) ? 1u : 0u;
    const uint32_t tilesPerFrame = perFrame >> 6;
    if (lpt) launchOrderTiles(stream, fs.tileCost, tilesPerFrame, frames, ls.tileOrder, ls.tileOrder + tilesPerFrame, ls.counts);
    const FusedArgs fa{perFrame, ls.q[1], ls.counts + kShards * kCursorStride, lastBounce, accum, result, radiance, ls.rdDev, (terminal && lastBounce == 1) ? 1u : 0u,
                       lpt ? ls.tileOrder : nullptr, lpt ? fs.tileCost : nullptr, frames, stackDepth};
    const ShardView cameraShards = lpt ? ShardView{cap, 0u, envBatch, envAffinity, ls.counts, ls.counts + kCursorBase} : shards(0, true);
    const uint32_t pixels32 = (uint32_t)pixels;
    // the one step of a launch set that touches the accumulation: after the previous set's (another stream may carry it)
    auto orderedAccumulate = [&]() {
        if (order.after && he == hipSuccess) he = hipStreamWaitEvent(stream, order.after, 0);
        launchAccumulate(stream, rd, pixels32, frames, radiance, accum, result);
        if (order.done && he == hipSuccess) he = hipEventRecord(order.done, stream);
    };
    if (packet && envFused) {
        // segment 0 in ONE kernel: camera rays are generated, traced in packets of 64 (frame-stacked where the set has the frames) and shaded by the same wave
        launchPacket(cnt, true, pg, stream, fs.nodes.p, fs.tris.p, s, rd, t, ls.q[0], ls.hb, cameraShards, frames, counters, rayTotals, fa);
        mark();  // (an empty "shade" interval keeps the trace / shade alternation of the event list)
    } else {
        // segment 0 as three kernels: camera rays into q[0] (a, b only), trace, shade
        launchGenerate(sg, stream, rd, t, ls.q[0], perFrame, frames);
        if (packet) launchPacket(cnt, false, pg, stream, fs.nodes.p, fs.tris.p, s, rd, t, ls.q[0], ls.hb, shards(0, true), frames, counters, rayTotals, fa);
        else launchTrace(cnt, 0, tg, stream, fv, s, rd, t, ls.q[0], ls.hb, shards(0, true), frames, counters, rayTotals);
        mark();
        launchShade(true, cnt, sg, stream, s, rd, t, ls.q[0], ls.hb, shards(0, true), ls.q[1], ls.counts + kShards * kCursorStride, 0u, lastBounce, accum, result,
                    radiance, counters, fa.terminalNext);
    }
    if (lastBounce == 0 && radiance) orderedAccumulate();
    mark();
    // segments 1..lastBounce: queue b lives in q[b & 1]
    for (uint32_t b = 1; b <= lastBounce; ++b) {
        RayQueue& qin = ls.q[b & 1];
        RayQueue& qout = ls.q[(b + 1) & 1];
        const bool last = terminal && b == lastBounce;
        launchTrace(cnt, last ? terminalMode : 0, tg, stream, fv, s, rd, t, qin, ls.hb, shards(b, false), 0u, counters, rayTotals);
        mark();
        if (last) launchMiss(sg, stream, s, qin, shards(b, false), radiance);
        else launchShade(false, cnt, sg, stream, s, rd, t, qin, ls.hb, shards(b, false), qout, ls.counts + (b + 1) * kShards * kCursorStride, b, lastBounce, accum,
                         result, radiance, counters, (terminal && b + 1 == lastBounce) ? 1u : 0u);
        if (b == lastBounce && radiance) orderedAccumulate();
        mark();
    }
    if (hipFailed(he, "hipEventRecord") || hipFailed(hipGetLastError(), "kernel launch")) return LaunchStatus::Device;
    return LaunchStatus::Ok;
}

}  // namespace okrt
