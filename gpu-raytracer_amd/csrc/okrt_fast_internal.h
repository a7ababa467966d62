// okrt_fast_internal.h — what the three translation units of the wavefront pipeline share (okrt_fast_pipeline.hip,
// okrt_fast_trace.hip, okrt_fast_shade.hip): queue views, tunables, the per-segment shading function, and the
// host-side launch wrappers of the kernels each unit defines. Not part of any public interface.
#pragma once
#include "okrt_fast.h"

namespace okrt {
namespace fastdetail {

// Streaming accesses (ray queues, hit records, radiance slots: written once, read once, GBs per launch set) carry the
// non-temporal hint so that they do not evict what IS reused — nodes, triangles, shading records, textures — from L2 and
// the Infinity Cache.
typedef float nt4_t __attribute__((ext_vector_type(4)));
typedef float nt2_t __attribute__((ext_vector_type(2)));
#ifdef OKRT_NO_NT
__device__ __forceinline__ float4 ldStream(const float4* p) { return *p; }
__device__ __forceinline__ float2 ldStream(const float2* p) { return *p; }
__device__ __forceinline__ uint32_t ldStream(const uint32_t* p) { return *p; }
__device__ __forceinline__ void stStream(float4* p, float4 v) { *p = v; }
__device__ __forceinline__ void stStream(float2* p, float2 v) { *p = v; }
__device__ __forceinline__ void stStream(uint32_t* p, uint32_t v) { *p = v; }
#else
__device__ __forceinline__ float4 ldStream(const float4* p) {
    const nt4_t v = __builtin_nontemporal_load(reinterpret_cast<const nt4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ldStream(const float2* p) {
    const nt2_t v = __builtin_nontemporal_load(reinterpret_cast<const nt2_t*>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ uint32_t ldStream(const uint32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stStream(float4* p, float4 v) {
    const nt4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<nt4_t*>(p));
}
__device__ __forceinline__ void stStream(float2* p, float2 v) {
    const nt2_t w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt2_t*>(p));
}
__device__ __forceinline__ void stStream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
#endif

constexpr int kTraceThreads = 256;
#ifndef OKRT_LDS_STACK
#define OKRT_LDS_STACK 10
#endif
#ifndef OKRT_PACKET_WAVES
#define OKRT_PACKET_WAVES 5
#endif
#ifndef OKRT_TRACE_WAVES
#define OKRT_TRACE_WAVES 7
#endif
constexpr int kLdsStack = OKRT_LDS_STACK;  // entries per lane kept in LDS
#ifndef OKRT_REFILL_BELOW
#define OKRT_REFILL_BELOW 40
#endif
#ifndef OKRT_NODE_MIN
#define OKRT_NODE_MIN 32
#endif
#ifndef OKRT_TRI_MIN
#define OKRT_TRI_MIN 24
#endif
#ifndef OKRT_DUAL_TRI_MIN
#define OKRT_DUAL_TRI_MIN 40
#endif
#ifndef OKRT_DUAL_SPECULATE
#define OKRT_DUAL_SPECULATE 0
#endif
#ifndef OKRT_DUAL_POLICY
#define OKRT_DUAL_POLICY 2  // which step blocks an iteration runs: 0 = lane-count thresholds, 1 = work retired per instruction issued, 2 = one block, triangles from kDualTriMin lanes
#endif
constexpr uint32_t kHold = 0xFFFFFFFEu;  // traceDualKernel without speculation: "pop the stack when the parked leaf is done"
constexpr uint32_t kDualTriMin = OKRT_DUAL_TRI_MIN;  // traceDualKernel: the triangle block runs once this many lanes hold a parked leaf
constexpr int kRefillBelow = OKRT_REFILL_BELOW;  // refill idle lanes when fewer than this many lanes are active
constexpr uint32_t kNodeMin = OKRT_NODE_MIN, kTriMin = OKRT_TRI_MIN;  // OKRT_DUAL_POLICY=0: a step block runs when at least this many lanes want it
constexpr uint32_t kDone = kEmptyRef;
constexpr uint32_t kShards = 8;      // queue shards = XCDs
constexpr uint32_t kBatch = 64;      // rays a wave reserves per cursor atomic (sweep: 64 best with 5 blocks/CU)
constexpr uint32_t kCursorStride = 32;  // words: one cursor per 128-byte line (device-scope atomics serialise per line)

// A sharded queue: shard s owns positions [s*cap, s*cap + count(s)). Camera rays: count(s) follows
// from `total`; queued rays: counts[] written by the shading kernel. cursors[] = next unreserved.
struct ShardView {
    uint32_t cap, total;
    uint32_t batch, xcdAffinity;  // tunables (developer knobs: OKRT_FAST_BATCH / _AFFINITY)
    const uint32_t* counts;  // counts[s * kCursorStride]; nullptr for camera rays
    uint32_t* cursors;       // cursors[s * kCursorStride]
    __device__ uint32_t count(uint32_t s) const {
        if (counts) return counts[s * kCursorStride];
        const uint32_t lo = s * cap;
        return total > lo ? min(total - lo, cap) : 0u;
    }
};

struct RayQueue {        // SoA, capacity = owned pixels
    float4* a;           // origin.xyz, dir.x
    float4* b;           // dir.y, dir.z, seed (bits), framebuffer offset (bits)
    float4* c;           // contribution.rgb, light.r
    float2* d;           // light.g, light.b
};
struct HitBuffer {
    float4* h;           // distance, bu, bv, hitIdx (bits)
    uint32_t* tri;       // source triangle index (kInvalid for a sphere hit or a miss)
};

struct FastView {
    const uint4* nodes;    // Node4Q as 4 x uint4
    const float4* tris;    // Tri48 as 3 x float4
    uint2* overflow;       // per-thread spill slab
    uint32_t overflowCap;  // entries per thread
    uint32_t nodeBytes, triBytes;  // sizes of the two tables (< 4 GiB each): the bounds of traceKernel's buffer descriptors
};

// Shading of one path segment from its hit record (RaytracerCS.hlsl:440-513): surface fetch, lights, environment on a
// miss, BSDF sample. Updates ray / throughput / radiance / RNG state in place; returns true when the path continues.
OKRT_DEV bool shadeSegment(const SceneView& s, const okrt_render_data& rd, Ray& ray, const Hit& h, uint32_t bounce, uint32_t lastBounce,
                           V3& contribution, V3& light, uint32_t& seed, PathCounters& cnt) {
    // Nothing to fetch for a hit that ends the path in a scene without emitters and lights (SceneView::lastSegmentDark): the
    // shading record, the texture taps and the ten draws of the BSDF sample would add +-0 to `light` and move a ray nobody traces.
    if (s.lastSegmentDark && bounce == lastBounce && h.hitIdx != kInvalid) return false;
    Surface sf = shadeFetch(s, ray, h, cnt);
    addLights(s, rd, ray, sf, bounce, contribution, light, seed);
    if (!sf.hit) {
        light = fma3(sampleEnvironment(s, ray.direction, cnt), contribution, light);  // RaytracerCS.hlsl:489-493
        return false;
    }
    scatter(sf, ray, contribution, light, seed);  // :495-513 (also on the last segment: emission is added here)
    return bounce < lastBounce;
}

// Does a triangle hit at distance d replace the current closest hit (packet kernel, known-answer probe)? Nearer wins
// (RaytracerCS.hlsl:316: strict <); among EQUAL distances the lower triangle index does, whatever the visit order: a packet tests
// the union of its rays' leaves in the packet's order, and a ray must not find another triangle because it shares its packet with
// other rays (frame-stacked packets regroup the same rays by launch-set size). The reference keeps the first one met in ITS order;
// either rule differs from it on exact ties only (tests: "up to equal-distance ties").
OKRT_DEV bool closerHit(float d, uint32_t tri, uint32_t idx, float best, uint32_t bestTri, uint32_t bestIdx) {
    return d < best || (d == best && (tri < bestTri || (tri == bestTri && idx < bestIdx)));  // (two instances of one mesh share triangle indices)
}

// Frame-stacked packets: packet number p (0 .. frames-1) of 8x8 tile `tileIdx`, lane l -> (ii = the lane's pixel as an index into
// the 8x8-tiled frame, f = its frame). Of the tile's `frames` packets every full group of EIGHT is cut as 4x2 pixels x 8 frames, every
// full group of FOUR of the rest as 4x4 pixels x 4 frames, and what is left over (frames % 4 packets) stays 8x8 pixels x 1 frame.
OKRT_DEV void stackedPacketLane(uint32_t frames, uint32_t p, uint32_t lane, uint32_t tileIdx, uint32_t& ii, uint32_t& f) {
    ii = (tileIdx << 6) + lane;
    f = p;
    if (p < (frames & ~7u)) {
        const uint32_t sb = p & 7u, lx = (lane & 3u) + 4u * (sb & 1u), ly = ((lane >> 2) & 1u) + 2u * (sb >> 1);
        ii = (tileIdx << 6) + ly * 8u + lx;
        f = (p & ~7u) + (lane >> 3);
    } else if (p < (frames & ~3u)) {
        const uint32_t qd = p & 3u, lx = (lane & 3u) + 4u * (qd & 1u), ly = ((lane >> 2) & 3u) + 4u * (qd >> 1);
        ii = (tileIdx << 6) + ly * 8u + lx;
        f = (p & ~3u) + (lane >> 4);
    }
}

// primary ray index -> pixel (8x8 tiles, row-major over tiles)
OKRT_DEV bool primaryPixel(const TileView& t, uint32_t i, uint32_t& px, uint32_t& lrow) {
    const uint32_t tilesX = (t.W + 7) >> 3;
    const uint32_t tile = i >> 6, l = i & 63;
    px = (tile % tilesX) * 8 + (l & 7);
    lrow = (tile / tilesX) * 8 + (l >> 3);
    return px < t.W && lrow < t.rows;
}

constexpr int kPacketStack = 128;  // shared entries per wave (>= 3 * tree depth + 4, checked on the host)

// FUSED: the whole camera segment in this kernel — the rays are generated here instead of being read from queue 0, and
// after the traversal the same wave shades its 64 hits and appends the surviving paths to queue 1. Against the
// generate / trace / shade kernel triple this removes 136 bytes per pixel and frame of queue and hit-record traffic and
// lets the issue-bound traversal of some waves overlap the memory-bound shading of others.
struct FusedArgs {
    uint32_t perFrame;        // queue positions per frame (tile padding included)
    RayQueue qOut;            // queue 1
    uint32_t* outCounts;      // its per-shard append counters
    uint32_t lastBounce;
    float4* accum;
    uint32_t* result;
    float4* radiance;
    const okrt_render_data* rdDev;  // the same RenderData in device memory: read where it is used instead of being held in SGPRs across the tile loop
    uint32_t terminalNext;    // 1 = the survivors' next segment is the last one of a scene where a hit there adds nothing (traceKernel MODE 1)
    // longest tiles first: tileOrder[r] = the tile of cost rank r (nullptr = frame-major order), tileCost[t] = node steps + triangle
    // tests tile t took in frame 0 of this set (what orderTilesKernel ranks the NEXT set by), frames = frames of this launch set
    const uint32_t* tileOrder;
    uint32_t* tileCost;
    uint32_t frames;
    // 1 = of a tile's `frames` packets every full group of eight is cut as 4x2 pixels x 8 consecutive frames, every full group of four
    // of the rest as 4x4 pixels x 4 frames, instead of 8x8 pixels x 1 frame (tracePacketKernel: "frame-stacked packets"); the queue
    // positions, shard counts and the tile order are the same either way
    uint32_t stackFrames;
};

// ---- launch wrappers (one per kernel family; `count` selects the variant that feeds okrt_counters) --------------------
// (stacked = 1: positions are tile-major, `frames` packets per tile cut by stackedPacketLane — what the camera-segment probe traces)
void launchGenerate(dim3 grid, hipStream_t stream, const okrt_render_data& rd, const TileView& tile, RayQueue q, uint32_t perFrame, uint32_t frames,
                    uint32_t stacked = 0);
void launchTrace(bool count, int mode, dim3 grid, hipStream_t stream, const FastView& fv, const SceneView& s, const okrt_render_data& rd, const TileView& tile,
                 RayQueue q, HitBuffer hb, const ShardView& sv, uint32_t cameraFrames, unsigned long long* counters, unsigned long long* rayTotals);
void launchPacket(bool count, bool fused, dim3 grid, hipStream_t stream, const uint4* nodes, const float4* tris, const SceneView& s,
                  const okrt_render_data& rd, const TileView& tile, RayQueue q, HitBuffer hb, const ShardView& sv, uint32_t cameraFrames,
                  unsigned long long* counters, unsigned long long* rayTotals, const FusedArgs& fa);
void launchShade(bool first, bool count, dim3 grid, hipStream_t stream, const SceneView& s, const okrt_render_data& rd, const TileView& tile,
                 RayQueue qIn, HitBuffer hb, const ShardView& in, RayQueue qOut, uint32_t* outCounts, uint32_t bounce, uint32_t lastBounce,
                 float4* accum, uint32_t* result, float4* radiance, unsigned long long* counters, uint32_t terminalNext);
// the misses of a last segment traced with traceKernel MODE 1 / 2: environment lookup into the pixel's radiance slot
void launchMiss(dim3 grid, hipStream_t stream, const SceneView& s, RayQueue qIn, const ShardView& in, float4* radiance);
// ranks the camera tiles of a launch set by the step count they took last time (descending) and writes the per-shard ray counts
// of the longest-first layout (FusedArgs::tileOrder)
void launchOrderTiles(hipStream_t stream, const uint32_t* tileCost, uint32_t tiles, uint32_t frames, uint32_t* order, uint32_t* keys, uint32_t* shardCounts);
void launchSetRenderData(hipStream_t stream, const okrt_render_data& rd, okrt_render_data* dst);
void launchAccumulate(hipStream_t stream, const okrt_render_data& rd, uint32_t pixels, uint32_t frames, const float4* radiance, float4* accum,
                      uint32_t* result);
void launchKatQueue(hipStream_t stream, bool collect, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats, RayQueue q, HitBuffer hb,
                    uint32_t cap, uint32_t* counts);
void launchFastKat(hipStream_t stream, const uint4* nodes, const float4* tris, const SceneView& s, uint32_t n, const float* in, uint32_t inFloats,
                   float* out, uint32_t outFloats);

}  // namespace fastdetail
}  // namespace okrt
