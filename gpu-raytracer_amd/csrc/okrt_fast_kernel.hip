// okrt_fast_kernel.hip — the MI355X wavefront path tracer (OKRT_TRAVERSAL_FAST).
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
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "okrt_fast.h"

namespace okrt {

namespace {

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
constexpr int kRefillBelow = 40;     // refill idle lanes when fewer than this many lanes are active
constexpr uint32_t kDone = kEmptyRef;
constexpr uint32_t kShards = 8;      // queue shards = XCDs
constexpr uint32_t kBatch = 64;      // rays a wave reserves per cursor atomic (sweep: 64 best with 5 blocks/CU)
constexpr uint32_t kCursorStride = 32;  // words: one cursor per 128-byte line (device-scope atomics serialise per line)

// A sharded queue: shard s owns positions [s*cap, s*cap + count(s)). Camera rays: count(s) follows
// from `total`; queued rays: counts[] written by the shading kernel. cursors[] = next unreserved.
struct ShardView {
    uint32_t cap, total;
    uint32_t batch, xcdAffinity, refillBelow;  // tunables (developer knobs: OKRT_FAST_BATCH / _AFFINITY / _REFILL)
    uint32_t nodeMin, triMin;                  // a step block runs when at least this many lanes want it (OKRT_FAST_NODE_MIN / _TRI_MIN)
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
};

// Shading of one path segment from its hit record (RaytracerCS.hlsl:440-513): surface fetch, lights, environment on a
// miss, BSDF sample. Updates ray / throughput / radiance / RNG state in place; returns true when the path continues.
OKRT_DEV bool shadeSegment(const SceneView& s, const okrt_render_data& rd, Ray& ray, const Hit& h, uint32_t bounce, uint32_t lastBounce,
                           V3& contribution, V3& light, uint32_t& seed, PathCounters& cnt) {
    Surface sf = shadeFetch(s, ray, h, cnt);
    addLights(s, rd, ray, sf, bounce, contribution, light, seed);
    if (!sf.hit) {
        light = fma3(sampleEnvironment(s, ray.direction, cnt), contribution, light);  // RaytracerCS.hlsl:489-493
        return false;
    }
    scatter(sf, ray, contribution, light, seed);  // :495-513 (also on the last segment: emission is added here)
    return bounce < lastBounce;
}

// primary ray index -> pixel (8x8 tiles, row-major over tiles)
OKRT_DEV bool primaryPixel(const TileView& t, uint32_t i, uint32_t& px, uint32_t& lrow) {
    const uint32_t tilesX = (t.W + 7) >> 3;
    const uint32_t tile = i >> 6, l = i & 63;
    px = (tile % tilesX) * 8 + (l & 7);
    lrow = (tile / tilesX) * 8 + (l >> 3);
    return px < t.W && lrow < t.rows;
}

// Camera rays for every owned pixel, in tile order (position i <-> pixel by primaryPixel): origin, direction,
// RNG state after the 4 camera draws, framebuffer offset. Padding positions (outside the image) get pix = kInvalid.
// With `frames` > 1 the positions of frame f follow those of frame f-1 and pix addresses the frame's radiance slot.
__global__ __launch_bounds__(256) void generateKernel(okrt_render_data rd, TileView tile, RayQueue q, uint32_t perFrame, uint32_t frames) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= perFrame * frames) return;
    const uint32_t f = i / perFrame, ii = i - f * perFrame;
    uint32_t px, lrow, seed = 0, pix = kInvalid;
    Ray ray;
    ray.origin = ray.direction = mk(0.f, 0.f, 0.f);
    if (primaryPixel(tile, ii, px, lrow)) {
        ray = primaryRay(rd, rd.numAccumulationFrames + f, px, tile.globalRow(lrow), seed);
        pix = f * (tile.rows * tile.W) + lrow * tile.W + px;
    }
    stStream(&q.a[i], make_float4(ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x));
    stStream(&q.b[i], make_float4(ray.direction.y, ray.direction.z, __uint_as_float(seed), __uint_as_float(pix)));
}

template <bool COUNT>
__global__ __launch_bounds__(kTraceThreads) __attribute__((amdgpu_waves_per_eu(OKRT_TRACE_WAVES, OKRT_TRACE_WAVES))) void traceKernel(FastView fv, SceneView s, okrt_render_data rd, TileView tile, RayQueue q,
                                                            HitBuffer hb, ShardView sv, uint32_t cameraFrames,
                                                            unsigned long long* counters, unsigned long long* rayTotals) {
    // (ref, entry distance) pairs, [entry][thread]; row kLdsStack is a dummy target for predicated-off pushes
    __shared__ uint2 sStack[kLdsStack + 1][kTraceThreads];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint2* myOverflow = fv.overflow + (size_t)(blockIdx.x * kTraceThreads + tid) * fv.overflowCap;
    if (!COUNT && blockIdx.x == 0 && tid == 0) {  // rays of this launch, always counted (okrt_counters.pixels / .segments)
        unsigned long long n = 0;
        if (cameraFrames) n = (unsigned long long)tile.W * tile.rows * cameraFrames;
        else for (uint32_t k = 0; k < kShards; ++k) n += sv.count(k);
        if (cameraFrames) atomicAdd(&rayTotals[0], n);
        atomicAdd(&rayTotals[1], n);
    }
    // wave-uniform work-fetch state
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t shard = sv.xcdAffinity ? (xcc & (kShards - 1)) : 0u, tried = 0, resNext = 0, resEnd = 0;
    uint32_t preRaw = 0, preShard = 0;  // prefetched reservation: preRaw is only meaningful in lane 0
    bool havePre = false;

    uint32_t rayIdx = kInvalid;   // the ray this lane is tracing
    uint32_t cur = kDone;         // node/leaf reference being processed, kDone = nothing
    uint32_t triJ = 0;            // cur is a leaf: the triangle to test next
    uint32_t sp = 0;
    V3 ro = mk(0, 0, 0), rdir = mk(0, 0, 0), iq = mk(0, 0, 0);
    float best = kFltMax, hbu = 0.f, hbv = 0.f;
    uint32_t hitIdx = kInvalid, hitTri = kInvalid;
    uint32_t nNodes = 0, nTris = 0, nSeg = 0;
    bool exhausted = false;

#ifdef OKRT_STAMPS
    unsigned long long tRefill = 0, tInner = 0, tLeaf = 0, tStart = __builtin_amdgcn_s_memtime(), nRefill = 0, nIter = 0, laneIters = 0;
    unsigned long long nodeWave = 0, nodeLane = 0, leafWave = 0, leafLane = 0;  // wave-iterations and active lane-iterations of the two step blocks
#define OKRT_T0 unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#define OKRT_T1(acc) acc += __builtin_amdgcn_s_memtime() - t0_;
#else
#define OKRT_T0
#define OKRT_T1(acc)
#endif
    // pop the next entry the current best does not already beat
    auto popNext = [&]() -> uint32_t {
        while (sp > 0) {
            --sp;
            const uint2 e = sp < (uint32_t)kLdsStack ? sStack[sp][tid] : myOverflow[sp - kLdsStack];
            if (__uint_as_float(e.y) < best) return e.x;
        }
        return kDone;
    };
    for (;;) {
#ifdef OKRT_STAMPS
        nIter++;
        laneIters += __popcll(__ballot(cur != kDone));
#endif
        // ---- refill: lanes whose traversal ended store their hit and take the next rays --------
        const unsigned long long idle = __ballot(cur == kDone);
        if ((uint32_t)__popcll(~idle) < sv.refillBelow) {
            OKRT_T0
#ifdef OKRT_STAMPS
            nRefill++;
#endif
            if (cur == kDone && rayIdx != kInvalid) {
                stStream(&hb.h[rayIdx], make_float4(best, hbu, hbv, __uint_as_float(hitIdx)));
                stStream(&hb.tri[rayIdx], hitTri);
                rayIdx = kInvalid;
            }
            for (int round = 0; round < 2 && !exhausted; ++round) {
                const unsigned long long need = __ballot(cur == kDone);
                if (!need) break;
                if (resNext == resEnd) {  // take the prefetched reservation (wave-uniform); steal when a shard is dry
                    while (tried < kShards) {
                        if (!havePre) {
                            if (lane == 0) preRaw = atomicAdd(&sv.cursors[shard * kCursorStride], sv.batch ? sv.batch : (uint32_t)__popcll(need));
                            preShard = shard;
                            havePre = true;
                        }
                        const uint32_t bpos = __builtin_amdgcn_readfirstlane(preRaw);  // waits for the atomic here
                        const uint32_t cnt = sv.count(preShard);
                        havePre = false;
                        if (bpos < cnt) {
                            resNext = preShard * sv.cap + bpos;
                            resEnd = preShard * sv.cap + min(bpos + (sv.batch ? sv.batch : (uint32_t)__popcll(need)), cnt);
                            tried = 0;
                            break;
                        }
                        // this shard is dry: look at all cursors with ONE load (lane k reads shard k) and jump to the
                        // next shard that still has rays, instead of probing each with a returning atomic
                        uint32_t seen = 0xFFFFFFFFu, lim = 0;
                        if (lane < kShards) {
                            seen = __hip_atomic_load(&sv.cursors[lane * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            lim = sv.count(lane);
                        }
                        const uint32_t live = (uint32_t)__ballot(seen < lim) & ((1u << kShards) - 1u);
                        if (!live) {
                            tried = kShards;
                            break;
                        }
                        const uint32_t rot = ((live >> shard) | (live << (kShards - shard))) & ((1u << kShards) - 1u);  // bit j = shard+j
                        shard = (shard + (uint32_t)__builtin_ctz(rot)) & (kShards - 1);
                        tried++;
                    }
                    if (resNext == resEnd) {
                        exhausted = true;
                        break;
                    }
                    if (sv.batch) {  // issue the NEXT reservation now; its round trip overlaps the traversal of this batch
                        if (lane == 0) preRaw = atomicAdd(&sv.cursors[shard * kCursorStride], sv.batch);
                        preShard = shard;
                        havePre = true;
                    }
                }
                const uint32_t avail = resEnd - resNext;
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (cur == kDone && rank < avail) {
                    const uint32_t mine = resNext + rank;
                    const float4 A = ldStream(&q.a[mine]), B = ldStream(&q.b[mine]);
                    Ray ray;
                    ray.origin = mk(A.x, A.y, A.z);
                    ray.direction = mk(A.w, B.x, B.y);
                    if (__float_as_uint(B.w) != kInvalid) {  // camera-ray padding outside the image carries pix = kInvalid
                        rayIdx = mine;
                        nSeg++;
                        ro = ray.origin;
                        // the shader re-normalises the direction per instance (RaytracerCS.hlsl:276)
                        rdir = normalize(ray.direction);
                        iq = rcp3(rdir);  // clamped to +-1e30 so that 0 * inf cannot poison the fused slab arithmetic
                        iq = mk(fminf(fmaxf(iq.x, -1e30f), 1e30f), fminf(fmaxf(iq.y, -1e30f), 1e30f), fminf(fmaxf(iq.z, -1e30f), 1e30f));
                        best = kFltMax;
                        hitIdx = kInvalid;
                        hitTri = kInvalid;
                        hbu = hbv = 0.f;
                        for (uint32_t i = 0; i < rd.numSpheres; i++) {  // RaytracerCS.hlsl:232-242 (original direction)
                            float d = rayAndSphere(ray, ld3(s.spheres[i].position), s.spheres[i].radius);
                            if (d > 0.f && d < best) {
                                best = d;
                                hitIdx = i;
                            }
                        }
                        sp = 0;
                        cur = 0;  // root node
                    }
                }
                resNext += min((uint32_t)__popcll(need), avail);
            }
            OKRT_T1(tRefill)
            if (__ballot(rayIdx != kInvalid) == 0ull) break;  // queues drained and every lane idle
        }

        // ---- one step per lane: a node visit OR a triangle test ("if-if" scheduling) ------------------------------
        // The first version descended in a while-while loop (all node steps, then all leaf triangles); with bounce
        // rays the lanes of a wave need very different numbers of steps before their next leaf and waited for the
        // slowest one. Here every lane does exactly one unit of work per iteration, whichever it needs; a block is
        // skipped when fewer than nodeMin / triMin lanes want it and the other block has work (those lanes wait one
        // round and join a fuller block). All loads of the iteration are issued together, through one pointer per
        // lane, before either block computes.
        const bool isNode = cur != kDone && !(cur & kLeafBit);
        const bool isTri = cur != kDone && (cur & kLeafBit);
        const uint32_t wantNode = (uint32_t)__popcll(__ballot(isNode)), wantTri = (uint32_t)__popcll(__ballot(isTri));
        bool runNode = wantNode >= sv.nodeMin, runTri = wantTri >= sv.triMin;
        if (!runNode && !runTri) {
            runNode = wantNode >= wantTri && wantNode;
            runTri = !runNode && wantTri;
        }
        const bool stepNode = isNode && runNode, stepTri = isTri && runTri;
        uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0, d2 = d0, d3 = d0;
        if (stepNode || stepTri) {
            const uint4* p = stepNode ? fv.nodes + (size_t)cur * 4 : (const uint4*)fv.tris + (size_t)((cur & 0x07FFFFFFu) + triJ) * 3;
            d0 = p[0];
            d1 = p[1];
            d2 = p[2];
            if (stepNode) d3 = p[3];
        }
        if (stepNode) {
            OKRT_T0
#ifdef OKRT_STAMPS
            nodeLane++;
            if ((int)lane == __ffsll((long long)__ballot(true)) - 1) nodeWave++;
#endif
            const uint4 n0 = d0, n1 = d1, n2 = d2, n3 = d3;
            if (COUNT) nNodes++;
            float tn[4];
            uint32_t ch[4] = {n1.x, n1.y, n1.z, n1.w};
            // slab test on the node's grid: t = (origin + q*step - o)/d = fma(q, step/d, (origin - o)/d)
            const float Ax = __uint_as_float((n0.w & 255u) << 23) * iq.x, Bx = (__uint_as_float(n0.x) - ro.x) * iq.x;
            const float Ay = __uint_as_float(((n0.w >> 8) & 255u) << 23) * iq.y, By = (__uint_as_float(n0.y) - ro.y) * iq.y;
            const float Az = __uint_as_float(((n0.w >> 16) & 255u) << 23) * iq.z, Bz = (__uint_as_float(n0.z) - ro.z) * iq.z;
            // entry planes are the lower ones for a positive direction component, the upper ones otherwise
            const uint32_t nearX = iq.x >= 0.f ? n2.x : n3.x, farX = iq.x >= 0.f ? n3.x : n2.x;
            const uint32_t nearY = iq.y >= 0.f ? n2.y : n3.y, farY = iq.y >= 0.f ? n3.y : n2.y;
            const uint32_t nearZ = iq.z >= 0.f ? n2.z : n3.z, farZ = iq.z >= 0.f ? n3.z : n2.z;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float tnear = fmaxf(fmaxf(fma_((float)((nearX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((nearY >> (8 * k)) & 255u), Ay, By)),
                                          fma_((float)((nearZ >> (8 * k)) & 255u), Az, Bz));
                const float tfar = fminf(fminf(fma_((float)((farX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((farY >> (8 * k)) & 255u), Ay, By)),
                                         fma_((float)((farZ >> (8 * k)) & 255u), Az, Bz));
                bool hit = tfar >= tnear && tfar > 0.f && tnear < best && ch[k] != kEmptyRef;
                tn[k] = hit ? tnear : kFltMax;  // kFltMax doubles as "miss" (never < best)
            }
            // sort the four (distance, child) pairs ascending: 5-comparator network, select-based (no branches)
#define OKRT_CSWAP(i, j)                                              \
    {                                                                 \
        const bool sw_ = tn[j] < tn[i];                               \
        const float lo_ = sw_ ? tn[j] : tn[i], hi_ = sw_ ? tn[i] : tn[j];     \
        const uint32_t cl_ = sw_ ? ch[j] : ch[i], chh_ = sw_ ? ch[i] : ch[j]; \
        tn[i] = lo_; tn[j] = hi_; ch[i] = cl_; ch[j] = chh_;          \
    }
            OKRT_CSWAP(0, 1) OKRT_CSWAP(2, 3) OKRT_CSWAP(0, 2) OKRT_CSWAP(1, 3) OKRT_CSWAP(1, 2)
#undef OKRT_CSWAP
            // hits are now ch[0..n-1]; push n-1 far ones so that ch[1] ends on top, continue with ch[0]
            const uint32_t n = (tn[0] < kFltMax) + (tn[1] < kFltMax) + (tn[2] < kFltMax) + (tn[3] < kFltMax);
            if (sp + 3 <= (uint32_t)kLdsStack) {  // fast path: three unconditional 8-byte LDS writes, misses go to the dummy row
                const uint32_t top = sp + n;      // entry k (1..n-1) lands at sp + n-1-k
                sStack[1 < n ? top - 2 : kLdsStack][tid] = make_uint2(ch[1], __float_as_uint(tn[1]));
                sStack[2 < n ? top - 3 : kLdsStack][tid] = make_uint2(ch[2], __float_as_uint(tn[2]));
                sStack[3 < n ? top - 4 : kLdsStack][tid] = make_uint2(ch[3], __float_as_uint(tn[3]));
                sp += n > 1 ? n - 1 : 0;
            } else {  // deep in the tree: spill to the per-thread HBM slab
#pragma unroll
                for (int k = 3; k >= 1; --k) {
                    if (tn[k] < kFltMax) {
                        if (sp < (uint32_t)kLdsStack) {
                            sStack[sp][tid] = make_uint2(ch[k], __float_as_uint(tn[k]));
                        } else if (sp - kLdsStack < fv.overflowCap) {
                            myOverflow[sp - kLdsStack] = make_uint2(ch[k], __float_as_uint(tn[k]));
                        } else {
                            continue;  // cannot happen: the slab is sized from the tree depth
                        }
                        sp++;
                    }
                }
            }
            cur = n ? ch[0] : popNext();
            triJ = 0;
            OKRT_T1(tInner)
        }
        if (stepTri) {  // Moeller-Trumbore on triangle triJ of the leaf
            OKRT_T0
#ifdef OKRT_STAMPS
            leafLane++;
            if ((int)lane == __ffsll((long long)__ballot(true)) - 1) leafWave++;
#endif
            if (COUNT) nTris++;
            float bu = 0.f, bv = 0.f;
            const float t = rayAndTriangleEdges(ro, rdir, mk(__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z)),
                                                mk(__uint_as_float(d1.x), __uint_as_float(d1.y), __uint_as_float(d1.z)),
                                                mk(__uint_as_float(d2.x), __uint_as_float(d2.y), __uint_as_float(d2.z)), bu, bv);
            if (t > 0.f) {
                const float d = length(rdir * t);  // RaytracerCS.hlsl:311-314 with M = identity (world-space triangles)
                if (d < best) {
                    best = d;
                    hbu = bu;
                    hbv = bv;
                    hitTri = d0.w;
                    hitIdx = d1.w;
                }
            }
            if (++triJ > ((cur >> 27) & 15u)) {  // count - 1 is stored
                cur = popNext();
                triJ = 0;
            }
            OKRT_T1(tLeaf)
        }
    }

#ifdef OKRT_STAMPS
    for (int off = 32; off > 0; off >>= 1) {
        nodeWave += __shfl_down(nodeWave, off, 64); nodeLane += __shfl_down(nodeLane, off, 64);
        leafWave += __shfl_down(leafWave, off, 64); leafLane += __shfl_down(leafLane, off, 64);
    }
    if (COUNT && counters && lane == 0) {  // cycle sums per wave: n_oct=refill, n_mesh=inner, n_node=leaf, n_tri=total, mesh_hits=#refills, sphere_hits=#iters, tex_taps=active lane-iters
#ifdef OKRT_STAMPS_ITERS  // outer iterations and live lanes per iteration instead of the refill / leaf cycle sums
        atomicAdd(&counters[2], laneIters); atomicAdd(&counters[3], tInner); atomicAdd(&counters[4], nIter);
#else
        atomicAdd(&counters[2], tRefill); atomicAdd(&counters[3], tInner); atomicAdd(&counters[4], tLeaf);
#endif
        atomicAdd(&counters[5], __builtin_amdgcn_s_memtime() - tStart);
        atomicAdd(&counters[6], nodeWave); atomicAdd(&counters[7], nodeLane); atomicAdd(&counters[8], leafWave); atomicAdd(&counters[9], leafLane);
    }
#endif
    if (COUNT && counters) {
        uint32_t vals[3] = {nSeg, nNodes, nTris};
        const int slot[3] = {1, 10, 11};  // segments, fast_nodes, fast_tris
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint32_t v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&counters[slot[k]], (unsigned long long)v);
        }
    }
}

// ---- packet traversal of camera rays --------------------------------------------------------------------------------
// Camera rays arrive one 8x8 pixel tile per 64 consecutive positions, i.e. one tile per wave, and the rays of a tile
// walk almost the same nodes. So the wave traverses the tree ONCE for all its lanes: the current node is wave-uniform,
// its 64 bytes come through the scalar cache into SGPRs (s_load, no vector-memory gather at all), every lane tests its
// own ray against the four child boxes, a child is visited when ANY lane enters it (`__ballot`), order = entry
// distance of the first lane that enters, one shared stack of node references in LDS, no divergent control flow.
// Each lane keeps its own closest hit with the same arithmetic as the per-ray kernel, so results agree bit for bit
// (ties aside). Used for segment 0 only; bounce rays are incoherent and stay on the per-ray kernel.
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
};
template <bool COUNT, bool FUSED>
__global__ __launch_bounds__(kTraceThreads) __attribute__((amdgpu_waves_per_eu(OKRT_PACKET_WAVES))) void tracePacketKernel(const uint4* __restrict__ nodes, const float4* __restrict__ tris,
                                                                  SceneView s, okrt_render_data rd, TileView tile, RayQueue q, HitBuffer hb,
                                                                  ShardView sv, uint32_t cameraFrames, unsigned long long* counters,
                                                                  unsigned long long* rayTotals, FusedArgs fa) {
    __shared__ uint32_t sShared[kTraceThreads / 64][kPacketStack];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* stack = sShared[wave];
    PathCounters shadeCnt = {};
    uint32_t nPixels = 0;
    if (!COUNT && blockIdx.x == 0 && tid == 0) {
        const unsigned long long n = (unsigned long long)tile.W * tile.rows * cameraFrames;
        atomicAdd(&rayTotals[0], n);
        atomicAdd(&rayTotals[1], n);
    }
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t shard = sv.xcdAffinity ? (xcc & (kShards - 1)) : 0u, tried = 0;
    uint32_t nNodes = 0, nTris = 0, nSeg = 0;

    // Tile fetch, software-pipelined two deep: the reservation (a returning atomic on the shard cursor, ~1.5 us) for the
    // tile after next is in flight while the next tile's rays are being loaded and the current tile is traversed.
    uint32_t preRaw = 0, preShard = 0;  // preRaw is only meaningful in lane 0
    bool havePre = false;
    auto acquire = [&]() -> uint32_t {  // wave-uniform: first queue position of the next tile, kInvalid when all shards are dry
        while (tried < kShards) {
            if (!havePre) {
                if (lane == 0) preRaw = atomicAdd(&sv.cursors[shard * kCursorStride], 64u);
                preShard = shard;
            }
            havePre = false;
            const uint32_t bpos = __builtin_amdgcn_readfirstlane(preRaw);  // waits for the atomic here
            const uint32_t cnt = sv.count(preShard);
            if (bpos < cnt) {
                tried = 0;
                if (lane == 0) preRaw = atomicAdd(&sv.cursors[shard * kCursorStride], 64u);  // the one after: consumed a tile later
                preShard = shard;
                havePre = true;
                return preShard * sv.cap + bpos;
            }
            uint32_t seen = 0xFFFFFFFFu, lim = 0;
            if (lane < kShards) {
                seen = __hip_atomic_load(&sv.cursors[lane * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lim = sv.count(lane);
            }
            const uint32_t live = (uint32_t)__ballot(seen < lim) & ((1u << kShards) - 1u);
            if (!live) {
                tried = kShards;
                break;
            }
            const uint32_t rot = ((live >> shard) | (live << (kShards - shard))) & ((1u << kShards) - 1u);
            shard = (shard + (uint32_t)__builtin_ctz(rot)) & (kShards - 1);
            tried++;
        }
        return kInvalid;
    };
    uint32_t base = acquire();
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A;
    if (!FUSED && base != kInvalid) {
        A = ldStream(&q.a[base + lane]);
        B = ldStream(&q.b[base + lane]);
    }
    uint32_t nextBase = kInvalid;
    float4 nA = A, nB = B;
#ifdef OKRT_STAMPS
    unsigned long long pT0 = __builtin_amdgcn_s_memtime(), pSetup = 0, pNode = 0, pLeaf = 0, pTiles = 0, pNodeSteps = 0, pLeafTris = 0, pMark = pT0;
#define OKRT_PMARK(acc) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - pMark; pMark = n_; }
#else
#define OKRT_PMARK(acc)
#endif
    for (; base != kInvalid; base = nextBase, A = nA, B = nB) {  // (the hand-over sits at the END of an iteration: no early wait)
        const uint32_t mine = base + lane;
        nextBase = acquire();
        if (!FUSED && nextBase != kInvalid) {  // in flight during the traversal below
            nA = ldStream(&q.a[nextBase + lane]);
            nB = ldStream(&q.b[nextBase + lane]);
        }
        Ray ray;
        uint32_t seed = 0, pix = kInvalid;
        if (FUSED) {  // generateKernel's work: position -> (frame, pixel) -> camera ray
            const uint32_t f = mine / fa.perFrame, ii = mine - f * fa.perFrame;
            uint32_t px, lrow;
            ray.origin = ray.direction = mk(0.f, 0.f, 0.f);
            if (primaryPixel(tile, ii, px, lrow)) {
                ray = primaryRay(*fa.rdDev, fa.rdDev->numAccumulationFrames + f, px, tile.globalRow(lrow), seed);
                pix = f * (tile.rows * tile.W) + lrow * tile.W + px;
            }
        } else {
            ray.origin = mk(A.x, A.y, A.z);
            ray.direction = mk(A.w, B.x, B.y);
            pix = __float_as_uint(B.w);
        }
        const bool valid = pix != kInvalid;
        if (__ballot(valid) == 0ull) continue;
        if (valid) nSeg++;
        const V3 ro = ray.origin;
        const V3 rdir = valid ? normalize(ray.direction) : mk(0.f, 0.f, 1.f);  // the shader re-normalises per instance (:276)
        V3 iq = rcp3(rdir);
        iq = mk(fminf(fmaxf(iq.x, -1e30f), 1e30f), fminf(fmaxf(iq.y, -1e30f), 1e30f), fminf(fmaxf(iq.z, -1e30f), 1e30f));
        float best = valid ? kFltMax : -1.f;  // an invalid lane can never enter a box (tnear < best fails)
        float hbu = 0.f, hbv = 0.f;
        uint32_t hitIdx = kInvalid, hitTri = kInvalid;
        if (valid)
            for (uint32_t i = 0; i < rd.numSpheres; i++) {  // RaytracerCS.hlsl:232-242 (original direction)
                float d = rayAndSphere(ray, ld3(s.spheres[i].position), s.spheres[i].radius);
                if (d > 0.f && d < best) {
                    best = d;
                    hitIdx = i;
                }
            }

        uint32_t sp = 0, cur = 0;  // wave-uniform
#ifdef OKRT_STAMPS
        pTiles++;
#endif
        OKRT_PMARK(pSetup)
        for (;;) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            if (!(cur & kLeafBit)) {
#ifdef OKRT_STAMPS
                pNodeSteps++;
#endif
                const uint4* np = nodes + (size_t)cur * 4;  // uniform address: scalar loads
                const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                if (COUNT) nNodes++;
                const float Ax = __uint_as_float((n0.w & 255u) << 23) * iq.x, Bx = (__uint_as_float(n0.x) - ro.x) * iq.x;
                const float Ay = __uint_as_float(((n0.w >> 8) & 255u) << 23) * iq.y, By = (__uint_as_float(n0.y) - ro.y) * iq.y;
                const float Az = __uint_as_float(((n0.w >> 16) & 255u) << 23) * iq.z, Bz = (__uint_as_float(n0.z) - ro.z) * iq.z;
                const uint32_t nearX = iq.x >= 0.f ? n2.x : n3.x, farX = iq.x >= 0.f ? n3.x : n2.x;
                const uint32_t nearY = iq.y >= 0.f ? n2.y : n3.y, farY = iq.y >= 0.f ? n3.y : n2.y;
                const uint32_t nearZ = iq.z >= 0.f ? n2.z : n3.z, farZ = iq.z >= 0.f ? n3.z : n2.z;
                const uint32_t ch[4] = {n1.x, n1.y, n1.z, n1.w};
                uint32_t key[4], ref[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tnear = fmaxf(fmaxf(fma_((float)((nearX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((nearY >> (8 * k)) & 255u), Ay, By)),
                                              fma_((float)((nearZ >> (8 * k)) & 255u), Az, Bz));
                    const float tfar = fminf(fminf(fma_((float)((farX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((farY >> (8 * k)) & 255u), Ay, By)),
                                             fma_((float)((farZ >> (8 * k)) & 255u), Az, Bz));
                    const bool hit = tfar >= tnear && tfar > 0.f && tnear < best;
                    const unsigned long long m = ch[k] != kEmptyRef ? __ballot(hit) : 0ull;
                    // order key: entry distance (clamped to >= 0, so its bits sort as an integer) of the first lane entering
                    const float t0 = __shfl(fmaxf(tnear, 0.f), m ? __ffsll((long long)m) - 1 : 0, 64);
                    key[k] = m ? __float_as_uint(t0) : 0xFFFFFFFFu;
                    ref[k] = ch[k];
                }
#define OKRT_USWAP(i, j)                                         \
    if (key[j] < key[i]) {                                       \
        const uint32_t tk_ = key[i]; key[i] = key[j]; key[j] = tk_; \
        const uint32_t tr_ = ref[i]; ref[i] = ref[j]; ref[j] = tr_; \
    }
                OKRT_USWAP(0, 1) OKRT_USWAP(2, 3) OKRT_USWAP(0, 2) OKRT_USWAP(1, 3) OKRT_USWAP(1, 2)
#undef OKRT_USWAP
                // push the far ones (3, 2, 1), continue with the nearest; everything here is wave-uniform
#pragma unroll
                for (int k = 3; k >= 1; --k)
                    if (key[k] != 0xFFFFFFFFu && sp < (uint32_t)kPacketStack) {
                        if (lane == 0) stack[sp] = ref[k];
                        sp++;
                    }
                OKRT_PMARK(pNode)
                if (key[0] != 0xFFFFFFFFu) {
                    cur = ref[0];
                    continue;
                }
            } else {
                const uint32_t count = ((cur >> 27) & 15u) + 1u;
#ifdef OKRT_STAMPS
                pLeafTris += count;
#endif
                const float4* tp = tris + (size_t)(cur & 0x07FFFFFFu) * 3;  // uniform address: scalar loads
                for (uint32_t j = 0; j < count; ++j, tp += 3) {
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    if (COUNT) nTris++;
                    float bu = 0.f, bv = 0.f;
                    const float t = rayAndTriangleEdges(ro, rdir, mk(t0.x, t0.y, t0.z), mk(t1.x, t1.y, t1.z), mk(t2.x, t2.y, t2.z), bu, bv);
                    if (t > 0.f) {
                        const float d = length(rdir * t);  // RaytracerCS.hlsl:311-314
                        if (d < best) {
                            best = d;
                            hbu = bu;
                            hbv = bv;
                            hitTri = __float_as_uint(t0.w);
                            hitIdx = __float_as_uint(t1.w);
                        }
                    }
                }
            }
            OKRT_PMARK(pLeaf)
            if (sp == 0) break;
            --sp;
            cur = stack[sp];  // same address in every lane (broadcast); written by lane 0 above
        }
        if (!FUSED) {
            if (valid) {
                stStream(&hb.h[mine], make_float4(best, hbu, hbv, __uint_as_float(hitIdx)));
                stStream(&hb.tri[mine], hitTri);
            }
        } else {  // shadeKernel<FIRST>'s work on the hits this wave holds in registers
            bool alive = false;
            V3 contribution = mk(1.f, 1.f, 1.f), light = mk(0.f, 0.f, 0.f);
            if (valid) {
                nPixels++;
                Hit h;
                h.distance = best;
                h.bu = hbu;
                h.bv = hbv;
                h.hitIdx = hitIdx;
                h.triIdx = hitTri;
                h.hitType = hitTri != kInvalid ? 1u : 0u;
                alive = shadeSegment(s, *fa.rdDev, ray, h, 0u, fa.lastBounce, contribution, light, seed, shadeCnt);
                if (!alive) {
                    if (fa.radiance) stStream(&fa.radiance[pix], make_float4(light.x, light.y, light.z, 1.f));
                    else writePixel(*fa.rdDev, light, 0, 0, fa.accum, fa.result, pix);
                }
            }
            const unsigned long long am = __ballot(alive);
            if (am) {  // survivors stay in the shard their tile came from: one atomic per wave and tile
                const uint32_t tileShard = mine / sv.cap;
                uint32_t obase = 0;
                if (lane == 0) obase = atomicAdd(&fa.outCounts[tileShard * kCursorStride], (uint32_t)__popcll(am));
                obase = __builtin_amdgcn_readfirstlane(obase);
                if (alive) {
                    const uint32_t o = tileShard * sv.cap + obase + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
                    stStream(&fa.qOut.a[o], make_float4(ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x));
                    stStream(&fa.qOut.b[o], make_float4(ray.direction.y, ray.direction.z, __uint_as_float(seed), __uint_as_float(pix)));
                    stStream(&fa.qOut.c[o], make_float4(contribution.x, contribution.y, contribution.z, light.x));
                    stStream(&fa.qOut.d[o], make_float2(light.y, light.z));
                }
            }
        }
    }
    if (FUSED && COUNT && counters) {  // what shadeKernel<FIRST, COUNT> reports
        uint32_t vals[5] = {nPixels, shadeCnt.meshHits, shadeCnt.sphereHits, shadeCnt.texTaps, shadeCnt.envTaps};
        const int slot[5] = {0, 6, 7, 8, 9};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            uint32_t v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&counters[slot[k]], (unsigned long long)v);
        }
    }
#ifdef OKRT_STAMPS
    if (COUNT && counters && lane == 0) {  // n_oct=setup cycles, n_mesh=node, n_node=leaf+pop, n_tri=total, mesh_hits=tiles, sphere_hits=node steps, tex_taps=leaf triangles
        atomicAdd(&counters[2], pSetup); atomicAdd(&counters[3], pNode); atomicAdd(&counters[4], pLeaf);
        atomicAdd(&counters[5], __builtin_amdgcn_s_memtime() - pT0);
        atomicAdd(&counters[6], pTiles); atomicAdd(&counters[7], pNodeSteps); atomicAdd(&counters[8], pLeafTris);
    }
#endif
    if (COUNT && counters) {
        uint32_t v = nSeg;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) {
            if (v) atomicAdd(&counters[1], (unsigned long long)v);
            // wave-uniform visit counts: every lane of the wave saw the same nodes / triangles
            atomicAdd(&counters[10], (unsigned long long)nNodes * 64ull);
            atomicAdd(&counters[11], (unsigned long long)nTris * 64ull);
        }
    }
}

// Shading of one path segment. FIRST: items are pixels (tile order), the camera ray is regenerated.
// Finished paths (miss, or last segment) accumulate; the rest are compacted into `qOut`.
template <bool FIRST, bool COUNT>
__global__ __launch_bounds__(256) void shadeKernel(SceneView s, okrt_render_data rd, TileView tile, RayQueue qIn, HitBuffer hb,
                                                  ShardView in, RayQueue qOut, uint32_t* outCounts, uint32_t bounce,
                                                  uint32_t lastBounce, float4* accum, uint32_t* result, float4* radiance,
                                                  unsigned long long* counters) {
    // positions are [shard][cap]; cap is a multiple of 256, so a block (and a wave) never straddles shards
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const uint32_t shard = (blockIdx.x * 256) / in.cap;
    const uint32_t total = shard * in.cap + in.count(shard);
    PathCounters cnt = {};
    bool alive = false;  // continues into the next segment
    Ray ray;
    V3 contribution = mk(1.f, 1.f, 1.f), light = mk(0.f, 0.f, 0.f);
    uint32_t seed = 0, pix = 0;
    bool valid = i < total;
    if (valid) {
        const float4 A = ldStream(&qIn.a[i]), B = ldStream(&qIn.b[i]);
        ray.origin = mk(A.x, A.y, A.z);
        ray.direction = mk(A.w, B.x, B.y);
        seed = __float_as_uint(B.z);
        pix = __float_as_uint(B.w);
        valid = pix != kInvalid;
        if (!FIRST) {  // camera rays start with throughput 1 and no radiance; later segments carry them
            const float4 C = ldStream(&qIn.c[i]);
            const float2 D = ldStream(&qIn.d[i]);
            contribution = mk(C.x, C.y, C.z);
            light = mk(C.w, D.x, D.y);
        }
    }
    if (valid) {
        float4 H = ldStream(&hb.h[i]);
        Hit h;
        h.distance = H.x;
        h.bu = H.y;
        h.bv = H.z;
        h.hitIdx = __float_as_uint(H.w);
        h.triIdx = ldStream(&hb.tri[i]);
        h.hitType = h.triIdx != kInvalid ? 1u : 0u;
        alive = shadeSegment(s, rd, ray, h, bounce, lastBounce, contribution, light, seed, cnt);
        if (!alive) {
            if (radiance) stStream(&radiance[pix], make_float4(light.x, light.y, light.z, 1.f));  // several frames in flight: added in frame order later
            else writePixel(rd, light, 0, 0, accum, result, pix);
        }
    }
    // ballot compaction of the surviving paths into the next queue: ranks inside the wave by ballot, waves inside
    // the block through LDS, ONE atomic per block on the shard's counter (one per wave saturated the counter line)
    __shared__ uint32_t sWaveCount[4], sBlockBase;
    const unsigned long long m = __ballot(alive);
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) sWaveCount[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = sWaveCount[0] + sWaveCount[1] + sWaveCount[2] + sWaveCount[3];
        sBlockBase = tot ? atomicAdd(&outCounts[shard * kCursorStride], tot) : 0u;  // survivors stay in their shard
    }
    __syncthreads();
    if (alive) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += sWaveCount[w];
        const uint32_t o = shard * in.cap + sBlockBase + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        stStream(&qOut.a[o], make_float4(ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x));
        stStream(&qOut.b[o], make_float4(ray.direction.y, ray.direction.z, __uint_as_float(seed), __uint_as_float(pix)));
        stStream(&qOut.c[o], make_float4(contribution.x, contribution.y, contribution.z, light.x));
        stStream(&qOut.d[o], make_float2(light.y, light.z));
    }
    if (COUNT && counters) {
        uint32_t vals[5] = {(FIRST && valid) ? 1u : 0u, cnt.meshHits, cnt.sphereHits, cnt.texTaps, cnt.envTaps};
        const int slot[5] = {0, 6, 7, 8, 9};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            uint32_t v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&counters[slot[k]], (unsigned long long)v);
        }
    }
}

// Adds the radiance slots of `frames` frames to the accumulation IN FRAME ORDER and resolves with the last frame's
// count: exactly what `frames` consecutive single-frame passes do (RaytracerCS.hlsl:526-536).
__global__ __launch_bounds__(256) void accumulateKernel(okrt_render_data rd, uint32_t pixels, uint32_t frames, const float4* radiance,
                                                       float4* accum, uint32_t* result) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pixels) return;
    float4 a = accum[p];
    for (uint32_t f = 0; f < frames; ++f) {
        const float4 r = ldStream(&radiance[(size_t)f * pixels + p]);
        a.x += r.x; a.y += r.y; a.z += r.z; a.w += 1.f;
    }
    accum[p] = a;
    const float n = (float)(rd.numAccumulationFrames + frames - 1u);
    result[p] = toUnorm8(a.x / n) | (toUnorm8(a.y / n) << 8) | (toUnorm8(a.z / n) << 16) | (toUnorm8(a.w / n) << 24);
}

template <class T>
bool devAlloc(T** p, size_t n, std::string& err) {
    if (*p) hipFree(*p);
    *p = nullptr;
    if (!n) return true;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) {
        err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

}  // namespace

constexpr uint32_t kMaxSegments = 66;                       // max_bounces <= 64
constexpr uint32_t kCursorBase = kMaxSegments * kShards * kCursorStride;
constexpr uint32_t kCountWords = kCursorBase + kMaxSegments * kShards * kCursorStride;

struct FastScene {
    DeviceArray<uint4> nodes;   // 4 x uint4 per Node4Q
    DeviceArray<float4> tris;   // 3 x float4 per Tri48
    uint32_t numNodes = 0, numTris = 0, maxDepth = 0;
    bool allIdentity = true;
    DeviceBuilder* builder = nullptr;  // device-side build scratch (okrt_fast_build_device.hip), made on first use
    FastBuildInfo info = {};
    // pipeline state, sized lazily for the tile
    size_t capacity = 0;
    RayQueue q[2] = {};
    HitBuffer hb = {};
    uint32_t* counts = nullptr;  // per segment b, shard s: count at (b*kShards + s)*kCursorStride, cursor at kCursorBase + the same
                                 // (every counter on its own 128-byte line: device-scope atomics serialise per line)
    uint2* overflow = nullptr;
    uint32_t overflowCap = 0;
    int traceBlocks = 0, cus = 0;
    size_t queueBudgetBytes = 0;  // OKRT_OPT_QUEUE_MEMORY_MB (0 = no limit)
    okrt_render_data* rdDev = nullptr;  // RenderData of the launch set in flight, for the fused camera kernel
    float4* radiance = nullptr;  // per-frame radiance slots when several frames share one launch set
    size_t radianceCap = 0;
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
    fs.info = {builder, fs.numNodes, fs.numTris, fs.maxDepth, deviceMs,
               std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count()};
    // spill slab: a visit pushes at most 3 entries, so 3*depth bounds the stack
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    fs.cus = cus;
    fs.traceBlocks = cus * (getenv("OKRT_FAST_BLOCKS_PER_CU") ? atoi(getenv("OKRT_FAST_BLOCKS_PER_CU")) : OKRT_TRACE_WAVES);
    uint32_t need = 3 * fs.maxDepth + 4;
    uint32_t cap = need > (uint32_t)kLdsStack ? need - kLdsStack : 1;
    if (cap != fs.overflowCap || !fs.overflow) {
        if (!devAlloc(&fs.overflow, (size_t)fs.traceBlocks * kTraceThreads * cap, err)) return false;
        fs.overflowCap = cap;
    }
    if (!fs.counts && !devAlloc(&fs.counts, kCountWords, err)) return false;
    return true;
}

// Known-answer probe (okrt_debug_kat, CLOSEST_HIT_FAST): one thread per ray walks the structure with a private stack —
// the same node decoding, slab arithmetic, triangle test and hit-distance formula as the trace kernels, none of their
// scheduling — so tests can compare the structure's closest hit with the oracle's ray by ray.
__global__ __launch_bounds__(64) void fastKatKernel(const uint4* __restrict__ nodes, const float4* __restrict__ tris, SceneView s, uint32_t n,
                                                    const float* __restrict__ in, uint32_t inFloats, float* __restrict__ out, uint32_t outFloats) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * inFloats;
    float* o = out + (size_t)i * outFloats;
    Ray ray;
    ray.origin = ld3(a);
    ray.direction = ld3(a + 3);
    const uint32_t numSpheres = __float_as_uint(a[6]);
    float best = kFltMax;
    uint32_t hitIdx = kInvalid, hitTri = kInvalid;
    for (uint32_t k = 0; k < numSpheres; k++) {
        const float d = rayAndSphere(ray, ld3(s.spheres[k].position), s.spheres[k].radius);
        if (d > 0.f && d < best) {
            best = d;
            hitIdx = k;
        }
    }
    const V3 ro = ray.origin, rdir = normalize(ray.direction);
    V3 iq = rcp3(rdir);
    iq = mk(fminf(fmaxf(iq.x, -1e30f), 1e30f), fminf(fmaxf(iq.y, -1e30f), 1e30f), fminf(fmaxf(iq.z, -1e30f), 1e30f));
    uint32_t stack[256];
    uint32_t sp = 0, visited = 0;
    stack[sp++] = 0;
    while (sp) {
        const uint32_t cur = stack[--sp];
        if (cur == kEmptyRef) continue;
        if (cur & kLeafBit) {
            const uint32_t count = ((cur >> 27) & 15u) + 1u;
            const float4* tp = tris + (size_t)(cur & 0x07FFFFFFu) * 3;
            for (uint32_t j = 0; j < count; ++j, tp += 3) {
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                float bu = 0.f, bv = 0.f;
                const float t = rayAndTriangleEdges(ro, rdir, mk(t0.x, t0.y, t0.z), mk(t1.x, t1.y, t1.z), mk(t2.x, t2.y, t2.z), bu, bv);
                if (t > 0.f) {
                    const float d = length(rdir * t);
                    if (d < best) {
                        best = d;
                        hitTri = __float_as_uint(t0.w);
                        hitIdx = __float_as_uint(t1.w);
                    }
                }
            }
            continue;
        }
        ++visited;
        const uint4* np = nodes + (size_t)cur * 4;
        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        const uint32_t ch[4] = {n1.x, n1.y, n1.z, n1.w};
        const float Ax = __uint_as_float((n0.w & 255u) << 23) * iq.x, Bx = (__uint_as_float(n0.x) - ro.x) * iq.x;
        const float Ay = __uint_as_float(((n0.w >> 8) & 255u) << 23) * iq.y, By = (__uint_as_float(n0.y) - ro.y) * iq.y;
        const float Az = __uint_as_float(((n0.w >> 16) & 255u) << 23) * iq.z, Bz = (__uint_as_float(n0.z) - ro.z) * iq.z;
        const uint32_t nearX = iq.x >= 0.f ? n2.x : n3.x, farX = iq.x >= 0.f ? n3.x : n2.x;
        const uint32_t nearY = iq.y >= 0.f ? n2.y : n3.y, farY = iq.y >= 0.f ? n3.y : n2.y;
        const uint32_t nearZ = iq.z >= 0.f ? n2.z : n3.z, farZ = iq.z >= 0.f ? n3.z : n2.z;
        for (int k = 0; k < 4; ++k) {
            const float tnear = fmaxf(fmaxf(fma_((float)((nearX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((nearY >> (8 * k)) & 255u), Ay, By)),
                                      fma_((float)((nearZ >> (8 * k)) & 255u), Az, Bz));
            const float tfar = fminf(fminf(fma_((float)((farX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((farY >> (8 * k)) & 255u), Ay, By)),
                                     fma_((float)((farZ >> (8 * k)) & 255u), Az, Bz));
            if (tfar >= tnear && tfar > 0.f && tnear < best && ch[k] != kEmptyRef && sp < 256u) stack[sp++] = ch[k];
        }
    }
    o[0] = best;
    o[1] = __uint_as_float(hitIdx != kInvalid ? 1u : 0u);
    o[2] = __uint_as_float(visited);
    o[3] = __uint_as_float(hitIdx);
    o[4] = __uint_as_float(hitTri);
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
    if (n) hipLaunchKernelGGL(fastKatKernel, dim3((n + 63) / 64), dim3(64), 0, stream, fs->nodes.p, fs->tris.p, s, n, in, inFloats, out, outFloats);
    return true;
}

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
    for (void* p : {(void*)fs->rdDev, (void*)fs->counts, (void*)fs->overflow, (void*)fs->hb.h, (void*)fs->hb.tri, (void*)fs->radiance})
        if (p) hipFree(p);
    for (RayQueue& q : fs->q)
        for (void* p : {(void*)q.a, (void*)q.b, (void*)q.c, (void*)q.d})
            if (p) hipFree(p);
    delete fs;
}

bool launchFast(FastScene& fs, const SceneView& s, const okrt_render_data& rd, const TileView& t, float4* accum, uint32_t* result,
                unsigned long long* counters, unsigned long long* rayTotals, hipStream_t stream, hipEvent_t* events, uint32_t frames,
                std::string& err) {
    const uint32_t perFrame = (((t.W + 7) >> 3) * ((t.rows + 7) >> 3)) << 6;  // camera rays of one frame incl. tile padding
    if (frames < 1 || frames > kMaxFramesPerLaunch || (uint64_t)perFrame * frames > 0x7FFFFFFFull) {
        err = "frames per launch out of range";
        return false;
    }
    const uint32_t total = perFrame * frames;
    const uint32_t cap = (((total + kShards - 1) / kShards + 255) / 256) * 256;
    const size_t items = (size_t)cap * kShards;
    // per item: two ray queues (16+16+16+8 B each) + hit record (16+4 B); per pixel and frame: a 16-byte radiance slot
    const size_t needBytes = items * (2 * 56 + 20) + (frames > 1 ? (size_t)t.rows * t.W * frames * 16 : 0);
    if (fs.queueBudgetBytes && needBytes > fs.queueBudgetBytes && frames > 1) {
        err = "launch set exceeds the queue memory budget";
        return false;
    }
    if (items > fs.capacity) {
        hipStreamSynchronize(stream);
        fs.capacity = 0;  // the old buffers are released first: a failed allocation must not leave a stale capacity behind
        for (RayQueue& q : fs.q)
            if (!devAlloc(&q.a, items, err) || !devAlloc(&q.b, items, err) || !devAlloc(&q.c, items, err) || !devAlloc(&q.d, items, err)) return false;
        if (!devAlloc(&fs.hb.h, items, err) || !devAlloc(&fs.hb.tri, items, err)) return false;
        fs.capacity = items;
    }
    const size_t pixels = (size_t)t.rows * t.W;
    float4* radiance = nullptr;
    if (frames > 1) {
        if (pixels * frames > fs.radianceCap) {
            hipStreamSynchronize(stream);
            fs.radianceCap = 0;
            if (!devAlloc(&fs.radiance, pixels * frames, err)) return false;
            fs.radianceCap = pixels * frames;
        }
        radiance = fs.radiance;
    }
    const uint32_t lastBounce = s.maxBounces;
    if (lastBounce + 2 > kMaxSegments) {
        err = "max_bounces too large for the fast pipeline";
        return false;
    }
    hipMemsetAsync(fs.counts, 0, (size_t)kCountWords * sizeof(uint32_t), stream);
    FastView fv{fs.nodes.p, fs.tris.p, fs.overflow, fs.overflowCap};
    const dim3 tb(kTraceThreads), tg(fs.traceBlocks);
    const dim3 sg((unsigned)(items / 256));
    const bool cnt = counters != nullptr;
    int ev = 0;
    auto mark = [&]() { if (events) hipEventRecord(events[ev++], stream); };
    static const uint32_t envBatch = getenv("OKRT_FAST_BATCH") ? (uint32_t)atoi(getenv("OKRT_FAST_BATCH")) : kBatch;
    static const uint32_t envAffinity = getenv("OKRT_FAST_AFFINITY") ? (uint32_t)atoi(getenv("OKRT_FAST_AFFINITY")) : 1u;
    static const uint32_t envNodeMin = getenv("OKRT_FAST_NODE_MIN") ? (uint32_t)atoi(getenv("OKRT_FAST_NODE_MIN")) : 32u;
    static const uint32_t envTriMin = getenv("OKRT_FAST_TRI_MIN") ? (uint32_t)atoi(getenv("OKRT_FAST_TRI_MIN")) : 24u;
    static const uint32_t envRefill = getenv("OKRT_FAST_REFILL") ? (uint32_t)atoi(getenv("OKRT_FAST_REFILL")) : (uint32_t)kRefillBelow;
    auto shards = [&](uint32_t b, bool camera) {
        return ShardView{cap, camera ? total : 0u, envBatch, envAffinity, envRefill, envNodeMin, envTriMin, camera ? nullptr : fs.counts + b * kShards * kCursorStride,
                         fs.counts + kCursorBase + b * kShards * kCursorStride};
    };
    mark();
    static const bool envPacket = !getenv("OKRT_FAST_PACKET") || atoi(getenv("OKRT_FAST_PACKET")) != 0;
    static const bool envFused = !getenv("OKRT_FAST_FUSED") || atoi(getenv("OKRT_FAST_FUSED")) != 0;
    const bool packet = envPacket && 3 * fs.maxDepth + 4 <= (uint32_t)kPacketStack;  // camera rays: one wave walks the tree for its tile
    static const int envPacketBlocks = getenv("OKRT_FAST_PACKET_BLOCKS") ? atoi(getenv("OKRT_FAST_PACKET_BLOCKS")) : 6;
    const dim3 pg((unsigned)(fs.cus * envPacketBlocks));  // (2 KB LDS: more blocks per CU fit, but the sweep is flat from 6 to 16)
    if (!fs.rdDev && !devAlloc(&fs.rdDev, 1, err)) return false;
    hipMemcpyAsync(fs.rdDev, &rd, sizeof(rd), hipMemcpyHostToDevice, stream);  // stream order: the previous launch set has read its copy
    const FusedArgs fa{perFrame, fs.q[1], fs.counts + kShards * kCursorStride, lastBounce, accum, result, radiance, fs.rdDev};
    if (packet && envFused) {
        // segment 0 in ONE kernel: camera rays are generated, traced as 8x8 packets and shaded by the same wave
        if (cnt)
            hipLaunchKernelGGL((tracePacketKernel<true, true>), pg, tb, 0, stream, fs.nodes.p, fs.tris.p, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals, fa);
        else
            hipLaunchKernelGGL((tracePacketKernel<false, true>), pg, tb, 0, stream, fs.nodes.p, fs.tris.p, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals, fa);
        mark();  // (an empty "shade" interval keeps the trace / shade alternation of the event list)
    } else {
        // segment 0 as three kernels: camera rays into q[0] (a, b only), trace, shade
        hipLaunchKernelGGL(generateKernel, sg, dim3(256), 0, stream, rd, t, fs.q[0], perFrame, frames);
        if (packet && cnt)
            hipLaunchKernelGGL((tracePacketKernel<true, false>), pg, tb, 0, stream, fs.nodes.p, fs.tris.p, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals, fa);
        else if (packet)
            hipLaunchKernelGGL((tracePacketKernel<false, false>), pg, tb, 0, stream, fs.nodes.p, fs.tris.p, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals, fa);
        else if (cnt)
            hipLaunchKernelGGL((traceKernel<true>), tg, tb, 0, stream, fv, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals);
        else
            hipLaunchKernelGGL((traceKernel<false>), tg, tb, 0, stream, fv, s, rd, t, fs.q[0], fs.hb, shards(0, true), frames, counters, rayTotals);
        mark();
        if (cnt)
            hipLaunchKernelGGL((shadeKernel<true, true>), sg, dim3(256), 0, stream, s, rd, t, fs.q[0], fs.hb, shards(0, true), fs.q[1], fs.counts + kShards * kCursorStride, 0u, lastBounce, accum, result, radiance, counters);
        else
            hipLaunchKernelGGL((shadeKernel<true, false>), sg, dim3(256), 0, stream, s, rd, t, fs.q[0], fs.hb, shards(0, true), fs.q[1], fs.counts + kShards * kCursorStride, 0u, lastBounce, accum, result, radiance, counters);
    }
    if (lastBounce == 0 && radiance)
        hipLaunchKernelGGL(accumulateKernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, rd, (uint32_t)pixels, frames, radiance, accum, result);
    mark();
    // segments 1..lastBounce: queue b lives in q[b & 1]
    for (uint32_t b = 1; b <= lastBounce; ++b) {
        RayQueue& qin = fs.q[b & 1];
        RayQueue& qout = fs.q[(b + 1) & 1];
        if (cnt)
            hipLaunchKernelGGL((traceKernel<true>), tg, tb, 0, stream, fv, s, rd, t, qin, fs.hb, shards(b, false), 0u, counters, rayTotals);
        else
            hipLaunchKernelGGL((traceKernel<false>), tg, tb, 0, stream, fv, s, rd, t, qin, fs.hb, shards(b, false), 0u, counters, rayTotals);
        mark();
        if (cnt)
            hipLaunchKernelGGL((shadeKernel<false, true>), sg, dim3(256), 0, stream, s, rd, t, qin, fs.hb, shards(b, false), qout, fs.counts + (b + 1) * kShards * kCursorStride, b, lastBounce, accum, result, radiance, counters);
        else
            hipLaunchKernelGGL((shadeKernel<false, false>), sg, dim3(256), 0, stream, s, rd, t, qin, fs.hb, shards(b, false), qout, fs.counts + (b + 1) * kShards * kCursorStride, b, lastBounce, accum, result, radiance, counters);
        if (b == lastBounce && radiance)
            hipLaunchKernelGGL(accumulateKernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, rd, (uint32_t)pixels, frames, radiance, accum, result);
        mark();
    }
    return true;
}

}  // namespace okrt
