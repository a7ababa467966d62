// okrt_fast_trace.hip — the traversal kernels of the wavefront pipeline (okrt_fast_pipeline.hip has the overview):
//   traceKernel         per-ray, persistent, one step per lane and iteration: every segment but the camera one
//   tracePacketKernel   one wave per packet of 64 camera rays (8x8 pixels of one frame, or 4x2 / 4x4 pixels of eight / four frames);
//                       FUSED = the whole camera segment (generate, trace, shade, append)
//   fastKatKernel       one thread per ray, the known-answer probe
#include "okrt_fast_internal.h"

#ifndef OKRT_ASM_POP
#define OKRT_ASM_POP 1
#endif
#ifndef OKRT_TRACE_EMPTY_GUARD
#define OKRT_TRACE_EMPTY_GUARD 1  // 0 = round 3's kernel, for showing that test_an_empty_child_slot_that_collapses... fails without the guard
#endif

namespace okrt {
namespace fastdetail {
namespace {

typedef uint32_t bufu4_t __attribute__((ext_vector_type(4)));
constexpr int kRawBufferWord3 = 0x00020000;  // untyped 32-bit data, no swizzle: the raw-buffer descriptor word of gfx9-family parts
__device__ __forceinline__ uint4 bufLoad16(__amdgpu_buffer_rsrc_t r, uint32_t byteOffset) {
    const bufu4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteOffset, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// 8 bytes from LDS with one ds_read_b64, kept out of the optimiser's reach (`p` must point into __shared__ memory)
__device__ __forceinline__ uint2 ldsRead64(const uint2* p) {
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint2*)p;
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// ---- per-ray traversal ---------------------------------------------------------------------------------------------
// Persistent waves; every iteration a lane does ONE unit of work, a node visit or a triangle test ("if-if" scheduling;
// the first version descended while-while — all node steps, then the whole leaf — and the lanes of a wave, which need
// very different numbers of steps before their next leaf, waited for the slowest one). A lane's pending work sits in two
// registers: `cur`, the node to visit next, and `leaf`, the leaf whose triangles are being tested; the wave runs ONE
// step block per iteration, the one more lanes can join (triangles from kDualTriMin takers, or when they outnumber the
// nodes), and all loads of the block are issued before anything is computed (buffer loads: one descriptor per table, a
// 32-bit offset per lane). Round 3
// measurements that shaped it (Sponza-class / 871 k-triangle scene, bounce trace ms per frame, 64-frame sets):
//   round-2 kernel (both blocks per iteration above lane thresholds, flat-addressed pops)        0.274 / 0.152
//   this loop                                                                                    0.252 / 0.140
//   ... with a parked leaf while the lane traverses on (OKRT_DUAL_SPECULATE=1: +9 % node visits)  0.282 / 0.163
//   ... both blocks in one iteration above thresholds (OKRT_DUAL_POLICY=0)                        0.269 / 0.143
//   ... 8 waves per SIMD (64 VGPRs: 48-56 B of scratch in the loop)                               0.659 / 0.383
//   this loop with buffer loads instead of 64-bit global addresses                               0.233 / 0.134
//   ... triangles (and nodes below the first visits) with sc1 / nt cache policy, to keep them out of L1   0.242-0.47 / 0.18-0.41
// MODE: what a finished ray leaves behind. 0 = the 20-byte hit record. 1 = hit-or-miss only, for the LAST segment of a scene in
// which a hit there adds nothing (SceneView::lastSegmentDark): the kernel that appended the ray wrote
// the radiance a hit ends with into its pixel's slot, so only a MISS stores anything here (the bits of kInvalid over the first
// word of its queue entry's `d` pair, which holds the +0 radiance such a path carries; missKernel shades those) — the traversal itself is the same closest-hit walk. 2 = as 1, and the ray ends at its first accepted hit
// (OKRT_OPT_LAST_SEGMENT_ANY_HIT, off by default, never the graded kernel).
template <bool COUNT, int MODE>
__global__ __launch_bounds__(kTraceThreads) __attribute__((amdgpu_waves_per_eu(OKRT_TRACE_WAVES, OKRT_TRACE_WAVES))) void traceKernel(
    FastView fv, SceneView s, okrt_render_data rd, TileView tile, RayQueue q, HitBuffer hb, ShardView sv, uint32_t cameraFrames, unsigned long long* counters,
    unsigned long long* rayTotals) {
    __shared__ uint2 sStack[kLdsStack + 1][kTraceThreads];  // (ref, entry distance), [entry][thread]; row kLdsStack = dummy for predicated-off pushes
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint2* myOverflow = fv.overflow + (size_t)(blockIdx.x * kTraceThreads + tid) * fv.overflowCap;
    if (!COUNT && blockIdx.x == 0 && tid == 0) {  // rays of this launch, always counted (okrt_counters.pixels / .segments)
        unsigned long long n = 0;
        if (cameraFrames) n = (unsigned long long)tile.W * tile.rows * cameraFrames;
        else for (uint32_t k = 0; k < kShards; ++k) n += sv.count(k);
        if (cameraFrames) atomicAdd(&rayTotals[0], n);
        atomicAdd(&rayTotals[1], n);
    }
    // Nodes and triangles are fetched through buffer descriptors: a 32-bit byte offset per lane instead of a 64-bit address
    // (two fewer VALU instructions per fetch and half the address registers; bounce trace 0.242 -> 0.233 ms per frame on the
    // Sponza-class scene, 0.137 -> 0.134 on the 871 k one), and a reference outside its table reads zeros instead of faulting.
    // Both tables stay below 4 GiB (fastBuild refuses larger ones; a D3D11 buffer cannot hold more than 2 GiB anyway).
    const __amdgpu_buffer_rsrc_t nodeRsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fv.nodes, 0, fv.nodeBytes, kRawBufferWord3);
    const __amdgpu_buffer_rsrc_t triRsrc = __builtin_amdgcn_make_buffer_rsrc((void*)fv.tris, 0, fv.triBytes, kRawBufferWord3);
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t shard = sv.xcdAffinity ? (xcc & (kShards - 1)) : 0u, tried = 0, resNext = 0, resEnd = 0;
    uint32_t preRaw = 0, preShard = 0;  // prefetched reservation: preRaw is only meaningful in lane 0
    bool havePre = false;

    uint32_t rayIdx = kInvalid;   // the ray this lane is tracing
    uint32_t cur = kDone;         // node (or, while the leaf slot is taken, a second leaf) to process next; kDone = none
    uint32_t leaf = kDone;        // parked leaf, kDone = none
    uint32_t triJ = 0;            // the triangle of `leaf` to test next
    uint32_t sp = 0;
    V3 ro = mk(0, 0, 0), rdir = mk(0, 0, 0), iq = mk(0, 0, 0);
    float best = kFltMax, hbu = 0.f, hbv = 0.f;
    uint32_t hitIdx = kInvalid, hitTri = kInvalid;
    uint32_t nNodes = 0, nTris = 0, nSeg = 0;
    bool exhausted = false;
#ifdef OKRT_STAMPS_BLOCKS  // diagnostic build: wave-level executions of the two step blocks and the lanes that took part (tools/stamps2.py)
    unsigned long long stNodeWave = 0, stNodeLane = 0, stTriWave = 0, stTriLane = 0, stIter = 0, stRefill = 0, stLive = 0;
#endif

    // next stack entry the current best does not already beat
    auto popNext = [&]() -> uint32_t {
        while (sp > 0) {
            --sp;
            uint2 e;
#if OKRT_ASM_POP
            // (left to itself the compiler merges the two sources into ONE flat load through a selected 64-bit address and
            // fetches the reference with a second, dependent flat load after the loop; the entry is in LDS in all but the
            // deepest corners of the tree and one ds_read_b64 brings both halves)
            if (sp < (uint32_t)kLdsStack) e = ldsRead64(&sStack[sp][tid]);
#else
            if (sp < (uint32_t)kLdsStack) e = sStack[sp][tid];
#endif
            else e = myOverflow[sp - kLdsStack];
            // (e.x != kDone: an EMPTY child slot can be entered — and pushed — when its inverted box collapses under rounding on all
            // three axes, a tiny node seen from far away by a ray through its grid origin; kEmptyRef == kDone would end the ray here)
            if (__uint_as_float(e.y) < best && (!OKRT_TRACE_EMPTY_GUARD || e.x != kDone)) return e.x;
        }
        return kDone;
    };
    // a leaf reference waiting in `cur` moves into the free leaf slot and the traversal goes on with the next entry
    auto parkLeaf = [&]() {
        if ((int)cur < -1 && leaf == kDone) {  // bit 31 set and not kDone
            leaf = cur;
            triJ = 0;
#if OKRT_DUAL_SPECULATE
            cur = popNext();
#else
            cur = kHold;  // depth-first order kept: the stack is popped once this leaf has been tested
#endif
        }
    };
    for (;;) {
        // ---- refill: lanes whose traversal ended store their hit and take the next rays --------
        const unsigned long long idle = __ballot(cur == kDone && leaf == kDone);
        if ((uint32_t)__popcll(~idle) < (uint32_t)kRefillBelow) {
#ifdef OKRT_STAMPS_BLOCKS
            stRefill++;
#endif
            if (cur == kDone && leaf == kDone && rayIdx != kInvalid) {
                if (MODE == 0) {
                    stStream(&hb.h[rayIdx], make_float4(best, hbu, hbv, __uint_as_float(hitIdx)));
                    stStream(&hb.tri[rayIdx], hitTri);
                } else if (hitIdx == kInvalid) {
                    stStream(reinterpret_cast<uint32_t*>(&q.d[rayIdx]), kInvalid);
                }
                rayIdx = kInvalid;
            }
            for (int round = 0; round < 2 && !exhausted; ++round) {
                const unsigned long long need = __ballot(rayIdx == kInvalid);
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
                if (rayIdx == kInvalid && rank < avail) {
                    const uint32_t mine = resNext + rank;
                    const float4 A = ldStream(&q.a[mine]), B = ldStream(&q.b[mine]);
                    Ray ray;
                    ray.origin = mk(A.x, A.y, A.z);
                    ray.direction = mk(A.w, B.x, B.y);
                    if (__float_as_uint(B.w) != kInvalid) {  // camera-ray padding outside the image carries pix = kInvalid
                        rayIdx = mine;
                        nSeg++;
                        ro = ray.origin;
                        rdir = normalize(ray.direction);  // the shader re-normalises the direction per instance (RaytracerCS.hlsl:276)
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
                        cur = (MODE == 2 && hitIdx != kInvalid) ? kDone : 0u;  // root node (any-hit: a sphere hit already ends the ray)
                        leaf = kDone;
                    }
                }
                resNext += min((uint32_t)__popcll(need), avail);
            }
            if (__ballot(rayIdx != kInvalid) == 0ull) break;  // queues drained and every lane idle
        }

        // ---- step blocks ----------------------------------------------------------------------------------------------
        // A block runs when at least kNodeMin / kTriMin lanes want it (both may run in one iteration: one wait for both
        // sets of loads); when neither has that many takers the bigger group runs, the other lanes wait a round and
        // join a fuller block. All loads of the iteration are issued together, through one pointer per lane.
        const bool hasNode = (int)cur >= 0, hasLeaf = leaf != kDone;
        const uint32_t wantNode = (uint32_t)__popcll(__ballot(hasNode)), wantTri = (uint32_t)__popcll(__ballot(hasLeaf));
#if OKRT_DUAL_POLICY == 0
        bool runNode = wantNode >= kNodeMin, runTri = wantTri >= kTriMin;
        if (!runNode && !runTri) {
            runNode = wantNode >= wantTri && wantNode;
            runTri = !runNode && wantTri;
        }
#elif OKRT_DUAL_POLICY == 1
        // the choice that retires the most work per instruction issued THIS iteration: a node visit is worth its ~176 VALU
        // instructions, a triangle test its ~95, every iteration pays ~40 of bookkeeping on top (x 1024, integers)
        const uint32_t effNode = 834u * wantNode, effTri = 721u * wantTri, effBoth = 579u * wantNode + 313u * wantTri;
        const bool runNode = wantNode && (effNode >= effTri || effBoth > effTri);
        const bool runTri = wantTri && (effTri > effNode || effBoth > effNode);
#else
        const bool runTri = wantTri >= kDualTriMin || wantTri > wantNode, runNode = !runTri;
#endif
        const bool stepNode = hasNode && runNode, stepTri = hasLeaf && runTri && !stepNode;  // (speculating lanes hold both: the node first)
#ifdef OKRT_STAMPS_BLOCKS
        stIter++;
        stLive += __popcll(__ballot(cur != kDone || leaf != kDone));
        if (runNode) { stNodeWave++; stNodeLane += __popcll(__ballot(stepNode)); }
        if (runTri) { stTriWave++; stTriLane += __popcll(__ballot(stepTri)); }
#endif
        uint4 d0, d1, d2, d3;
        if (stepNode) {
            const uint32_t off = cur * 64u;
            d0 = bufLoad16(nodeRsrc, off);
            d1 = bufLoad16(nodeRsrc, off + 16);
            d2 = bufLoad16(nodeRsrc, off + 32);
            d3 = bufLoad16(nodeRsrc, off + 48);
        } else if (stepTri) {
            const uint32_t off = ((leaf & 0x07FFFFFFu) + triJ) * 48u;
            d0 = bufLoad16(triRsrc, off);
            d1 = bufLoad16(triRsrc, off + 16);
            d2 = bufLoad16(triRsrc, off + 32);
        }
        if (stepNode) {
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
                const bool hit = tfar >= tnear && tfar > 0.f && tnear < best;  // (empty slots carry an inverted box: entered only in the corner case popNext filters)
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
                        if (sp < (uint32_t)kLdsStack) sStack[sp][tid] = make_uint2(ch[k], __float_as_uint(tn[k]));
                        else if (sp - kLdsStack < fv.overflowCap) myOverflow[sp - kLdsStack] = make_uint2(ch[k], __float_as_uint(tn[k]));
                        else continue;  // cannot happen: the slab is sized from the tree depth
                        sp++;
                    }
                }
            }
            cur = (n && (!OKRT_TRACE_EMPTY_GUARD || ch[0] != kDone)) ? ch[0] : popNext();  // (ch[0] == kDone: the empty slot of the comment in popNext, as the nearest "hit")
            parkLeaf();
        }
        if (stepTri) {  // Moeller-Trumbore on triangle triJ of the parked leaf
            if (COUNT) nTris++;
            float bu = 0.f, bv = 0.f;
            const float t = rayAndTriangleEdges(ro, rdir, mk(__uint_as_float(d0.x), __uint_as_float(d0.y), __uint_as_float(d0.z)),
                                                mk(__uint_as_float(d1.x), __uint_as_float(d1.y), __uint_as_float(d1.z)),
                                                mk(__uint_as_float(d2.x), __uint_as_float(d2.y), __uint_as_float(d2.z)), bu, bv);
            if (t > 0.f) {
                const float d = length(rdir * t);  // RaytracerCS.hlsl:311-314 with M = identity (world-space triangles)
                if (d < best) {  // nearer wins (:316, strict)
                    best = d;
                    hbu = bu;
                    hbv = bv;
                    hitTri = d0.w;
                    hitIdx = d1.w;
                    if (MODE == 2) {  // first accepted hit: done (the stack is simply abandoned)
                        sp = 0;
                        triJ = 15u;
                    }
                }
            }
            if (++triJ > ((leaf >> 27) & 15u)) {  // count - 1 is stored
                leaf = kDone;
#if OKRT_DUAL_SPECULATE
                // what the stack held may be beaten now: a node in `cur` is re-checked when visited (its children fail
                // tnear < best); a second leaf waiting in `cur` moves up
                parkLeaf();
#else
                cur = popNext();
                parkLeaf();
#endif
            }
        }
    }
#ifdef OKRT_STAMPS_BLOCKS
    if (COUNT && counters && lane == 0) {  // pixels += iterations; n_oct.. = node block executions, its lanes, triangle block executions, its lanes
        (void)stRefill; (void)stLive;
        atomicAdd(&counters[0], stLive);
        atomicAdd(&counters[1], stIter << 32);  // (iterations ride in the high half of `segments`)
        atomicAdd(&counters[2], stNodeWave); atomicAdd(&counters[3], stNodeLane); atomicAdd(&counters[4], stTriWave); atomicAdd(&counters[5], stTriLane);
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
// Camera rays arrive in packets of 64 consecutive positions, one packet per wave — an 8x8 pixel tile of one frame, or, where the
// launch set has the frames, 4x2 / 4x4 pixels of eight / four consecutive frames (frame-stacked packets: stackedPacketLane) — and
// the rays of a packet walk almost the same nodes. So the wave traverses the tree ONCE for all its lanes: the current node is wave-uniform,
// its 64 bytes come through the scalar cache into SGPRs (s_load, no vector-memory gather at all), every lane tests its
// own ray against the four child boxes, a child is visited when ANY lane enters it (`__ballot`), order = entry
// distance of the first lane that enters, one shared stack of node references in LDS, no divergent control flow.
// Each lane keeps its own closest hit with the same arithmetic as the per-ray kernel, so results agree bit for bit
// (ties aside). Used for segment 0 only; bounce rays are incoherent and stay on the per-ray kernel.
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
        uint32_t costTile = kInvalid;  // wave-uniform: the tile whose step count this wave records (frame 0 of the set), else kInvalid
        if (FUSED) {  // generateKernel's work: position -> (frame, pixel) -> camera ray
            uint32_t f, ii;
            if (fa.tileOrder) {
                // longest tiles first (okrt_fast_pipeline.hip: "camera tiles, longest first"): shard s holds the tiles of cost rank
                // s, s + 8, s + 16, ... in that order, all frames of a tile next to each other
                const uint32_t s_ = base / sv.cap, u = (base - s_ * sv.cap) >> 6;
                const uint32_t rk = u / fa.frames;
                f = u - rk * fa.frames;
                const uint32_t tileIdx = fa.tileOrder[rk * kShards + s_];
                ii = (tileIdx << 6) + lane;
                if (f == 0u) costTile = tileIdx;
                if (fa.stackFrames) {
                    // frame-stacked packets: of the tile's `frames` packets, every full group of EIGHT is cut as 4x2 pixels x 8 frames and
                    // every full group of FOUR of the rest as 4x4 pixels x 4 frames, instead of 8x8 pixels x 1 frame; what is left over
                    // (frames % 4 packets) stays 8x8. The rays of a pixel differ by their jitter only, so such a packet's footprint is an
                    // eighth / a quarter of a tile's and it enters fewer boxes and leaves (64-frame sets, 8x8x1 / 4x4x4 / 4x2x8: Sponza
                    // camera kernel 0.169 / 0.155 / 0.150 ms per frame, 871 k dragon 0.086 / 0.066 / 0.061; 2x2x16 gains more here and loses
                    // it in missKernel's scattered slot writes, LABNOTES r05). Which rays share a packet cannot change a ray's hit: nearest
                    // wins, equal distances go to the lower triangle index (closerHit), so a frame is the same picture in a set of one and
                    // in a set of twenty.
                    stackedPacketLane(fa.frames, f, lane, tileIdx, ii, f);
                }
            } else {
                f = mine / fa.perFrame;
                ii = mine - f * fa.perFrame;
            }
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
        uint32_t tileCostSteps = 0;  // wave-uniform: node steps + triangle tests of this tile
#ifdef OKRT_ITER_HIST
        uint32_t tileSteps = 0;
#endif
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
                if (FUSED) tileCostSteps++;
#ifdef OKRT_ITER_HIST
                tileSteps++;
#endif
                const float Ax = __uint_as_float((n0.w & 255u) << 23) * iq.x, Bx = (__uint_as_float(n0.x) - ro.x) * iq.x;
                const float Ay = __uint_as_float(((n0.w >> 8) & 255u) << 23) * iq.y, By = (__uint_as_float(n0.y) - ro.y) * iq.y;
                const float Az = __uint_as_float(((n0.w >> 16) & 255u) << 23) * iq.z, Bz = (__uint_as_float(n0.z) - ro.z) * iq.z;
                const uint32_t ch[4] = {n1.x, n1.y, n1.z, n1.w};
                uint32_t key[4], ref[4];
                const uint32_t nearX = iq.x >= 0.f ? n2.x : n3.x, farX = iq.x >= 0.f ? n3.x : n2.x;
                const uint32_t nearY = iq.y >= 0.f ? n2.y : n3.y, farY = iq.y >= 0.f ? n3.y : n2.y;
                const uint32_t nearZ = iq.z >= 0.f ? n2.z : n3.z, farZ = iq.z >= 0.f ? n3.z : n2.z;
                // (choosing the entry planes ONCE per wave on the scalar unit when the tile's rays share an octant — nearly always — saves 12
                // vector instructions per step on paper and cost 4 % in practice: two copies of the child loop, 32 B of scratch. LABNOTES r05)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    key[k] = 0xFFFFFFFFu;
                    ref[k] = ch[k];
                    if (ch[k] == kEmptyRef) continue;  // wave-uniform (the node sits in SGPRs): a scalar branch skips the whole child
                    const float tnear = fmaxf(fmaxf(fma_((float)((nearX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((nearY >> (8 * k)) & 255u), Ay, By)),
                                              fma_((float)((nearZ >> (8 * k)) & 255u), Az, Bz));
                    const float tfar = fminf(fminf(fma_((float)((farX >> (8 * k)) & 255u), Ax, Bx), fma_((float)((farY >> (8 * k)) & 255u), Ay, By)),
                                             fma_((float)((farZ >> (8 * k)) & 255u), Az, Bz));
                    // which lanes enter child k, as a lane mask straight from the three compares (v_cmp into SGPR pairs, combined on the
                    // scalar unit): `__ballot(a && b && c)` costs two more vector instructions per child (a 0 / 1 select and its compare)
                    const unsigned long long m = __builtin_amdgcn_fcmpf(tfar, tnear, 3 /* oge */) & __builtin_amdgcn_fcmpf(tfar, 0.f, 2 /* ogt */) &
                                                 __builtin_amdgcn_fcmpf(tnear, best, 4 /* olt */);
                    // order key: entry distance (clamped to >= 0, so its bits sort as an integer) of the first lane entering.
                    // The lane index is wave-uniform, so this is ONE v_readlane into an SGPR (a __shfl costs an LDS round
                    // trip through ds_bpermute, four per node step) and the sort below runs on the scalar unit
                    if (m) key[k] = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(__builtin_amdgcn_fmed3f(tnear, 0.f, kFltMax)), __ffsll((long long)m) - 1);
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
                if (FUSED) tileCostSteps += count;
#ifdef OKRT_STAMPS
                pLeafTris += count;
#endif
                const float4* tp = tris + (size_t)(cur & 0x07FFFFFFu) * 3;  // uniform address: scalar loads
                for (uint32_t j = 0; j < count; ++j, tp += 3) {
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    if (COUNT) nTris++;
#ifdef OKRT_ITER_HIST
                    tileSteps++;
#endif
                    float bu = 0.f, bv = 0.f;
                    const float t = rayAndTriangleEdges(ro, rdir, mk(t0.x, t0.y, t0.z), mk(t1.x, t1.y, t1.z), mk(t2.x, t2.y, t2.z), bu, bv);
                    if (t > 0.f) {
                        const float d = length(rdir * t);  // RaytracerCS.hlsl:311-314
                        if (closerHit(d, __float_as_uint(t0.w), __float_as_uint(t1.w), best, hitTri, hitIdx)) {
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
#ifdef OKRT_ITER_HIST
        if (COUNT && counters && lane == 0) {
            uint32_t b_ = 0;
            while (b_ < 7 && tileSteps > (16u << b_)) b_++;
            atomicAdd(&counters[2 + b_], 1ull);
            atomicMax(&counters[10], (unsigned long long)tileSteps);
        }
#endif
        if (!FUSED) {
            if (valid) {
                stStream(&hb.h[mine], make_float4(best, hbu, hbv, __uint_as_float(hitIdx)));
                stStream(&hb.tri[mine], hitTri);
            }
        } else {  // shadeKernel<FIRST>'s work on the hits this wave holds in registers
            if (fa.tileCost && costTile != kInvalid && lane == 0) fa.tileCost[costTile] = tileCostSteps;  // what the next set orders its tiles by
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
                    // the next segment is the last and a hit there adds nothing (traceKernel MODE 1): what a hit ends with, now
                    if (fa.terminalNext) stStream(&fa.radiance[pix], make_float4(0.f, 0.f, 0.f, 1.f));
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
#ifndef OKRT_ITER_HIST
            // wave-uniform visit counts: every lane of the wave saw the same nodes / triangles
            atomicAdd(&counters[10], (unsigned long long)nNodes * 64ull);
            atomicAdd(&counters[11], (unsigned long long)nTris * 64ull);
#endif
        }
    }
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
                    if (closerHit(d, __float_as_uint(t0.w), __float_as_uint(t1.w), best, hitTri, hitIdx)) {
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

// OKRT_KAT_CLOSEST_HIT_TRACE: ray i of the probe sits at queue position (i % kShards) * cap + i / kShards, so every shard (and the
// stealing between them) takes part; `collect` = false fills the queue and the shard counts, true copies the hit records back
__global__ __launch_bounds__(256) void katQueueKernel(bool collect, uint32_t n, const float* __restrict__ in, uint32_t inFloats, float* __restrict__ out,
                                                     uint32_t outFloats, RayQueue q, HitBuffer hb, uint32_t cap, uint32_t* counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (!collect && i < kShards) counts[i * kCursorStride] = (n + kShards - 1 - i) / kShards;
    if (i >= n) return;
    const uint32_t pos = (i % kShards) * cap + i / kShards;
    if (!collect) {
        const float* a = in + (size_t)i * inFloats;
        q.a[pos] = make_float4(a[0], a[1], a[2], a[3]);
        q.b[pos] = make_float4(a[4], a[5], 0.f, __uint_as_float(i));
    } else {
        const float4 h = hb.h[pos];
        float* o = out + (size_t)i * outFloats;
        o[0] = h.x;
        o[1] = __uint_as_float(__float_as_uint(h.w) != kInvalid ? 1u : 0u);
        o[2] = 0.f;
        o[3] = h.w;
        o[4] = __uint_as_float(hb.tri[pos]);
    }
}

}  // namespace

void launchKatQueue(hipStream_t stream, bool collect, uint32_t n, const float* in, uint32_t inFloats, float* out, uint32_t outFloats, RayQueue q, HitBuffer hb,
                    uint32_t cap, uint32_t* counts) {
    hipLaunchKernelGGL(katQueueKernel, dim3((std::max(n, kShards) + 255) / 256), dim3(256), 0, stream, collect, n, in, inFloats, out, outFloats, q, hb, cap, counts);
}

void launchTrace(bool count, int mode, dim3 grid, hipStream_t stream, const FastView& fv, const SceneView& s, const okrt_render_data& rd, const TileView& tile,
                 RayQueue q, HitBuffer hb, const ShardView& sv, uint32_t cameraFrames, unsigned long long* counters, unsigned long long* rayTotals) {
    const dim3 tb(kTraceThreads);
    if (mode == 2) hipLaunchKernelGGL((traceKernel<false, 2>), grid, tb, 0, stream, fv, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals);
    else if (mode == 1) hipLaunchKernelGGL((traceKernel<false, 1>), grid, tb, 0, stream, fv, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals);
    else if (count) hipLaunchKernelGGL((traceKernel<true, 0>), grid, tb, 0, stream, fv, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals);
    else hipLaunchKernelGGL((traceKernel<false, 0>), grid, tb, 0, stream, fv, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals);
}

void launchPacket(bool count, bool fused, dim3 grid, hipStream_t stream, const uint4* nodes, const float4* tris, const SceneView& s,
                  const okrt_render_data& rd, const TileView& tile, RayQueue q, HitBuffer hb, const ShardView& sv, uint32_t cameraFrames,
                  unsigned long long* counters, unsigned long long* rayTotals, const FusedArgs& fa) {
    const dim3 tb(kTraceThreads);
    if (count && fused) hipLaunchKernelGGL((tracePacketKernel<true, true>), grid, tb, 0, stream, nodes, tris, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals, fa);
    else if (fused) hipLaunchKernelGGL((tracePacketKernel<false, true>), grid, tb, 0, stream, nodes, tris, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals, fa);
    else if (count) hipLaunchKernelGGL((tracePacketKernel<true, false>), grid, tb, 0, stream, nodes, tris, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals, fa);
    else hipLaunchKernelGGL((tracePacketKernel<false, false>), grid, tb, 0, stream, nodes, tris, s, rd, tile, q, hb, sv, cameraFrames, counters, rayTotals, fa);
}

void launchFastKat(hipStream_t stream, const uint4* nodes, const float4* tris, const SceneView& s, uint32_t n, const float* in, uint32_t inFloats,
                   float* out, uint32_t outFloats) {
    if (n) hipLaunchKernelGGL(fastKatKernel, dim3((n + 63) / 64), dim3(64), 0, stream, nodes, tris, s, n, in, inFloats, out, outFloats);
}

}  // namespace fastdetail
}  // namespace okrt
