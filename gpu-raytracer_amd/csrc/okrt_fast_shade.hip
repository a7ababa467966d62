// okrt_fast_shade.hip — the non-traversal kernels of the wavefront pipeline (okrt_fast_pipeline.hip has the overview):
// camera-ray generation (unfused form), per-segment shading + queue compaction, in-order accumulation of frame slots.
#include "okrt_fast_internal.h"

namespace okrt {
namespace fastdetail {
namespace {

// Camera rays for every owned pixel, in tile order (position i <-> pixel by primaryPixel): origin, direction,
// RNG state after the 4 camera draws, framebuffer offset. Padding positions (outside the image) get pix = kInvalid.
// With `frames` > 1 the positions of frame f follow those of frame f-1 and pix addresses the frame's radiance slot.
__global__ __launch_bounds__(256) void generateKernel(okrt_render_data rd, TileView tile, RayQueue q, uint32_t perFrame, uint32_t frames, uint32_t stacked) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= perFrame * frames) return;
    uint32_t f = i / perFrame, ii = i - f * perFrame;
    if (stacked) {  // tile-major: the `frames` packets of a tile next to each other, cut like the fused camera kernel cuts them
        const uint32_t packet = i >> 6, t = packet / frames;
        stackedPacketLane(frames, packet - t * frames, i & 63u, t, ii, f);
    }
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

// Shading of one path segment. FIRST: items are pixels (tile order), the camera ray is regenerated.
// Finished paths (miss, or last segment) accumulate; the rest are compacted into `qOut`.
template <bool FIRST, bool COUNT>
__global__ __launch_bounds__(256) void shadeKernel(SceneView s, okrt_render_data rd, TileView tile, RayQueue qIn, HitBuffer hb,
                                                  ShardView in, RayQueue qOut, uint32_t* outCounts, uint32_t bounce,
                                                  uint32_t lastBounce, float4* accum, uint32_t* result, float4* radiance,
                                                  unsigned long long* counters, uint32_t terminalNext) {
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
        if (terminalNext) stStream(&radiance[pix], make_float4(0.f, 0.f, 0.f, 1.f));  // (see tracePacketKernel: the next segment is traced by traceKernel MODE 1)
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

// The misses of the last segment when traceKernel ran in MODE 1 / 2 (hits were settled by the kernel that appended the ray):
// RaytracerCS.hlsl:489-493 with the +0 radiance such a path carries (SceneView::lastSegmentDark).
__global__ __launch_bounds__(256) void missKernel(SceneView s, RayQueue qIn, ShardView in, float4* radiance) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t shard = (blockIdx.x * 256) / in.cap;
    if (i >= shard * in.cap + in.count(shard)) return;
    if (ldStream(reinterpret_cast<const uint32_t*>(&qIn.d[i])) != kInvalid) return;  // (+0 = still the radiance the path carried: it hit something)
    const float4 A = ldStream(&qIn.a[i]), B = ldStream(&qIn.b[i]), C = ldStream(&qIn.c[i]);
    PathCounters cnt = {};
    const V3 light = fma3(sampleEnvironment(s, mk(A.w, B.x, B.y), cnt), mk(C.x, C.y, C.z), mk(0.f, 0.f, 0.f));
    stStream(&radiance[__float_as_uint(B.w)], make_float4(light.x, light.y, light.z, 1.f));
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

// Camera tiles, longest first. A launch set's tail is its longest dependent chain: the Sponza-class frame has tiles of 275
// traversal steps among an average of 25, and one of those reserved last keeps the camera kernel alive for ~0.15 ms on an
// emptied GPU — a fifth of a rank's whole region in an 8-way split. The same tile costs the same in the next frame, so every set
// records the steps of its frame-0 tiles and the next set issues its tiles in descending order of that (one block: counting sort
// by min(steps, 1023), any order inside a bucket). order[r] = tile of rank r; shard s of the camera queue then walks ranks s,
// s + 8, ... (a stratified share of every cost class per XCD), all frames of a tile in a row. Which tile is traced when changes
// nothing in the picture (radiance goes to the pixel's own slot).
// (tileCost may be rewritten by another slot's camera kernel while this runs — sets in flight share it — so every cost is read
// ONCE, into `keys`: both passes of the sort must see the same bucket for a tile or the result is no permutation.)
__global__ __launch_bounds__(1024) void orderTilesKernel(const uint32_t* tileCost, uint32_t tiles, uint32_t frames, uint32_t* __restrict__ order,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ shardCounts) {
    __shared__ uint32_t hist[1024], start[1024];
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 1024) {
        const uint32_t b = 1023u - min(__hip_atomic_load(&tileCost[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 1023u);  // bucket 0 = the most expensive
        keys[t] = b;
        atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    // exclusive prefix sum of the 1024 bucket counts (Hillis-Steele in LDS, ten steps; a serial loop in one lane took 40 us of the
    // 41 this kernel cost on the 32 400 tiles of a 1080p frame — on the critical path of every launch set)
    start[tid] = hist[tid];
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? start[tid - d] : 0u;
        __syncthreads();
        start[tid] += v;
        __syncthreads();
    }
    start[tid] -= hist[tid];
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 1024) order[atomicAdd(&start[keys[t]], 1u)] = t;
    if (tid < kShards) shardCounts[tid * kCursorStride] = ((tiles + kShards - 1u - tid) / kShards) * frames * 64u;  // ranks tid, tid + 8, ...
}

// RenderData from the kernel-argument segment into device memory (what the fused camera kernel reads).
__global__ __launch_bounds__(64) void setRenderDataKernel(okrt_render_data rd, okrt_render_data* dst) {
    static_assert(sizeof(okrt_render_data) == 60 * sizeof(uint32_t), "RenderData is 60 words");
    if (threadIdx.x < 60) reinterpret_cast<uint32_t*>(dst)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&rd)[threadIdx.x];
}

}  // namespace

void launchOrderTiles(hipStream_t stream, const uint32_t* tileCost, uint32_t tiles, uint32_t frames, uint32_t* order, uint32_t* keys, uint32_t* shardCounts) {
    hipLaunchKernelGGL(orderTilesKernel, dim3(1), dim3(1024), 0, stream, tileCost, tiles, frames, order, keys, shardCounts);
}

void launchSetRenderData(hipStream_t stream, const okrt_render_data& rd, okrt_render_data* dst) {
    hipLaunchKernelGGL(setRenderDataKernel, dim3(1), dim3(64), 0, stream, rd, dst);
}

void launchGenerate(dim3 grid, hipStream_t stream, const okrt_render_data& rd, const TileView& tile, RayQueue q, uint32_t perFrame, uint32_t frames,
                    uint32_t stacked) {
    hipLaunchKernelGGL(generateKernel, grid, dim3(256), 0, stream, rd, tile, q, perFrame, frames, stacked);
}

void launchShade(bool first, bool count, dim3 grid, hipStream_t stream, const SceneView& s, const okrt_render_data& rd, const TileView& tile,
                 RayQueue qIn, HitBuffer hb, const ShardView& in, RayQueue qOut, uint32_t* outCounts, uint32_t bounce, uint32_t lastBounce,
                 float4* accum, uint32_t* result, float4* radiance, unsigned long long* counters, uint32_t terminalNext) {
    const dim3 tb(256);
    if (first && count) hipLaunchKernelGGL((shadeKernel<true, true>), grid, tb, 0, stream, s, rd, tile, qIn, hb, in, qOut, outCounts, bounce, lastBounce, accum, result, radiance, counters, terminalNext);
    else if (first) hipLaunchKernelGGL((shadeKernel<true, false>), grid, tb, 0, stream, s, rd, tile, qIn, hb, in, qOut, outCounts, bounce, lastBounce, accum, result, radiance, counters, terminalNext);
    else if (count) hipLaunchKernelGGL((shadeKernel<false, true>), grid, tb, 0, stream, s, rd, tile, qIn, hb, in, qOut, outCounts, bounce, lastBounce, accum, result, radiance, counters, terminalNext);
    else hipLaunchKernelGGL((shadeKernel<false, false>), grid, tb, 0, stream, s, rd, tile, qIn, hb, in, qOut, outCounts, bounce, lastBounce, accum, result, radiance, counters, terminalNext);
}

void launchMiss(dim3 grid, hipStream_t stream, const SceneView& s, RayQueue qIn, const ShardView& in, float4* radiance) {
    hipLaunchKernelGGL(missKernel, grid, dim3(256), 0, stream, s, qIn, in, radiance);
}

void launchAccumulate(hipStream_t stream, const okrt_render_data& rd, uint32_t pixels, uint32_t frames, const float4* radiance, float4* accum,
                      uint32_t* result) {
    hipLaunchKernelGGL(accumulateKernel, dim3((pixels + 255) / 256), dim3(256), 0, stream, rd, pixels, frames, radiance, accum, result);
}

}  // namespace fastdetail
}  // namespace okrt
