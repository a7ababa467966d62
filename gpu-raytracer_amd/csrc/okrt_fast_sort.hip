// okrt_fast_sort.hip — DIAGNOSTIC (OKRT_FAST_SORT=<mode> in the environment; off by default, never used by a graded number):
// re-orders the bounce-ray queue between the camera segment and traceKernel with a full radix sort, to measure what ray
// coherence is worth to traceKernel BEFORE engineering a binned append (LABNOTES.md, "ray binning"). A sorted queue is the
// upper bound of what any binning at append time can reach.
//   mode 1  direction octant, rays of an octant in their original (tile) order
//   mode 2  direction octant, then a 24-bit Morton code of the origin on the root node's grid
//   mode 3  Morton code of the origin, then the octant
//   mode 4  12-bit Morton cell, then the octant, then the original order
// Rays stay inside their shard (the shard index is the top of the key; a shard's unused positions sort behind its rays), so
// the shard counters, the hit-record index = queue index rule and every downstream kernel are untouched.
#include <rocprim/device/device_radix_sort.hpp>

#include "okrt_fast_internal.h"

namespace okrt {
namespace fastdetail {
namespace {

__device__ __forceinline__ uint32_t spread8(uint32_t v) {  // 8 bits -> every third bit
    v = (v | (v << 16)) & 0x0300F00Fu;
    v = (v | (v << 8)) & 0x0300F0C3u;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void sortKeyKernel(const uint4* __restrict__ nodes, RayQueue q, ShardView sv, uint32_t items, uint32_t mode,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const uint32_t shard = i / sv.cap, local = i - shard * sv.cap;
    uint32_t sub = 0x0FFFFFFFu;  // unused positions: behind every ray of the shard
    if (local < sv.count(shard)) {
        const float4 A = q.a[i], B = q.b[i];
        const uint32_t oct = (A.w < 0.f ? 1u : 0u) | (B.x < 0.f ? 2u : 0u) | (B.y < 0.f ? 4u : 0u);
        const uint4 n0 = nodes[0];  // root grid: origin + [0, 255] * 2^e per axis
        const float sx = __uint_as_float((n0.w & 255u) << 23), sy = __uint_as_float(((n0.w >> 8) & 255u) << 23), sz = __uint_as_float(((n0.w >> 16) & 255u) << 23);
        const uint32_t cx = (uint32_t)fminf(fmaxf((A.x - __uint_as_float(n0.x)) / sx, 0.f), 255.f);
        const uint32_t cy = (uint32_t)fminf(fmaxf((A.y - __uint_as_float(n0.y)) / sy, 0.f), 255.f);
        const uint32_t cz = (uint32_t)fminf(fmaxf((A.z - __uint_as_float(n0.z)) / sz, 0.f), 255.f);
        const uint32_t morton = spread8(cx) | (spread8(cy) << 1) | (spread8(cz) << 2);  // 24 bits
        if (mode == 1) sub = oct << 24;
        else if (mode == 2) sub = (oct << 24) | morton;
        else if (mode == 3) sub = (morton << 3) | oct;
        else sub = ((morton >> 12) << 15) | (oct << 12);
        sub = min(sub, 0x0FFFFFFEu);
    }
    keys[i] = (shard << 28) | sub;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void sortGatherKernel(RayQueue src, RayQueue dst, const uint32_t* __restrict__ idx, uint32_t items) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const uint32_t j = idx[i];
    dst.a[i] = src.a[j];
    dst.b[i] = src.b[j];
    dst.c[i] = src.c[j];
    dst.d[i] = src.d[j];
}

struct SortScratch {
    uint32_t *keysA = nullptr, *keysB = nullptr, *valsA = nullptr, *valsB = nullptr;
    void* tmp = nullptr;
    size_t items = 0, tmpBytes = 0;
};
SortScratch g_scratch[4];  // one per queue slot (the diagnostic runs one context per process)

}  // namespace

bool sortQueue(uint32_t slot, uint32_t mode, hipStream_t stream, const uint4* nodes, RayQueue src, RayQueue dst, const ShardView& sv, uint32_t items, std::string& err) {
    SortScratch& s = g_scratch[slot & 3];
    auto ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        err = std::string("OKRT_FAST_SORT: ") + what + ": " + hipGetErrorString(e);
        return false;
    };
    size_t need = 0;
    if (!ok(rocprim::radix_sort_pairs(nullptr, need, s.keysA, s.keysB, s.valsA, s.valsB, (size_t)items, 0u, 32u, stream), "sort size")) return false;
    if (items > s.items || need > s.tmpBytes) {
        hipStreamSynchronize(stream);
        for (void* p : {(void*)s.keysA, (void*)s.keysB, (void*)s.valsA, (void*)s.valsB, s.tmp})
            if (p) hipFree(p);
        s = SortScratch{};
        if (!ok(hipMalloc((void**)&s.keysA, items * 4), "alloc") || !ok(hipMalloc((void**)&s.keysB, items * 4), "alloc") || !ok(hipMalloc((void**)&s.valsA, items * 4), "alloc") ||
            !ok(hipMalloc((void**)&s.valsB, items * 4), "alloc") || !ok(hipMalloc(&s.tmp, need), "alloc"))
            return false;
        s.items = items;
        s.tmpBytes = need;
    }
    const dim3 grid((items + 255) / 256), tb(256);
    hipLaunchKernelGGL(sortKeyKernel, grid, tb, 0, stream, nodes, src, sv, items, mode, s.keysA, s.valsA);
    size_t bytes = s.tmpBytes;
    if (!ok(rocprim::radix_sort_pairs(s.tmp, bytes, s.keysA, s.keysB, s.valsA, s.valsB, (size_t)items, 0u, 32u, stream), "sort")) return false;
    hipLaunchKernelGGL(sortGatherKernel, grid, tb, 0, stream, src, dst, s.valsB, items);
    return ok(hipGetLastError(), "launch");
}

}  // namespace fastdetail
}  // namespace okrt
