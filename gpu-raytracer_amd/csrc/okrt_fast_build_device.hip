// okrt_fast_build_device.hip — the traversal structure (okrt_fast.h) built on the GPU.
//
// Replaces, for scene edits, the host path of okrt_fast_build.cpp (and with it the step the reference spends in
// BvhBuilder.cpp:72-97 + the octree rebuild of Application.cpp:312-317 whenever an entity moves): the uploaded tables
// already live in HBM, so the derived structure is rebuilt there in a few milliseconds instead of being re-derived on
// the host and uploaded. Same output contract as buildFastHost — Node4Q nodes + Tri48 triangles over exactly the
// triangle set the reference shader can reach — so every kernel that traces it is unchanged.
//
//   host   octree walk -> reachable instances; per mesh BVH (cached until t2/t0 change) the leaf triangle ranges
//   bake   one thread per (instance, triangle): world-space vertices (the host's fma chain, bit for bit), box,
//          centroid; block-reduced scene bounds
//   sort   63-bit Morton keys of the centroids, rocPRIM radix sort (key, primitive) pairs
//   PLOC   parallel locally-ordered clustering (Meister & Bittner 2018): every cluster looks kRadius neighbours
//          left and right in Morton order for the partner with the smallest union area, mutual pairs merge, the
//          cluster array is compacted (rocPRIM select) — repeated until one cluster is left. Quality is that of
//          a SAH build, cost is ~3 sweeps over the input.
//   top    clustering stops at ~1024 clusters; the levels above them are built top-down by binned SAH on the host
//          (buildTopSAH, okrt_fast_build.cpp: < 1 ms) and uploaded as ordinary BVH2 nodes — a Morton window cannot see
//          the large occluders that decide the upper levels
//   collapse  level by level: each thread turns one BVH2 subtree root into a Node4Q by opening the largest inner
//          child until four, quantises the child boxes (the host's makeNode4Q arithmetic), emits leaves of <= 4
//          triangles as Tri48. Slots are handed out by wave-aggregated atomics (one atomic per wave and counter:
//          a single device-scope word sustains only ~90 atomics/us).
// The tree topology is deterministic (atomics only decide where a node lives, never what it contains).
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "okrt_fast.h"

namespace okrt {

namespace {

constexpr int kBT = 256;              // threads per block of every build kernel
constexpr int kRadius = 16;           // PLOC search radius (default)
constexpr int kMaxRadius = 64;
constexpr uint32_t kLeafTris = 6;     // leaves hold <= 6 triangles, decided by the host build's SAH rule (leafLike)
constexpr uint32_t kIdentityBit = 0x80000000u;

struct alignas(16) Node2D {  // BVH2 node: two float4
    float mn[3]; uint32_t left;   // leaf (id < numPrims): left = primitive index
    float mx[3]; uint32_t right;
};
struct BakeRange { uint32_t instance /* | kIdentityBit */, triStart, count, outOffset; };

struct Counters {  // one 128-byte line each
    uint32_t v[32];
};
enum { kCntInvalid = 0, kCntNode2 = 1, kCntQueue = 2, kCntNode4 = 3, kCntTri = 4, kCntSelected = 5, kCntBounds = 6, kNumCnt = 7 };

__device__ __forceinline__ uint32_t orderedBits(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float fromOrdered(uint32_t u) {
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    std::memcpy(&f, &u, 4);
#endif
    return f;
}

// exclusive wave prefix of `mine` + one atomic on *counter; every lane of the wave must call it
__device__ __forceinline__ uint32_t waveAlloc(uint32_t* counter, uint32_t mine) {
    const int lane = __lane_id();
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + incl - mine;
}

__device__ __forceinline__ float halfArea(float ex, float ey, float ez) {
    if (!(ex >= 0.f) || !(ey >= 0.f) || !(ez >= 0.f)) return 0.f;
    return ex * ey + ey * ez + ez * ex;
}

// ---- bake ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBT) void bakeKernel(const BakeRange* __restrict__ ranges, uint32_t numRanges, uint32_t total,
                                                  const TriangleRec* __restrict__ triPos, const MeshRec* __restrict__ meshes,
                                                  float4* __restrict__ prims, Node2D* __restrict__ nodes, uint32_t* __restrict__ cnt,
                                                  float4* __restrict__ cent, Counters* __restrict__ counters) {
    const uint32_t t = blockIdx.x * kBT + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (t < total) {
        uint32_t a = 0, b = numRanges;  // last range with outOffset <= t
        while (b - a > 1) {
            const uint32_t m = (a + b) >> 1;
            if (ranges[m].outOffset <= t) a = m; else b = m;
        }
        const BakeRange r = ranges[a];
        const uint32_t inst = r.instance & ~kIdentityBit, j = r.triStart + (t - r.outOffset);
        const float* src = triPos[j].p;
        float p[9];
        if (r.instance & kIdentityBit) {
#pragma unroll
            for (int k = 0; k < 9; ++k) p[k] = src[k];
        } else {
            const float* M = meshes[inst].M;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float q0 = src[3 * v], q1 = src[3 * v + 1], q2 = src[3 * v + 2];
#pragma unroll
                for (int c = 0; c < 3; ++c)  // mul(float4(q,1), M), the host build's chain (okrt_fast_build.cpp)
                    p[3 * v + c] = fmaf(M[4 * c + 2], q2, fmaf(M[4 * c + 1], q1, M[4 * c] * q0)) + M[4 * c + 3];
            }
        }
        float mn[3], mx[3];
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(p[k], fminf(p[3 + k], p[6 + k]));
            mx[k] = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) finite = finite && isfinite(p[k]);
        prims[3 * (size_t)t] = make_float4(p[0], p[1], p[2], __uint_as_float(j));
        prims[3 * (size_t)t + 1] = make_float4(p[3], p[4], p[5], __uint_as_float(inst));
        prims[3 * (size_t)t + 2] = make_float4(p[6], p[7], p[8], 0.f);
        Node2D n;
        n.mn[0] = mn[0]; n.mn[1] = mn[1]; n.mn[2] = mn[2]; n.left = t;
        n.mx[0] = mx[0]; n.mx[1] = mx[1]; n.mx[2] = mx[2]; n.right = kEmptyRef;
        nodes[t] = n;
        cnt[t] = 1u;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = 0.5f * mn[k] + 0.5f * mx[k];  // cannot overflow
        cent[t] = make_float4(c[0], c[1], c[2], finite ? 1.f : 0.f);
        if (finite) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = c[k]; hi[k] = c[k]; }
        }
    }
    // block reduction of the centroid bounds, then 6 atomics per block
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    }
    __shared__ float sLo[kBT / 64][3], sHi[kBT / 64][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sLo[wave][k] = lo[k]; sHi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float l = sLo[0][k], h = sHi[0][k];
        for (int w = 1; w < kBT / 64; ++w) { l = fminf(l, sLo[w][k]); h = fmaxf(h, sHi[w][k]); }
        if (l <= h) {
            atomicMin(&counters[kCntBounds].v[k], orderedBits(l));
            atomicMax(&counters[kCntBounds].v[3 + k], orderedBits(h));
        }
    }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {  // bit i -> bit 3i
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kBT) void mortonKernel(const float4* __restrict__ cent, uint32_t total, Counters* __restrict__ counters,
                                                    unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t t = blockIdx.x * kBT + threadIdx.x;
    bool invalid = false;
    if (t < total) {
        const float4 c = cent[t];
        unsigned long long key = ~0ull;
        if (c.w != 0.f) {
            const float cv[3] = {c.x, c.y, c.z};
            uint32_t q[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float l = fromOrdered(counters[kCntBounds].v[k]), h = fromOrdered(counters[kCntBounds].v[3 + k]);
                const float ext = h - l;
                float f = ext > 0.f ? (cv[k] - l) / ext : 0.f;
                f = fminf(fmaxf(f * 2097152.f, 0.f), 2097151.f);
                q[k] = (uint32_t)f;
            }
            key = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
        } else {
            invalid = true;
        }
        keys[t] = key;
        vals[t] = t;
    }
    const unsigned long long m = __ballot(invalid);
    if (m && __lane_id() == 0) atomicAdd(&counters[kCntInvalid].v[0], (uint32_t)__popcll(m));
}

// ---- PLOC ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBT) void nearestKernel(const Node2D* __restrict__ nodes, const uint32_t* __restrict__ clusters, uint32_t n,
                                                     uint32_t* __restrict__ nn, int kRadius) {
    __shared__ float4 sLo[kBT + 2 * kMaxRadius], sHi[kBT + 2 * kMaxRadius];
    const int base = (int)(blockIdx.x * kBT) - kRadius;
    for (int k = threadIdx.x; k < kBT + 2 * kRadius; k += kBT) {
        const int g = base + k;
        if (g >= 0 && g < (int)n) {
            const float4* nd = (const float4*)&nodes[clusters[g]];
            sLo[k] = nd[0];
            sHi[k] = nd[1];
        }
    }
    __syncthreads();
    const int i = (int)(blockIdx.x * kBT + threadIdx.x);
    if (i >= (int)n) return;
    const float4 a0 = sLo[threadIdx.x + kRadius], a1 = sHi[threadIdx.x + kRadius];
    float best = INFINITY;
    int bj = -1;
    for (int d = -kRadius; d <= kRadius; ++d) {
        const int g = i + d;
        if (d == 0 || g < 0 || g >= (int)n) continue;
        const float4 b0 = sLo[threadIdx.x + kRadius + d], b1 = sHi[threadIdx.x + kRadius + d];
        float area = halfArea(fmaxf(a1.x, b1.x) - fminf(a0.x, b0.x), fmaxf(a1.y, b1.y) - fminf(a0.y, b0.y),
                              fmaxf(a1.z, b1.z) - fminf(a0.z, b0.z));
        if (!(area == area)) area = INFINITY;  // 0 * inf
        if (bj < 0 || area < best) {  // ties keep the smaller index: the globally best pair is always mutual
            best = area;
            bj = g;
        }
    }
    nn[i] = (uint32_t)bj;
}

__global__ __launch_bounds__(kBT) void mergeKernel(Node2D* __restrict__ nodes, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ clusters,
                                                   const uint32_t* __restrict__ nn, uint32_t n, uint32_t* __restrict__ next,
                                                   uint8_t* __restrict__ keep, Counters* __restrict__ counters) {
    const uint32_t i = blockIdx.x * kBT + threadIdx.x;
    bool create = false;
    uint32_t j = 0, ci = 0;
    if (i < n) {
        j = nn[i];
        ci = clusters[i];
        const bool mutual = j < n && nn[j] == i;
        create = mutual && i < j;
        keep[i] = (!mutual || i < j) ? 1 : 0;
        if (!create) next[i] = ci;
    }
    const unsigned long long m = __ballot(create);
    uint32_t base = 0;
    const int lane = __lane_id();
    const int leader = m ? __ffsll((long long)m) - 1 : 0;
    if (m && lane == leader) base = atomicAdd(&counters[kCntNode2].v[0], (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (create) {
        const uint32_t id = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t cj = clusters[j];
        const float4* A = (const float4*)&nodes[ci];
        const float4* B = (const float4*)&nodes[cj];
        const float4 a0 = A[0], a1 = A[1], b0 = B[0], b1 = B[1];
        float4* out = (float4*)&nodes[id];
        out[0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), __uint_as_float(ci));
        out[1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __uint_as_float(cj));
        cnt[id] = cnt[ci] + cnt[cj];
        next[i] = id;
    }
}

// The last clusters (n <= kTailClusters) are merged down to the root by ONE block with the cluster array in LDS: with
// few clusters left an iteration still merges only a fraction of them (45 of the 58 iterations of a 262 k triangle
// scene run below 1000 clusters), and each global iteration costs five launches and a host read-back.
constexpr int kTailClusters = 1024;
constexpr uint32_t kTopClusters = 1024;  // clusters handed to the host's SAH for the upper levels (0 = PLOC all the way)
__global__ __launch_bounds__(kTailClusters) void plocTailKernel(Node2D* __restrict__ nodes, uint32_t* __restrict__ cnt, uint32_t* __restrict__ clusters,
                                                                uint32_t n, Counters* __restrict__ counters, int kRadius) {
    __shared__ float4 sLo[kTailClusters], sHi[kTailClusters];
    __shared__ uint32_t sId[kTailClusters], sNn[kTailClusters], sWave[kTailClusters / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t id = 0;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (tid < (int)n) {
        id = clusters[tid];
        const float4* nd = (const float4*)&nodes[id];
        lo = nd[0];
        hi = nd[1];
    }
    const uint32_t nodeBase = counters[kCntNode2].v[0];
    uint32_t created = 0;
    while (n > 1) {
        sLo[tid] = lo;
        sHi[tid] = hi;
        sId[tid] = id;
        __syncthreads();
        if (tid < (int)n) {
            float best = INFINITY;
            int bj = -1;
            for (int d = -kRadius; d <= kRadius; ++d) {
                const int g = tid + d;
                if (d == 0 || g < 0 || g >= (int)n) continue;
                const float4 b0 = sLo[g], b1 = sHi[g];
                float area = halfArea(fmaxf(hi.x, b1.x) - fminf(lo.x, b0.x), fmaxf(hi.y, b1.y) - fminf(lo.y, b0.y), fmaxf(hi.z, b1.z) - fminf(lo.z, b0.z));
                if (!(area == area)) area = INFINITY;
                if (bj < 0 || area < best) {
                    best = area;
                    bj = g;
                }
            }
            sNn[tid] = (uint32_t)bj;
        }
        __syncthreads();
        bool create = false, keep = false;
        uint32_t j = 0;
        if (tid < (int)n) {
            j = sNn[tid];
            const bool mutual = sNn[j] == (uint32_t)tid;
            create = mutual && (uint32_t)tid < j;
            keep = !mutual || (uint32_t)tid < j;
        }
        // one block scan for both: low half counts created nodes, high half kept clusters
        const uint32_t mine = (create ? 1u : 0u) | (keep ? 0x10000u : 0u);
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < kTailClusters / 64; ++w) {
            const uint32_t v = sWave[w];
            if (w < wave) before += v;
            total += v;
        }
        const uint32_t excl = before + incl - mine;
        if (create) {
            const uint32_t newId = nodeBase + created + (excl & 0xFFFFu), cj = sId[j];
            const float4 b0 = sLo[j], b1 = sHi[j];
            lo = make_float4(fminf(lo.x, b0.x), fminf(lo.y, b0.y), fminf(lo.z, b0.z), __uint_as_float(id));
            hi = make_float4(fmaxf(hi.x, b1.x), fmaxf(hi.y, b1.y), fmaxf(hi.z, b1.z), __uint_as_float(cj));
            float4* out = (float4*)&nodes[newId];
            out[0] = lo;
            out[1] = hi;
            cnt[newId] = cnt[id] + cnt[cj];
            id = newId;
        }
        created += total & 0xFFFFu;
        __syncthreads();  // every read of this round's LDS state is done
        if (keep) {
            const uint32_t p = excl >> 16;
            sLo[p] = lo;
            sHi[p] = hi;
            sId[p] = id;
        }
        __syncthreads();
        n = total >> 16;
        if (tid < (int)n) {
            lo = sLo[tid];
            hi = sHi[tid];
            id = sId[tid];
        }
        __syncthreads();
    }
    if (tid == 0) {
        clusters[0] = id;
        counters[kCntNode2].v[0] = nodeBase + created;
    }
}

// boxes and triangle counts of the remaining clusters, for the host's SAH over the upper levels (buildTopSAH)
__global__ __launch_bounds__(kBT) void gatherClustersKernel(const Node2D* __restrict__ nodes, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ clusters,
                                                            uint32_t n, float* __restrict__ boxes6, uint32_t* __restrict__ weights) {
    const uint32_t i = blockIdx.x * kBT + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = clusters[i];
    const Node2D& nd = nodes[id];
    for (int k = 0; k < 3; ++k) {
        boxes6[6 * i + k] = nd.mn[k];
        boxes6[6 * i + 3 + k] = nd.mx[k];
    }
    weights[i] = cnt[id];
}

// ---- collapse --------------------------------------------------------------------------------------
// The host's makeNode4Q (okrt_fast_build.cpp), same arithmetic in double.
__device__ void makeNode4QDevice(const float (*bmn)[3], const float (*bmx)[3], const uint32_t* refs, int n, uint4* out) {
    Node4Q nd;
    nd.pad0 = nd.pad1 = 0u;
    nd.qlo[0] = nd.qlo[1] = nd.qlo[2] = 0u;
    nd.qhi[0] = nd.qhi[1] = nd.qhi[2] = 0u;
    uint32_t exps = 0;
    double org[3], step[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float fl = FLT_MAX, fh = -FLT_MAX;
        for (int i = 0; i < n; ++i) { fl = fminf(fl, bmn[i][a]); fh = fmaxf(fh, bmx[i][a]); }
        const double lo = fl, hi = fh;
        const double pad = fmax(fmax((hi - lo) * (1.0 / 512.0), fmax(fabs(lo), fabs(hi)) * 4e-6), 1e-30);
        float o = (float)(lo - pad);
        if ((double)o > lo - pad) o = nextafterf(o, -INFINITY);
        const double ext = (hi + pad) - (double)o;
        int k;
        frexp(ext / 255.0, &k);
        k = min(max(k, -126), 127);
        while ((double)o + 255.0 * ldexp(1.0, k) < hi && k < 127) ++k;
        nd.origin[a] = o;
        org[a] = o;
        step[a] = ldexp(1.0, k);
        exps |= (uint32_t)(k + 127) << (8 * a);
    }
    nd.exps = exps;
    for (int i = 0; i < 4; ++i) {
        if (i >= n) {
            nd.child[i] = kEmptyRef;
#pragma unroll
            for (int a = 0; a < 3; ++a) nd.qlo[a] |= 255u << (8 * i);
            continue;
        }
        nd.child[i] = refs[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double ql = floor((bmn[i][a] - org[a]) / step[a]);
            double qh = ceil((bmx[i][a] - org[a]) / step[a]);
            ql = fmin(fmax(ql, 0.0), 255.0);
            qh = fmin(fmax(qh, 0.0), 255.0);
            nd.qlo[a] |= (uint32_t)ql << (8 * i);
            nd.qhi[a] |= (uint32_t)qh << (8 * i);
        }
    }
    const uint4* src = (const uint4*)&nd;
    out[0] = src[0]; out[1] = src[1]; out[2] = src[2]; out[3] = src[3];
}

__device__ __forceinline__ void emitTri(const float4* __restrict__ prims, uint32_t prim, float4* __restrict__ tris, uint32_t slot) {
    const float4 p0 = prims[3 * (size_t)prim], p1 = prims[3 * (size_t)prim + 1], p2 = prims[3 * (size_t)prim + 2];
    tris[3 * (size_t)slot] = make_float4(p2.x, p2.y, p2.z, p0.w);                                 // p2, srcTri
    tris[3 * (size_t)slot + 1] = make_float4(p0.x - p2.x, p0.y - p2.y, p0.z - p2.z, p1.w);         // E1 = p0 - p2, instance
    tris[3 * (size_t)slot + 2] = make_float4(p1.x - p2.x, p1.y - p2.y, p1.z - p2.z, 0.f);          // E2 = p1 - p2
}

// The host build's leaf rule (okrt_fast_build.cpp): a subtree of <= kLeafTris triangles becomes a leaf when testing them
// all costs no more than one more node + its two halves (triangle test = node visit = 1).
__device__ __forceinline__ bool leafLike(const Node2D* __restrict__ nodes, const uint32_t* __restrict__ cnt, uint32_t id, uint32_t c) {
    if (c > kLeafTris) return false;
    if (c <= 1u) return true;
    const Node2D& n = nodes[id];
    const Node2D& l = nodes[n.left];
    const Node2D& r = nodes[n.right];
    const float A = halfArea(n.mx[0] - n.mn[0], n.mx[1] - n.mn[1], n.mx[2] - n.mn[2]);
    const float split = halfArea(l.mx[0] - l.mn[0], l.mx[1] - l.mn[1], l.mx[2] - l.mn[2]) * (float)cnt[n.left] +
                        halfArea(r.mx[0] - r.mn[0], r.mx[1] - r.mn[1], r.mx[2] - r.mn[2]) * (float)cnt[n.right];
    return (float)c * A <= A + split;
}

// One queue item = (Node4Q slot, BVH2 node that is not leafLike).
__global__ __launch_bounds__(kBT) void collapseKernel(const Node2D* __restrict__ nodes, const uint32_t* __restrict__ cnt, uint32_t numPrims,
                                                      const float4* __restrict__ prims, const uint2* __restrict__ queueIn, uint32_t numIn,
                                                      uint2* __restrict__ queueOut, Counters* __restrict__ counters, uint4* __restrict__ outNodes,
                                                      float4* __restrict__ outTris) {
    const uint32_t t = blockIdx.x * kBT + threadIdx.x;
    const bool active = t < numIn;
    uint32_t kids[4] = {0, 0, 0, 0}, kc[4] = {0, 0, 0, 0};
    bool kl[4] = {false, false, false, false};  // kid is emitted as a leaf
    float bmn[4][3], bmx[4][3];
    int n = 0;
    uint32_t slot = 0, needNodes = 0, needTris = 0;
    if (active) {
        const uint2 item = queueIn[t];
        slot = item.x;
        kids[0] = nodes[item.y].left;
        kids[1] = nodes[item.y].right;
        n = 2;
        for (int i = 0; i < 2; ++i) {
            kc[i] = cnt[kids[i]];
            kl[i] = leafLike(nodes, cnt, kids[i], kc[i]);
        }
        while (n < 4) {  // open the inner child with the largest area
            int pick = -1;
            float best = -1.f;
            for (int i = 0; i < n; ++i) {
                if (kl[i]) continue;
                const Node2D& c = nodes[kids[i]];
                const float ar = halfArea(c.mx[0] - c.mn[0], c.mx[1] - c.mn[1], c.mx[2] - c.mn[2]);
                if (ar > best) { best = ar; pick = i; }
            }
            if (pick < 0) break;
            const uint32_t l = nodes[kids[pick]].left, r = nodes[kids[pick]].right;
            kids[pick] = l; kc[pick] = cnt[l]; kl[pick] = leafLike(nodes, cnt, l, kc[pick]);
            kids[n] = r; kc[n] = cnt[r]; kl[n] = leafLike(nodes, cnt, r, kc[n]);
            ++n;
        }
        for (int i = 0; i < n; ++i) {
            const Node2D& c = nodes[kids[i]];
            for (int a = 0; a < 3; ++a) { bmn[i][a] = c.mn[a]; bmx[i][a] = c.mx[a]; }
            if (kl[i]) needTris += kc[i]; else needNodes += 1;
        }
    }
    uint32_t nodeBase = waveAlloc(&counters[kCntNode4].v[0], needNodes);
    uint32_t triBase = waveAlloc(&counters[kCntTri].v[0], needTris);
    const uint32_t queueBase = waveAlloc(&counters[kCntQueue].v[0], needNodes);
    if (!active) return;
    uint32_t refs[4] = {kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef};
    uint32_t q = queueBase;
    for (int i = 0; i < n; ++i) {
        if (kl[i]) {
            const uint32_t first = triBase;
            uint32_t st[8];
            int sp = 0;
            st[sp++] = kids[i];
            while (sp) {
                const uint32_t id = st[--sp];
                if (id < numPrims) {
                    emitTri(prims, id, outTris, triBase++);
                } else {
                    st[sp++] = nodes[id].right;
                    st[sp++] = nodes[id].left;
                }
            }
            refs[i] = kLeafBit | ((kc[i] - 1u) << 27) | first;
        } else {
            refs[i] = nodeBase;
            queueOut[q++] = make_uint2(nodeBase, kids[i]);
            ++nodeBase;
        }
    }
    makeNode4QDevice(bmn, bmx, refs, n, outNodes + 4 * (size_t)slot);
}

// a scene of <= kLeafTris triangles: one leaf under the root
__global__ void singleLeafKernel(const Node2D* __restrict__ nodes, const uint32_t* __restrict__ cnt, uint32_t root, uint32_t numPrims,
                                 const float4* __restrict__ prims, uint4* __restrict__ outNodes, float4* __restrict__ outTris) {
    if (threadIdx.x || blockIdx.x) return;
    float bmn[1][3], bmx[1][3];
    for (int a = 0; a < 3; ++a) { bmn[0][a] = nodes[root].mn[a]; bmx[0][a] = nodes[root].mx[a]; }
    uint32_t st[8], out = 0;
    int sp = 0;
    st[sp++] = root;
    while (sp) {
        const uint32_t id = st[--sp];
        if (id < numPrims) emitTri(prims, id, outTris, out++);
        else { st[sp++] = nodes[id].right; st[sp++] = nodes[id].left; }
    }
    const uint32_t ref = kLeafBit | ((cnt[root] - 1u) << 27);
    makeNode4QDevice(bmn, bmx, &ref, 1, outNodes);
}

__global__ void emptyRootKernel(uint4* outNodes) {
    if (threadIdx.x || blockIdx.x) return;
    makeNode4QDevice(nullptr, nullptr, nullptr, 0, outNodes);
}

bool isIdentity(const float* m) {
    static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return std::memcmp(m, I, sizeof(I)) == 0;
}

#define OKRT_BUILD_HIP(call)                                                              \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return false;                                                                 \
        }                                                                                 \
    } while (0)

}  // namespace

struct DeviceBuilder {
    // scratch, reused between builds
    DeviceArray<BakeRange> ranges;
    DeviceArray<float4> prims, cent;
    DeviceArray<Node2D> nodes;
    DeviceArray<uint32_t> cnt, valsA, valsB, nn, next;
    DeviceArray<unsigned long long> keysA, keysB;
    DeviceArray<uint8_t> keep, tmp;
    DeviceArray<float> topBoxes;
    DeviceArray<uint2> queueA, queueB;
    DeviceArray<Counters> counters;
    uint32_t* pinned = nullptr;  // host-visible read-back words
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per mesh BVH root: the leaf triangle ranges (merged); valid until t0/t2 change
    std::unordered_map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> leafRanges;
    // staging for the SAH-built upper levels
    std::vector<float> hostBoxes;
    std::vector<uint32_t> hostWeights, hostClusters, hostCounts;
    std::vector<Node2D> hostNodes;
    ~DeviceBuilder() {
        ranges.release(); prims.release(); cent.release(); nodes.release(); cnt.release(); valsA.release(); valsB.release();
        nn.release(); next.release(); keysA.release(); keysB.release(); keep.release(); tmp.release(); queueA.release();
        queueB.release(); counters.release(); topBoxes.release();
        if (pinned) hipHostFree(pinned);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
    }
};

DeviceBuilder* deviceBuilderCreate() { return new DeviceBuilder(); }
void deviceBuilderDestroy(DeviceBuilder* b) { delete b; }

bool buildFastDevice(DeviceBuilder& B, const FastInputs& in, const FastDeviceTables& dev, hipStream_t stream, FastDeviceResult& out, std::string& err) {
    if (in.geometryChanged) B.leafRanges.clear();
    if (!B.pinned) OKRT_BUILD_HIP(hipHostMalloc((void**)&B.pinned, 64 * sizeof(uint32_t)));
    if (!B.ev0) {
        OKRT_BUILD_HIP(hipEventCreate(&B.ev0));
        OKRT_BUILD_HIP(hipEventCreate(&B.ev1));
    }

    // ---- host: which (instance, triangle range)s the reference shader can reach ----------------------
    std::vector<BakeRange> ranges;
    uint32_t total = 0;
    bool allIdentity = true;
    if (in.numOct) {
        std::vector<uint8_t> seen(in.numMeshes, 0);
        std::vector<uint32_t> stack{0};
        while (!stack.empty()) {
            const OctNodeRec& n = in.oct[stack.back()];
            stack.pop_back();
            for (uint32_t i = n.meshesStart; i < n.meshesEnd && i < in.numMeshes; ++i) {
                if (seen[i]) continue;
                seen[i] = 1;
                const MeshRec& m = in.meshes[i];
                auto it = B.leafRanges.find(m.bvhNodeStartIdx);
                if (it == B.leafRanges.end()) {
                    std::vector<std::pair<uint32_t, uint32_t>> leaves;
                    std::vector<uint32_t> st{m.bvhNodeStartIdx};
                    while (!st.empty()) {
                        const BvhNodeRec& b = in.bvh[st.back()];
                        st.pop_back();
                        if (b.firstChild != kInvalid) {
                            st.push_back(b.firstChild);
                            st.push_back(b.firstChild + 1);
                        } else if (b.triStart != kInvalid && b.triEnd > b.triStart) {
                            leaves.emplace_back(b.triStart, b.triEnd);
                        }
                    }
                    std::sort(leaves.begin(), leaves.end());
                    std::vector<std::pair<uint32_t, uint32_t>> merged;
                    for (auto& l : leaves) {
                        if (!merged.empty() && l.first == merged.back().second) merged.back().second = l.second;
                        else merged.push_back(l);  // overlapping leaves stay separate: the shader would test those triangles twice, too
                    }
                    it = B.leafRanges.emplace(m.bvhNodeStartIdx, std::move(merged)).first;
                }
                const bool ident = isIdentity(m.M);
                allIdentity = allIdentity && ident;
                for (auto& l : it->second) {
                    if ((uint64_t)total + (l.second - l.first) > 0x07FFFFFFull) {
                        err = "more than 2^27 triangle instances";
                        return false;
                    }
                    ranges.push_back({i | (ident ? kIdentityBit : 0u), l.first, l.second - l.first, total});
                    total += l.second - l.first;
                }
            }
            for (uint32_t c = 0; c < n.numChildren; ++c) stack.push_back(n.firstChild + c);
        }
    }
    out.allIdentity = allIdentity;
    out.numTris = 0;
    out.numNodes = 1;
    out.maxDepth = 0;
    out.buildMs = 0.f;

    const size_t N = total;
    if (!out.nodes.need(std::max<size_t>(N, 1) * 4, err) || !out.tris.need(std::max<size_t>(N, 1) * 3, err)) return false;
    OKRT_BUILD_HIP(hipEventRecord(B.ev0, stream));
    if (!N) {
        emptyRootKernel<<<1, 64, 0, stream>>>(out.nodes.p);
        OKRT_BUILD_HIP(hipEventRecord(B.ev1, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        return true;
    }
    if (!B.ranges.need(ranges.size(), err, true) || !B.prims.need(3 * N, err, true) || !B.cent.need(N, err, true) || !B.nodes.need(2 * N, err, true) ||
        !B.cnt.need(2 * N, err, true) || !B.valsA.need(N, err, true) || !B.valsB.need(N, err, true) || !B.nn.need(N, err, true) || !B.next.need(N, err, true) ||
        !B.keysA.need(N, err, true) || !B.keysB.need(N, err, true) || !B.keep.need(N, err, true) || !B.queueA.need(N, err, true) || !B.queueB.need(N, err, true) ||
        !B.counters.need(kNumCnt, err, true))
        return false;
    size_t sortBytes = 0, selBytes = 0;
    OKRT_BUILD_HIP(rocprim::radix_sort_pairs(nullptr, sortBytes, B.keysA.p, B.keysB.p, B.valsA.p, B.valsB.p, N, 0u, 64u, stream));
    OKRT_BUILD_HIP(rocprim::select(nullptr, selBytes, B.next.p, B.keep.p, B.valsA.p, &B.counters.p[kCntSelected].v[0], N, stream));
    if (!B.tmp.need(std::max(sortBytes, selBytes), err)) return false;

    Counters init[kNumCnt];
    std::memset(init, 0, sizeof(init));
    init[kCntNode2].v[0] = (uint32_t)N;  // inner BVH2 nodes are handed out from N upward
    init[kCntNode4].v[0] = 1;            // slot 0 is the root
    for (int k = 0; k < 3; ++k) {
        init[kCntBounds].v[k] = 0xFFFFFFFFu;
        init[kCntBounds].v[3 + k] = 0u;
    }
    OKRT_BUILD_HIP(hipMemcpyAsync(B.counters.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    OKRT_BUILD_HIP(hipMemcpyAsync(B.ranges.p, ranges.data(), ranges.size() * sizeof(BakeRange), hipMemcpyHostToDevice, stream));
    const uint32_t blocksN = (uint32_t)((N + kBT - 1) / kBT);
    bakeKernel<<<blocksN, kBT, 0, stream>>>(B.ranges.p, (uint32_t)ranges.size(), (uint32_t)N, dev.triPos, dev.meshes, B.prims.p, B.nodes.p,
                                            B.cnt.p, B.cent.p, B.counters.p);
    mortonKernel<<<blocksN, kBT, 0, stream>>>(B.cent.p, (uint32_t)N, B.counters.p, B.keysA.p, B.valsA.p);
    size_t bytes = B.tmp.cap;
    OKRT_BUILD_HIP(rocprim::radix_sort_pairs(B.tmp.p, bytes, B.keysA.p, B.keysB.p, B.valsA.p, B.valsB.p, N, 0u, 64u, stream));
    OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, &B.counters.p[kCntInvalid].v[0], 4, hipMemcpyDeviceToHost, stream));
    OKRT_BUILD_HIP(hipStreamSynchronize(stream));  // `ranges`/`init` may go; the invalid count is here
    uint32_t n = (uint32_t)N - B.pinned[0];  // non-finite triangles sort last (key ~0) and are never clustered
    if (!n) {
        emptyRootKernel<<<1, 64, 0, stream>>>(out.nodes.p);
        OKRT_BUILD_HIP(hipEventRecord(B.ev1, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        return true;
    }

    // ---- PLOC: clusters = node ids in Morton order (a leaf's id is its primitive index) ---------------
    static const int radius = std::min(std::max(getenv("OKRT_BUILD_RADIUS") ? atoi(getenv("OKRT_BUILD_RADIUS")) : kRadius, 1), kMaxRadius);
    uint32_t* cur = B.valsB.p;
    uint32_t* alt = B.valsA.p;
    // clustering stops at kTopClusters clusters: the levels above them are built top-down by SAH on the host (a few
    // thousand boxes, < 1 ms) — PLOC's Morton window cannot see the large occluders that matter up there
    static const uint32_t topClusters = getenv("OKRT_BUILD_TOP") ? (uint32_t)atoi(getenv("OKRT_BUILD_TOP")) : kTopClusters;
    const uint32_t stopAt = std::max<uint32_t>(topClusters, (uint32_t)kTailClusters);
    for (int iter = 0; n > stopAt; ++iter) {
        if (iter > 4096) {
            err = "device build: clustering did not converge";
            return false;
        }
        const uint32_t blocks = (n + kBT - 1) / kBT;
        nearestKernel<<<blocks, kBT, 0, stream>>>(B.nodes.p, cur, n, B.nn.p, radius);
        mergeKernel<<<blocks, kBT, 0, stream>>>(B.nodes.p, B.cnt.p, cur, B.nn.p, n, B.next.p, B.keep.p, B.counters.p);
        bytes = B.tmp.cap;
        OKRT_BUILD_HIP(rocprim::select(B.tmp.p, bytes, B.next.p, B.keep.p, alt, &B.counters.p[kCntSelected].v[0], (size_t)n, stream));
        OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, &B.counters.p[kCntSelected].v[0], 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        const uint32_t m = B.pinned[0];
        if (m >= n || !m) {
            err = "device build: clustering made no progress";
            return false;
        }
        n = m;
        std::swap(cur, alt);
    }
    uint32_t root = 0;
    if (topClusters && n > 1) {
        if (!B.topBoxes.need((size_t)n * 6, err, true)) return false;
        float* dBoxes = B.topBoxes.p;  // (B.nn is free now: it takes the counts)
        gatherClustersKernel<<<(n + kBT - 1) / kBT, kBT, 0, stream>>>(B.nodes.p, B.cnt.p, cur, n, dBoxes, B.nn.p);
        B.hostBoxes.resize((size_t)n * 6);
        B.hostWeights.resize(n);
        B.hostClusters.resize(n);
        OKRT_BUILD_HIP(hipMemcpyAsync(B.hostBoxes.data(), dBoxes, (size_t)n * 24, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipMemcpyAsync(B.hostWeights.data(), B.nn.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipMemcpyAsync(B.hostClusters.data(), cur, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, &B.counters.p[kCntNode2].v[0], 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        const uint32_t nodeBase = B.pinned[0];
        std::vector<TopNode> top;
        const uint32_t topRoot = buildTopSAH(B.hostBoxes.data(), B.hostWeights.data(), n, top);
        auto ref = [&](uint32_t r) { return r < n ? B.hostClusters[r] : nodeBase + (r - n); };
        B.hostNodes.resize(top.size());
        B.hostCounts.resize(top.size());
        for (size_t k = 0; k < top.size(); ++k) {
            Node2D& nd = B.hostNodes[k];
            for (int a = 0; a < 3; ++a) { nd.mn[a] = top[k].mn[a]; nd.mx[a] = top[k].mx[a]; }
            nd.left = ref(top[k].left);
            nd.right = ref(top[k].right);
            B.hostCounts[k] = top[k].count;
        }
        OKRT_BUILD_HIP(hipMemcpyAsync(B.nodes.p + nodeBase, B.hostNodes.data(), top.size() * sizeof(Node2D), hipMemcpyHostToDevice, stream));
        OKRT_BUILD_HIP(hipMemcpyAsync(B.cnt.p + nodeBase, B.hostCounts.data(), top.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));  // the staging vectors may be reused by the next build
        root = ref(topRoot);
    } else {
        if (n > 1) plocTailKernel<<<1, kTailClusters, 0, stream>>>(B.nodes.p, B.cnt.p, cur, n, B.counters.p, radius);
        OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, cur, 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        root = B.pinned[0];
    }

    // ---- collapse ----------------------------------------------------------------------------------------
    OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned + 2, &B.cnt.p[root], 4, hipMemcpyDeviceToHost, stream));
    OKRT_BUILD_HIP(hipStreamSynchronize(stream));
    const uint32_t rootCount = B.pinned[2];
    if (rootCount <= kLeafTris) {
        singleLeafKernel<<<1, 64, 0, stream>>>(B.nodes.p, B.cnt.p, root, (uint32_t)N, B.prims.p, out.nodes.p, out.tris.p);
        out.numTris = rootCount;
        out.maxDepth = 1;
        OKRT_BUILD_HIP(hipEventRecord(B.ev1, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        OKRT_BUILD_HIP(hipEventElapsedTime(&out.buildMs, B.ev0, B.ev1));
        return true;
    }
    const uint2 first = make_uint2(0u, root);
    OKRT_BUILD_HIP(hipMemcpyAsync(B.queueA.p, &first, sizeof(first), hipMemcpyHostToDevice, stream));
    uint2* qin = B.queueA.p;
    uint2* qout = B.queueB.p;
    uint32_t numIn = 1, levels = 0;
    while (numIn) {
        OKRT_BUILD_HIP(hipMemsetAsync(&B.counters.p[kCntQueue].v[0], 0, 4, stream));
        collapseKernel<<<(numIn + kBT - 1) / kBT, kBT, 0, stream>>>(B.nodes.p, B.cnt.p, (uint32_t)N, B.prims.p, qin, numIn, qout, B.counters.p,
                                                                   out.nodes.p, out.tris.p);
        OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, &B.counters.p[kCntQueue].v[0], 4, hipMemcpyDeviceToHost, stream));
        OKRT_BUILD_HIP(hipStreamSynchronize(stream));
        numIn = B.pinned[0];
        std::swap(qin, qout);
        ++levels;
    }
    OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned, &B.counters.p[kCntNode4].v[0], 4, hipMemcpyDeviceToHost, stream));
    OKRT_BUILD_HIP(hipMemcpyAsync(B.pinned + 1, &B.counters.p[kCntTri].v[0], 4, hipMemcpyDeviceToHost, stream));
    OKRT_BUILD_HIP(hipEventRecord(B.ev1, stream));
    OKRT_BUILD_HIP(hipStreamSynchronize(stream));
    OKRT_BUILD_HIP(hipGetLastError());
    out.numNodes = B.pinned[0];
    out.numTris = B.pinned[1];
    out.maxDepth = levels + 1;  // node levels + the leaf level
    OKRT_BUILD_HIP(hipEventElapsedTime(&out.buildMs, B.ev0, B.ev1));
    return true;
}

}  // namespace okrt
