// pmc_calibrate.hip — known-byte kernels in THIS repository's memory access patterns, to turn rocprofv3's FETCH_SIZE /
// WRITE_SIZE into bytes per pattern before anything is called HBM-bound (MI355X_MICROARCH.md, "HBM": FETCH_SIZE is only
// calibrated for 16-B-per-lane streaming reads, where it reports half the bytes; other widths are to be calibrated on a
// known byte count in one's own pattern). Every kernel touches each byte of its table exactly ONCE per launch (indices
// are a bijection of the thread index), so the algorithmic bytes of a launch are the table size, whatever the caches do.
//
//   stream16_read     16 B per lane, coalesced (the control: ray queues, hit records, radiance slots)
//   gather64_read     one random 64-byte record per lane as 4 x 16 B (Node4Q visit, shading record)
//   gather48_read     one random 48-byte record per lane as 3 x 16 B (Tri48 test)
//   texel_taps_read   four 4-byte texels of a 2x2 footprint inside one random 64-byte tile per lane (bilinear tap)
//   stream16_write    16 B per lane, coalesced (queue appends)
//   scatter20_write   16 B + 4 B per lane at a random record (the round-2 hit record: float4 + uint32 in two arrays)
//   scatter16_write   16 B per lane at a random record (a packed hit record)
//
// Each runs on a 64 MiB table (fits the 256 MiB Infinity Cache; second launch onward is served from it) and on a 1 GiB
// table (does not). Usage: pmc_calibrate [launches-per-case]; driven by tools/pmc_calibrate.py under rocprofv3.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                              \
        }                                                                          \
    } while (0)

// i -> a permutation of [0, n) for n a power of two (odd multiplier + xor-shift, both bijections mod 2^k)
__device__ __forceinline__ uint32_t scramble(uint32_t i, uint32_t mask) {
    uint32_t x = (i * 2654435761u) & mask;
    x ^= x >> 7;
    x = (x * 40503u + 12345u) & mask;
    return x;
}

__global__ __launch_bounds__(256) void stream16_read(const uint4* __restrict__ t, uint32_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint4 v = t[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void gather64_read(const uint4* __restrict__ t, uint32_t records, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < records; i += gridDim.x * 256) {
        const uint4* p = t + (size_t)scramble(i, records - 1) * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void gather48_read(const uint4* __restrict__ t, uint32_t records, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < records; i += gridDim.x * 256) {
        const uint4* p = t + (size_t)scramble(i, records - 1) * 3;  // (records is a power of two; the table holds 3 x records uint4)
        const uint4 a = p[0], b = p[1], c = p[2];
        acc += a.x ^ b.y ^ c.z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void texel_taps_read(const uint32_t* __restrict__ t, uint32_t tiles, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < tiles; i += gridDim.x * 256) {
        const uint32_t tile = scramble(i, tiles - 1);
        const uint32_t* p = t + (size_t)tile * 16;
        const uint32_t x = tile & 1u, y = (tile >> 1) & 1u;  // a 2x2 footprint at (x..x+1, y..y+1) of the 4x4 tile
        acc += p[y * 4 + x] ^ p[y * 4 + x + 1] ^ p[(y + 1) * 4 + x] ^ p[(y + 1) * 4 + x + 1];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void stream16_write(uint4* __restrict__ t, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) t[i] = make_uint4(i, i + 1, i + 2, i + 3);
}
__global__ __launch_bounds__(256) void scatter20_write(uint4* __restrict__ a, uint32_t* __restrict__ b, uint32_t records) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < records; i += gridDim.x * 256) {
        const uint32_t r = scramble(i, records - 1);
        a[r] = make_uint4(i, i + 1, i + 2, i + 3);
        b[r] = i;
    }
}
__global__ __launch_bounds__(256) void scatter16_write(uint4* __restrict__ a, uint32_t records) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < records; i += gridDim.x * 256) a[scramble(i, records - 1)] = make_uint4(i, i + 1, i + 2, i + 3);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? std::atoi(argv[1]) : 3;
    const size_t sizes[2] = {64ull << 20, 1ull << 30};
    uint4* table = nullptr;
    uint32_t *aux = nullptr, *sink = nullptr;
    CK(hipMalloc((void**)&table, sizes[1] + (sizes[1] >> 2)));  // 1 GiB (+ 25 %: the 48-byte case reads 3/4 of 1.25 GiB ... kept simple: see below)
    CK(hipMalloc((void**)&aux, sizes[1] / 4));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(table, 1, sizes[1] + (sizes[1] >> 2)));
    CK(hipMemset(aux, 1, sizes[1] / 4));
    const dim3 grid(256 * 8), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::printf("case,table_bytes,algorithmic_bytes,launch,ms,GBps\n");
    for (size_t bytes : sizes) {
        const uint32_t n16 = (uint32_t)(bytes / 16), rec64 = (uint32_t)(bytes / 64), tiles = rec64;
        const uint32_t rec48 = (uint32_t)(bytes / 64);  // power of two; 48 B each -> 3/4 of `bytes` are touched
        const uint32_t rec20 = (uint32_t)(bytes / 16);  // 16 B into `table` + 4 B into `aux` per record
        struct Case { const char* name; size_t alg; int id; } cases[] = {
            {"stream16_read", bytes, 0}, {"gather64_read", bytes, 1}, {"gather48_read", (size_t)rec48 * 48, 2}, {"texel_taps_read", (size_t)tiles * 16, 3},
            {"stream16_write", bytes, 4}, {"scatter20_write", (size_t)rec20 * 20, 5}, {"scatter16_write", bytes, 6}};
        for (const Case& c : cases)
            for (int l = 0; l < launches; ++l) {
                CK(hipEventRecord(e0, 0));
                switch (c.id) {
                    case 0: hipLaunchKernelGGL(stream16_read, grid, block, 0, 0, table, n16, sink); break;
                    case 1: hipLaunchKernelGGL(gather64_read, grid, block, 0, 0, table, rec64, sink); break;
                    case 2: hipLaunchKernelGGL(gather48_read, grid, block, 0, 0, table, rec48, sink); break;
                    case 3: hipLaunchKernelGGL(texel_taps_read, grid, block, 0, 0, (const uint32_t*)table, tiles, sink); break;
                    case 4: hipLaunchKernelGGL(stream16_write, grid, block, 0, 0, table, n16); break;
                    case 5: hipLaunchKernelGGL(scatter20_write, grid, block, 0, 0, table, aux, rec20); break;
                    case 6: hipLaunchKernelGGL(scatter16_write, grid, block, 0, 0, table, n16); break;
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::printf("%s,%zu,%zu,%d,%.4f,%.1f\n", c.name, bytes, c.alg, l, ms, (double)c.alg / (ms * 1e-3) / 1e9);
            }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
