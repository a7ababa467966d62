// l1_gather_rate — how many per-lane 16-byte gathers a CU's vector L1 retires per cycle, by shape (MI355X, gfx950).
// The bounce-ray traversal kernel gathers 64-byte nodes (4 x dwordx4 per lane) and 48-byte triangles (3 x dwordx4) from tables that sit in
// L2; this tool measures the same shapes in isolation, from an L2-resident and a larger table:
//   rec64      every lane its own random 64-byte record, 4 x dwordx4                        (the node fetch)
//   rec48      every lane its own random 48-byte record, 3 x dwordx4                        (the triangle fetch)
//   rec64quad  the four lanes of a quad fetch ONE record together, a 16-byte piece each, four records per quad and round
//              (same bytes and instructions as rec64, but each instruction touches 16 lines instead of 64)
//   rec16      every lane one random 16-byte piece                                           (the per-lane rate itself)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/calib/l1_gather_rate tools/calib/l1_gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void gatherKernel(const uint4* __restrict__ table, uint32_t records, uint32_t rounds, uint32_t* __restrict__ sink) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    uint32_t acc = 0, state = mix(gid + 1);
    for (uint32_t r = 0; r < rounds; ++r) {
        state = state * 1664525u + 1013904223u;
        const uint32_t mine = mix(state) % records;
        if (SHAPE == 0) {  // rec64
            const uint4* p = table + (size_t)mine * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
        } else if (SHAPE == 1) {  // rec48
            const uint4* p = table + (size_t)mine * 3;
            const uint4 a = p[0], b = p[1], c = p[2];
            acc ^= a.x ^ b.y ^ c.z;
        } else if (SHAPE == 2) {  // rec64quad: instruction k fetches the record chosen by quad lane k, lane j its piece j
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t idx = __shfl(mine, (lane & ~3u) + k, 64);
                v[k] = table[(size_t)idx * 4 + (lane & 3u)];
            }
            acc ^= v[0].x ^ v[1].y ^ v[2].z ^ v[3].w;
        } else {  // rec16
            const uint4 a = table[(size_t)mine * 4 + (state >> 30)];
            acc ^= a.x;
        }
    }
    if (acc == 0x12345678u) sink[gid] = acc;
}

template <int SHAPE>
static double run(const uint4* table, uint32_t records, uint32_t rounds, uint32_t blocks, uint32_t* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gatherKernel<SHAPE><<<blocks, 256>>>(table, records, rounds, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        gatherKernel<SHAPE><<<blocks, 256>>>(table, records, rounds, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const uint32_t blocks = cus * 7, rounds = 2000;  // 7 blocks of 4 waves per CU = 7 waves per SIMD, the traversal kernel's occupancy
    uint32_t* sink; CK(hipMalloc((void**)&sink, (size_t)blocks * 256 * 4));
    printf("device %s, %d CUs, %.2f GHz nominal; %u blocks x 256 threads x %u rounds\n", prop.name, cus, ghz, blocks, rounds);
    printf("%-10s %8s %10s %12s %14s %16s\n", "shape", "table", "ms", "GB/s", "Glane-loads/s", "lane-loads/clk/CU");
    for (size_t mib : {2, 4, 16, 64, 512}) {
        const size_t bytes = mib << 20;
        uint4* table; CK(hipMalloc((void**)&table, bytes));
        std::vector<uint32_t> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
        CK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
        const double lanes = (double)blocks * 256 * rounds;
        struct { const char* name; double ms; int loads; int recBytes; } rows[4] = {
            {"rec64", run<0>(table, (uint32_t)(bytes / 64), rounds, blocks, sink), 4, 64},
            {"rec48", run<1>(table, (uint32_t)(bytes / 48), rounds, blocks, sink), 3, 48},
            {"rec64quad", run<2>(table, (uint32_t)(bytes / 64), rounds, blocks, sink), 4, 64},
            {"rec16", run<3>(table, (uint32_t)(bytes / 64), rounds, blocks, sink), 1, 16},
        };
        for (auto& r : rows) {
            const double loadsPerS = lanes * r.loads / (r.ms * 1e-3);
            printf("%-10s %6zuMi %10.3f %12.1f %14.1f %16.3f\n", r.name, mib, r.ms, lanes * r.recBytes / (r.ms * 1e-3) * 1e-9, loadsPerS * 1e-9, loadsPerS / (ghz * 1e9) / cus);
        }
        CK(hipFree(table));
    }
    return 0;
}
