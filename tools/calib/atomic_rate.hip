// atomic_rate — how many device-scope atomics per second ONE address (one 128-byte line) of an MI355X takes, and what a wave
// waits for a returning one. The wavefront pipeline reserves work and output space with one atomicAdd per wave and tile / batch on a
// per-shard cursor (8 shards, one line each: okrt_fast_internal.h kCursorStride); this measures the ceiling of that pattern.
//   same     every wave of the grid adds to ONE word            (what a single global cursor would be)
//   shards   wave w adds to word (xcc & 7) * 32                 (the pipeline's layout: one line per XCD)
//   lines    wave w adds to word (w % 1024) * 32                (no sharing: the memory system's own rate)
// returning = the wave reads the old value (s_waitcnt before the next one: a dependent chain), fire = no return value, no wait.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/calib/atomic_rate tools/calib/atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LAYOUT, bool RETURNING, bool WIDE>
__global__ __launch_bounds__(256) void atomicKernel(uint32_t* words, uint32_t perWave, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t* p = words + (LAYOUT == 0 ? 0u : LAYOUT == 1 ? (xcc & 7u) * 32u : (wave & 1023u) * 32u);
    uint32_t acc = 0;
    if (lane == 0) {
        for (uint32_t i = 0; i < perWave; ++i) {
            if (WIDE) {
                if (RETURNING) acc += (uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(p), 1ull);
                else __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (RETURNING) acc += atomicAdd(p, 64u);
                else __hip_atomic_fetch_add(p, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (acc == 0xFFFFFFFFu) *sink = acc;
    }
}

template <int LAYOUT, bool RETURNING, bool WIDE>
static void run(const char* name, uint32_t* words, uint32_t* sink, int blocks, uint32_t perWave) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(words, 0, 1024 * 32 * 4));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((atomicKernel<LAYOUT, RETURNING, WIDE>), dim3(blocks), dim3(256), 0, 0, words, perWave, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double n = (double)blocks * 4 * perWave;
    const int addresses = LAYOUT == 0 ? 1 : LAYOUT == 1 ? 8 : 1024;
    printf("%-8s %-9s %-6s waves %5d x %4u atomics: %8.3f ms  %8.1f M atomics/s total  %7.2f M/s per address  %7.1f ns per wave and atomic\n", name,
           RETURNING ? "returning" : "fire", WIDE ? "64-bit" : "32-bit", blocks * 4, perWave, best, n / best / 1e3, n / best / 1e3 / addresses,
           best * 1e6 / perWave);
}

int main(int argc, char** argv) {
    int cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    uint32_t *words, *sink;
    CK(hipMalloc(&words, 1024 * 32 * 4));
    CK(hipMalloc(&sink, 4));
    printf("# %s, %d CUs\n", prop.name, cus);
    for (int perCu : {1, 4, 7}) {
        const int blocks = cus * perCu;
        const uint32_t n = perCu == 1 ? 256 : 64;
        run<0, true, false>("same", words, sink, blocks, n);
        run<0, false, false>("same", words, sink, blocks, n);
        run<0, false, true>("same", words, sink, blocks, n);
        run<1, true, false>("shards", words, sink, blocks, n);
        run<1, false, false>("shards", words, sink, blocks, n);
        run<2, true, false>("lines", words, sink, blocks, n);
        run<2, false, false>("lines", words, sink, blocks, n);
    }
    return 0;
}
