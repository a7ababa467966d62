// issue_rate — how many wave-instructions per cycle one SIMD of an MI355X (gfx950) starts, by kind and by resident waves.
// The traversal kernels are "issue-bound" by their own counters (profiles/r05_pmc_*: VALU + SALU instructions per SIMD quad-cycle); this
// tool measures what the denominators of that statement are, in isolation: a loop of INDEPENDENT instructions of one kind (or an
// alternation of two kinds), W waves per SIMD, every CU busy, no memory traffic.
//   fma        v_fma_f32                 (the slab arithmetic, Moeller-Trumbore)
//   cndmask    v_cndmask_b32             (the select-based child sort)
//   minmax     v_max_f32 / v_min_f32     (slab reduction)
//   cvt        v_cvt_f32_ubyte0          (8-bit plane decode)
//   salu       s_add_u32 / s_and_b32     (what the packet kernel's scalar sort and the per-iteration bookkeeping are made of)
//   fma+salu   one of each per pair      (do the vector and the scalar port issue side by side?)
//   cndm_vcc / cndm_sgpr  v_cndmask_b32 selecting by VCC / by an SGPR pair;  cmp_sgpr  v_cmp_*_f32 into an SGPR pair;
//   cswap      the sort network's compare-and-swap (1 v_cmp + 4 v_cndmask);  minmax_u  v_min_u32 / v_max_u32 (packed-key swap)
// Output: instructions per cycle per SIMD = N / (s_memtime delta of the wave that ran longest), and per quad-cycle (x4: the unit of the
// SQ_* counters). build: hipcc --offload-arch=gfx950 -O3 -o tools/calib/issue_rate tools/calib/issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 64;  // instructions (or pairs) per loop trip

template <int KIND>
__global__ __launch_bounds__(1024) void issueKernel(uint32_t trips, unsigned long long* cycles, float* sink) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-9f;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    unsigned long long mask = 0x5555AAAA3333CCCCull ^ blockIdx.x, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pm = {m, m}, pc = {c, c};
    if (KIND == 1 || KIND == 10) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < kUnroll / 8; ++k) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            } else if (KIND == 1) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if (KIND == 2) {
                asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (KIND == 3) {
                asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n"
                             "v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte0 %5, %5\n v_cvt_f32_ubyte0 %6, %6\n v_cvt_f32_ubyte0 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 4) {
                asm volatile("s_add_u32 %0, %0, 3\n s_and_b32 %1, %1, 0xffff\n s_add_u32 %2, %2, 5\n s_and_b32 %3, %3, 0xfff\n"
                             "s_add_u32 %0, %0, 7\n s_and_b32 %1, %1, 0xff\n s_add_u32 %2, %2, 9\n s_and_b32 %3, %3, 0xf"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            } else if (KIND == 6) {  // select through an SGPR pair (what the compiler emits for `c ? a : b` with the condition of a v_cmp_..._e64)
                asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                             "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "s"(mask));
            } else if (KIND == 7) {  // compare into an SGPR pair
                asm volatile("v_cmp_lt_f32 %0, %4, %8\n v_cmp_lt_f32 %1, %5, %8\n v_cmp_lt_f32 %2, %6, %8\n v_cmp_lt_f32 %3, %7, %8\n"
                             "v_cmp_gt_f32 %0, %4, %8\n v_cmp_gt_f32 %1, %5, %8\n v_cmp_gt_f32 %2, %6, %8\n v_cmp_gt_f32 %3, %7, %8"
                             : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(m));
            } else if (KIND == 8) {  // the compare-and-swap of the child sort: one compare, four selects on its result (x2 per block = 10 instructions)
                asm volatile("v_cmp_lt_f32 %8, %1, %0\n v_cndmask_b32 %4, %0, %1, %8\n v_cndmask_b32 %5, %1, %0, %8\n v_cndmask_b32 %6, %2, %3, %8\n v_cndmask_b32 %7, %3, %2, %8\n"
                             "v_cmp_lt_f32 %9, %5, %4\n v_cndmask_b32 %0, %4, %5, %9\n v_cndmask_b32 %1, %5, %4, %9\n v_cndmask_b32 %2, %6, %7, %9\n v_cndmask_b32 %3, %7, %6, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m0), "=&s"(m1));
            } else if (KIND == 10) {  // VOP3 encoding of the VCC select
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                             "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if (KIND == 11) {  // VCC select whose VCC comes from a v_cmp in the same block (the compiler's usual pairing), no self-dependence
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %4, %1, %8, vcc\n v_cndmask_b32 %5, %2, %8, vcc\n v_cndmask_b32 %6, %3, %8, vcc\n"
                             "v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %7, %0, %8, vcc\n v_cndmask_b32 %4, %2, %8, vcc\n v_cndmask_b32 %5, %3, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            } else if (KIND >= 20 && KIND < 40) {
#define OKRT_OP8(op, tail) asm volatile(op " %0, %0, " tail "\n " op " %1, %1, " tail "\n " op " %2, %2, " tail "\n " op " %3, %3, " tail "\n " \
                                        op " %4, %4, " tail "\n " op " %5, %5, " tail "\n " op " %6, %6, " tail "\n " op " %7, %7, " tail \
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
                if (KIND == 20) OKRT_OP8("v_mul_f32", "%8");
                else if (KIND == 21) OKRT_OP8("v_add_f32", "%8");
                else if (KIND == 22) OKRT_OP8("v_sub_f32", "%8");
                else if (KIND == 23) OKRT_OP8("v_fmac_f32", "%8");
                else if (KIND == 24) OKRT_OP8("v_max3_f32", "%8, %9");
                else if (KIND == 25) OKRT_OP8("v_and_b32", "%8");
                else if (KIND == 26) OKRT_OP8("v_add_u32", "%8");
                else if (KIND == 27) OKRT_OP8("v_lshlrev_b32", "%8");
                else if (KIND == 28) OKRT_OP8("v_med3_f32", "%8, %9");
                else if (KIND == 29) OKRT_OP8("v_bfe_u32", "8, 8");
                else if (KIND == 30) OKRT_OP8("v_mul_lo_u32", "%8");
                else if (KIND == 31) OKRT_OP8("v_mul_u32_u24", "%8");
                else if (KIND == 32) OKRT_OP8("v_xor_b32", "%8");
                else if (KIND == 33) OKRT_OP8("v_lshrrev_b32", "%8");
#undef OKRT_OP8
            } else if (KIND == 41) {  // register moves (the compiler's phi copies: 89 of the bounce kernel's 644 static vector instructions)
                asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (KIND == 40) {  // packed fp32: two FMAs per lane and instruction
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            } else if (KIND == 9) {  // the same swap on packed keys: v_min_u32 / v_max_u32 (a key-value sort with the value in the key's low bits)
                asm volatile("v_min_u32 %4, %0, %1\n v_max_u32 %5, %0, %1\n v_min_u32 %6, %2, %3\n v_max_u32 %7, %2, %3\n"
                             "v_min_u32 %0, %4, %6\n v_max_u32 %1, %4, %6\n v_min_u32 %2, %5, %7\n v_max_u32 %3, %5, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else {  // 5: fma + salu interleaved (8 of each per block = 16 instructions)
                asm volatile("v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, 3\n v_fma_f32 %1, %1, %12, %13\n s_and_b32 %9, %9, 0xffff\n"
                             "v_fma_f32 %2, %2, %12, %13\n s_add_u32 %10, %10, 5\n v_fma_f32 %3, %3, %12, %13\n s_and_b32 %11, %11, 0xfff\n"
                             "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, 7\n v_fma_f32 %5, %5, %12, %13\n s_and_b32 %9, %9, 0xff\n"
                             "v_fma_f32 %6, %6, %12, %13\n s_add_u32 %10, %10, 9\n v_fma_f32 %7, %7, %12, %13\n s_and_b32 %11, %11, 0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                             : "v"(m), "v"(c) : "scc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    const float r = p0.x + p1.y + p2.x + p3.y + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 ^ s1 ^ s2 ^ s3) + (float)(m0 ^ m1 ^ m2 ^ m3);
    if (r == 1.2345e-30f) sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void measure(const char* name, double perSlot, int cus, unsigned long long* dCycles, float* sink) {
    const uint32_t trips = 4000;
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        // one block per CU (two of 1024 threads for 8 waves per SIMD): a block of 256 * w threads puts w waves on each SIMD of its CU
        const int threads = 256 * (wavesPerSimd == 8 ? 4 : wavesPerSimd), blocks = cus * (wavesPerSimd == 8 ? 2 : 1);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        issueKernel<KIND><<<blocks, threads>>>(trips, dCycles, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        issueKernel<KIND><<<blocks, threads>>>(trips, dCycles, sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc((size_t)blocks * (threads / 64));
        CK(hipMemcpy(cyc.data(), dCycles, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long worst = 0;
        double mean = 0;
        for (auto c : cyc) { worst = c > worst ? c : worst; mean += (double)c; }
        mean /= (double)cyc.size();
        // a SIMD ran wavesPerSimd waves side by side (mean ~ worst says the blocks were spread evenly), each trips * kUnroll * perSlot instructions
        const double insts = (double)wavesPerSimd * trips * kUnroll * perSlot;
        printf("%-10s %5d %12.3f %14.0f %14.0f %12.3f %12.3f\n", name, wavesPerSimd, ms, mean, (double)worst, insts / (double)worst, 4.0 * insts / (double)worst);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
}

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned long long* dCycles;
    float* sink;
    CK(hipMalloc((void**)&dCycles, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
    CK(hipMalloc((void**)&sink, (size_t)cus * 2 * 1024 * sizeof(float)));
    printf("# %d CUs; unrolled loop of independent instructions, 4000 trips x %d slots; one block of 256*W threads per CU (W waves per SIMD; W = 8: two blocks of 1024)\n", cus, kUnroll);
    printf("# rates = W * instructions per wave / the slowest wave's s_memtime span (shader clocks)\n");
    printf("%-10s %5s %12s %14s %14s %12s %12s\n", "kind", "W", "kernel ms", "mean cyc/wave", "worst cyc/wave", "inst/cyc/SIMD", "inst/quad-cyc");
    measure<0>("fma", 1, cus, dCycles, sink);
    measure<1>("cndm_vcc", 1, cus, dCycles, sink);
    measure<10>("cndm_vcc64", 1, cus, dCycles, sink);
    measure<11>("cmp+cndm_v", 1, cus, dCycles, sink);
    measure<6>("cndm_sgpr", 1, cus, dCycles, sink);
    measure<7>("cmp_sgpr", 1, cus, dCycles, sink);
    measure<8>("cswap", 10.0 / 8.0, cus, dCycles, sink);
    measure<9>("minmax_u", 1, cus, dCycles, sink);
    measure<2>("minmax", 1, cus, dCycles, sink);
    measure<20>("mul_f32", 1, cus, dCycles, sink);
    measure<21>("add_f32", 1, cus, dCycles, sink);
    measure<22>("sub_f32", 1, cus, dCycles, sink);
    measure<23>("fmac_f32", 1, cus, dCycles, sink);
    measure<40>("pk_fma", 1, cus, dCycles, sink);
    measure<24>("max3_f32", 1, cus, dCycles, sink);
    measure<28>("med3_f32", 1, cus, dCycles, sink);
    measure<25>("and_b32", 1, cus, dCycles, sink);
    measure<26>("add_u32", 1, cus, dCycles, sink);
    measure<27>("lshlrev", 1, cus, dCycles, sink);
    measure<29>("bfe_u32", 1, cus, dCycles, sink);
    measure<30>("mul_lo_u32", 1, cus, dCycles, sink);
    measure<31>("mul_u24", 1, cus, dCycles, sink);
    measure<32>("xor_b32", 1, cus, dCycles, sink);
    measure<41>("mov_b32", 1, cus, dCycles, sink);
    measure<33>("lshrrev", 1, cus, dCycles, sink);
    measure<3>("cvt", 1, cus, dCycles, sink);
    measure<4>("salu", 1, cus, dCycles, sink);
    measure<5>("fma+salu", 2, cus, dCycles, sink);
    return 0;
}
