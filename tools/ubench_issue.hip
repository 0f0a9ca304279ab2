// Single-wave issue behaviour of v_mad_u64_u32 on gfx950: is one wave per SIMD enough to keep the integer
// multiplier busy?  Every variant forces ONE workgroup per CU with a large dynamic LDS allocation.
// Findings on MI355X (profiles/r01/ubench_issue.jsonl), cycles per wave-instruction at 2.4 GHz:
//   * one wave per SIMD issues ANY VALU instruction only every ~5 cycles (v_add_u32 5.1, v_and 4.8,
//     v_mul_lo 5.3); v_mad_u64_u32 5.9.  Two waves per SIMD: 2.9 / 4.6.  Eight: 2.4 / 4.4.
//   * back-to-back inline-asm v_mad_u64_u32 with the SAME scalar carry-out measure 8.9 cycles — an
//     artefact: the compiler's hazard recogniser puts an s_nop between asm blocks that clobber the same
//     SGPR pair (compiled code, which writes s[6:7] everywhere, gets no such nop).
//   * cheap VALU instructions do not hide behind the multiplier: mad + and = 12.7, + 2 and = 16.7.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

enum V { MAD_VCC = 0, MAD_SROT, MAD_AND, MAD_AND2, MAD_CHAIN1, MAD_SFIX, MIX10, NUMV };
static const char* kNames[] = {"mad64 sdst=vcc, 16 chains (asm per instruction: s_nop artefact)",
                               "mad64 sdst rotating over 4 sgpr pairs, 16 chains", "mad64 + v_and (2 instr)",
                               "mad64 + 2 v_and (3 instr)", "mad64 dependent chain x1", "mad64 sdst=s[20:21] fixed, 16 chains (s_nop artefact)",
                               "7 mad64 + v_and + v_lshrrev_b64 + v_lshl_add_u64 in one asm block (10 instr)"};
static const int kInstr[] = {1, 1, 2, 3, 1, 1, 10};

template <int VAR, int THREADS>
__global__ void __launch_bounds__(THREADS) k(uint32_t* out, int iters, uint32_t sa, uint32_t sb) {
    extern __shared__ uint32_t lds[];
    constexpr int NACC = 16;
    uint64_t acc[NACC], aux[NACC];
    uint32_t x[NACC];
    uint32_t a = sa * (threadIdx.x * 2 + 1) + 12345u, b = sb + threadIdx.x * 7u;
#pragma unroll
    for (int k_ = 0; k_ < NACC; ++k_) { acc[k_] = ((uint64_t)(a + k_) << 20) | (b ^ k_); aux[k_] = acc[k_] * 3; x[k_] = a ^ k_; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k_ = 0; k_ < NACC; ++k_) {
                if constexpr (VAR == MAD_VCC) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (VAR == MAD_SROT) {
                    if ((k_ & 3) == 0) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "s20", "s21");
                    if ((k_ & 3) == 1) asm volatile("v_mad_u64_u32 %0, s[22:23], %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "s22", "s23");
                    if ((k_ & 3) == 2) asm volatile("v_mad_u64_u32 %0, s[24:25], %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "s24", "s25");
                    if ((k_ & 3) == 3) asm volatile("v_mad_u64_u32 %0, s[26:27], %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "s26", "s27");
                } else if constexpr (VAR == MAD_AND) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, %1, %2" : "+v"(acc[k_]), "+v"(x[k_]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (VAR == MAD_AND2) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, %1, %2\n\tv_and_b32 %1, %1, %3" : "+v"(acc[k_]), "+v"(x[k_]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (VAR == MAD_CHAIN1) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (VAR == MAD_SFIX) {
                    asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc[k_]) : "v"(a), "v"(b) : "s20", "s21");
                } else if constexpr (VAR == MIX10) {
                    asm volatile("v_mad_u64_u32 %0, s[20:21], %3, %4, %0\n\tv_mad_u64_u32 %1, s[20:21], %3, %4, %1\n\t"
                                 "v_mad_u64_u32 %0, s[20:21], %4, %3, %0\n\tv_mad_u64_u32 %1, s[20:21], %4, %3, %1\n\t"
                                 "v_mad_u64_u32 %0, s[20:21], %3, %3, %0\n\tv_mad_u64_u32 %1, s[20:21], %4, %4, %1\n\t"
                                 "v_mad_u64_u32 %0, s[20:21], %3, %4, %0\n\tv_and_b32 %2, %2, %3\n\t"
                                 "v_lshrrev_b64 %1, 29, %1\n\tv_lshl_add_u64 %0, %0, 0, %1"
                                 : "+v"(acc[k_]), "+v"(aux[k_]), "+v"(x[k_]) : "v"(a), "v"(b) : "s20", "s21");
                }
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int k_ = 0; k_ < NACC; ++k_) s ^= acc[k_] + aux[k_] + x[k_];
    if (s == 0x123456789abcdefull) { out[0] = (uint32_t)s; lds[threadIdx.x] = 1; }
}

template <int VAR, int THREADS>
static void run(int blocks, int lds_bytes, int iters, uint32_t* d_out, int ncu) {
    auto kern = k<VAR, THREADS>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), lds_bytes, 0, d_out, iters, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), lds_bytes, 0, d_out, iters, 3u, 5u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // cycles per group per SIMD: a SIMD hosts THREADS/256 waves of each resident block
    const double groups_per_wave = (double)iters * 64;
    const double rounds = (double)((blocks + ncu - 1) / ncu);
    const double waves_per_simd = THREADS / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (groups_per_wave * rounds * waves_per_simd);
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %.0f, \"blocks\": %d, \"lds_kb\": %d, \"ms\": %.3f, \"instr_per_group\": %d, "
           "\"cycles_per_group_per_simd_at_2.4GHz\": %.3f}\n", kNames[VAR], waves_per_simd, blocks, lds_bytes / 1024, ms, kInstr[VAR], cyc);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    uint32_t* d_out; CK(hipMalloc(&d_out, 4096));
    const int BIG = 100 * 1024;
    run<MAD_VCC, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_VCC, 256>(2 * ncu, BIG, iters, d_out, ncu);      // one block per CU at a time: must take twice as long
    run<MAD_SFIX, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_SROT, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_SROT, 512>(ncu, BIG, iters, d_out, ncu);
    run<MAD_SROT, 1024>(ncu, BIG, iters, d_out, ncu);
    run<MAD_CHAIN1, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_AND, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_AND2, 256>(ncu, BIG, iters, d_out, ncu);
    run<MAD_AND, 512>(ncu, BIG, iters, d_out, ncu);
    run<MAD_AND2, 512>(ncu, BIG, iters, d_out, ncu);
    run<MIX10, 256>(ncu, BIG, iters, d_out, ncu);
    run<MIX10, 512>(ncu, BIG, iters, d_out, ncu);
    run<MIX10, 1024>(ncu, BIG, iters, d_out, ncu);
    CK(hipFree(d_out));
    return 0;
}
