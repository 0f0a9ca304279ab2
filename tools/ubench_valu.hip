// Instruction-throughput microbenchmark for gfx950 (MI355X): fixes the integer-VALU roofline
// denominator used by bench.py / DESIGN.md.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Each kernel runs ITER iterations of an unrolled body of 64 independent-chain instructions on
// 16 accumulators per lane; we report wave-instructions/s and cycles per wave-instruction per SIMD
// (assuming 1024 SIMDs at the clock measured by a v_add_u32 calibration = 2 cycles/wave-instr... no:
// we print raw rates; the reader divides).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NACC = 16;
constexpr int REP = 4;   // body = NACC*REP instructions

enum Op { MAD64 = 0, MAD64_SGPR, MUL_LO, MUL_HI, MAD_U24, MULHI_U24, FMA64, FMA32, ADD_U32, ADDC_PAIR, LSHL_ADD_U64,
          ALIGNBIT, MAD64_ADDC, DOT2_U16, DOT4_U8, PKFMA32, LSHR64, AND_B32, MAD_U32, MAD64_C, NUM_OPS };
static const char* kNames[] = {"v_mad_u64_u32", "v_mad_u64_u32(sgpr src)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
  "v_mul_hi_u32_u24", "v_fma_f64", "v_fma_f32", "v_add_u32", "v_add_co+v_addc_co (pair=2 instr)", "v_lshl_add_u64",
  "v_alignbit_b32", "v_mad_u64_u32+v_addc_co (pair=2 instr)", "v_dot2_u32_u16", "v_dot4_u32_u8", "v_pk_fma_f32",
  "v_lshrrev_b64", "v_and_b32", "v_mad_u32_u24(dup)", "u64 += u32*u32 (compiler)"};
static const int kInstrPerSlot[] = {1,1,1,1,1,1,1,1,1,2,1,1,2,1,1,1,1,1,1,1};

template <int OP>
__global__ void __launch_bounds__(256) k_bench(uint32_t* out, int iters, uint32_t seed_a, uint32_t seed_b) {
    uint64_t acc[NACC];
    uint32_t cnt[NACC];
    uint32_t a = seed_a * (threadIdx.x * 2 + 1) + 12345u, b = seed_b + threadIdx.x * 7u;
    uint32_t sb = seed_b | 1u;  // uniform
#pragma unroll
    for (int k = 0; k < NACC; ++k) { acc[k] = ((uint64_t)(a + k) << 20) | (b ^ k); cnt[k] = k; }
    double da = 1.0 + 1e-9 * a, db = 1.0 - 1e-9 * b;
    float fa = 1.0f + 1e-5f * (a & 255), fb = 1.0f - 1e-6f * (b & 255);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if constexpr (OP == MAD64_C) {
                    acc[k] += (uint64_t)(a + r) * (uint32_t)(b + k);
                } else if constexpr (OP == MAD64) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (OP == MAD64_SGPR) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "s"(sb), "v"(b) : "vcc");
                } else if constexpr (OP == MUL_LO) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == MUL_HI) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == MAD_U24 || OP == MAD_U32) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
                    acc[k] = lo;
                } else if constexpr (OP == MULHI_U24) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == FMA64) {
                    double d = __longlong_as_double((long long)acc[k]);
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(da), "v"(db));
                    acc[k] = (uint64_t)__double_as_longlong(d);
                } else if constexpr (OP == FMA32) {
                    float f = __uint_as_float((uint32_t)acc[k]);
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(fa), "v"(fb));
                    acc[k] = __float_as_uint(f);
                } else if constexpr (OP == PKFMA32) {
                    uint64_t pa = ((uint64_t)__float_as_uint(fa) << 32) | __float_as_uint(fb);
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc[k]) : "v"(pa));
                } else if constexpr (OP == ADD_U32) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == AND_B32) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == ADDC_PAIR) {
                    uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                                 : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                    acc[k] = ((uint64_t)hi << 32) | lo;
                } else if constexpr (OP == LSHL_ADD_U64) {
                    uint64_t ab = ((uint64_t)a << 32) | b;
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(ab));
                } else if constexpr (OP == LSHR64) {
                    asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[k]));
                } else if constexpr (OP == ALIGNBIT) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(lo) : "v"(a));
                    acc[k] = lo;
                } else if constexpr (OP == MAD64_ADDC) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
                                 : "+v"(acc[k]), "+v"(cnt[k]) : "v"(a), "v"(b) : "vcc");
                } else if constexpr (OP == DOT2_U16) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
                    acc[k] = lo;
                } else if constexpr (OP == DOT4_U8) {
                    uint32_t lo = (uint32_t)acc[k];
                    asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
                    acc[k] = lo;
                }
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s ^= acc[k] + cnt[k];
    if (s == 0x123456789abcdefull) out[0] = (uint32_t)s;   // never true in practice; keeps the chains live
}

template <int OP>
static void run(int waves_per_simd, int iters, uint32_t* d_out, int ncu) {
    int blocks = ncu * waves_per_simd;   // 256 threads = 4 waves => one wave per SIMD per block
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8, 3u, 5u);  // warmup
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_bench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u, 5u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double slots = (double)blocks * 4.0 * iters * NACC * REP;          // wave-slots (a slot may be 2 instr)
    double per_simd_slot_cycles = (ms * 1e-3 * 2.4e9) / (slots / (ncu * 4.0));
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_slots_per_s\": %.4e, \"lane_ops_per_s\": %.4e, "
           "\"cycles_per_slot_per_simd_at_2.4GHz\": %.3f, \"instr_per_slot\": %d}\n",
           kNames[OP], waves_per_simd, ms, slots / (ms * 1e-3), slots * 64.0 / (ms * 1e-3), per_simd_slot_cycles,
           kInstrPerSlot[OP]);
    fflush(stdout);
}

template <int OP> static void sweep(uint32_t* d_out, int ncu, int iters) {
    for (int w : {1, 2, 4, 8}) run<OP>(w, iters, d_out, ncu);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"gcn\": \"%s\"}\n", p.name, ncu, p.clockRate, p.gcnArchName);
    uint32_t* d_out; CK(hipMalloc(&d_out, 4096));
    sweep<ADD_U32>(d_out, ncu, iters);
    sweep<FMA32>(d_out, ncu, iters);
    sweep<MAD64>(d_out, ncu, iters);
    sweep<MAD64_C>(d_out, ncu, iters);
    sweep<MAD64_SGPR>(d_out, ncu, iters);
    sweep<MAD64_ADDC>(d_out, ncu, iters);
    sweep<MUL_LO>(d_out, ncu, iters);
    sweep<MUL_HI>(d_out, ncu, iters);
    sweep<MAD_U24>(d_out, ncu, iters);
    sweep<MULHI_U24>(d_out, ncu, iters);
    sweep<FMA64>(d_out, ncu, iters);
    sweep<PKFMA32>(d_out, ncu, iters);
    sweep<ADDC_PAIR>(d_out, ncu, iters);
    sweep<LSHL_ADD_U64>(d_out, ncu, iters);
    sweep<LSHR64>(d_out, ncu, iters);
    sweep<ALIGNBIT>(d_out, ncu, iters);
    sweep<AND_B32>(d_out, ncu, iters);
    sweep<DOT2_U16>(d_out, ncu, iters);
    sweep<DOT4_U8>(d_out, ncu, iters);
    CK(hipFree(d_out));
    return 0;
}
