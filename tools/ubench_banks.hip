// Does the register choice of v_mad_u64_u32's operands change its rate?  acc[k] += (u64)a[(k + S) & 15] * b[(k + T) & 15] over 16
// accumulators for several (S, T): the compiler allocates a[], b[], acc[] in consecutive registers, so S and T move the
// operands' VGPR banks (index mod 4) against the accumulator's.  Also: the multiplier from an SGPR (as the modulus limbs of
// the digit kernels).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_banks.hip -o tools/ubench_banks
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int NACC = 16;
template <int S, int T, bool SGPRB>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t sa, uint32_t sb) {
    uint64_t acc[NACC];
    uint32_t a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = sa * (threadIdx.x * 2 + 1) + 12345u * i; b[i] = SGPRB ? (sb + 977u * i) : (sb + threadIdx.x * 7u + 977u * i); acc[i] = ((uint64_t)a[i] << 20) | b[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int kk = 0; kk < NACC; ++kk) acc[kk] += (uint64_t)a[(kk + S + r) & 15] * b[(kk + T + 3 * r) & 15];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(a[i]));
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s ^= acc[i];
    if (s == 0x123456789abcdefull) out[0] = (uint32_t)s;
}
template <int S, int T, bool SGPRB>
static void run(uint32_t* d_out, int ncu, int wps) {
    const int blocks = ncu * wps, iters = 20000 / wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<S, T, SGPRB>), dim3(blocks), dim3(256), 0, 0, d_out, 100, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<S, T, SGPRB>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u, 5u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = (double)blocks * 256.0 * iters * NACC * 8;
    printf("{\"S\": %d, \"T\": %d, \"b_in_sgpr\": %s, \"waves_per_simd\": %d, \"T_mac_s\": %.2f, \"cycles_per_mad_at_2.4GHz\": %.2f}\n", S, T, SGPRB ? "true" : "false", wps,
           macs / (ms * 1e-3) / 1e12, (double)ncu * 4 * 64 * 2.4e9 / (macs / (ms * 1e-3)));
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    uint32_t* d; CK(hipMalloc(&d, 64));
    for (int wps : {1, 2, 4}) {
        run<0, 0, false>(d, p.multiProcessorCount, wps);
        run<1, 0, false>(d, p.multiProcessorCount, wps);
        run<2, 1, false>(d, p.multiProcessorCount, wps);
        run<3, 2, false>(d, p.multiProcessorCount, wps);
        run<5, 7, false>(d, p.multiProcessorCount, wps);
        run<0, 0, true>(d, p.multiProcessorCount, wps);
        run<3, 2, true>(d, p.multiProcessorCount, wps);
    }
    return 0;
}
