// Dev probe: do two waves per SIMD with 256 VGPRs each (two 74 KB workgroups per CU) compute correctly?  Every lane mixes NR registers
// for `iters` rounds; the host repeats it.  Prints the lanes / blocks that differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int NR = 236;
__host__ __device__ inline uint32_t mixone(uint32_t a, uint32_t b) { return (a ^ (b * 2654435761u)) * 40503u + (b >> 7); }
template <int WPS>
__global__ void __launch_bounds__(256, WPS) k(uint32_t* out, int iters, int use_lds) {
    extern __shared__ uint32_t lds[];
    uint32_t r[NR];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = gid * 977u + i * 131u + 7u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = mixone(r[i], r[(i + 1) % NR]);
        if (use_lds) { lds[threadIdx.x] = r[it % 4]; __syncthreads(); r[0] ^= lds[threadIdx.x]; __syncthreads(); }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) s = s * 31u + r[i];
    out[gid] = s;
}
static uint32_t host(uint32_t gid, int iters, int use_lds) {
    uint32_t r[NR];
    for (int i = 0; i < NR; ++i) r[i] = gid * 977u + i * 131u + 7u;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < NR; ++i) r[i] = mixone(r[i], r[(i + 1) % NR]);
        if (use_lds) r[0] ^= r[it % 4];
    }
    uint32_t s = 0;
    for (int i = 0; i < NR; ++i) s = s * 31u + r[i];
    return s;
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount, iters = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = ncu * wps, n = grid * 256, bytes = 75776;
        uint32_t* d; CK(hipMalloc(&d, n * 4));
        if (wps == 1) { CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), bytes, 0, d, iters, 1); }
        else { CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), bytes, 0, d, iters, 1); }
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
        int bad = 0, first = -1; unsigned lanes = 0;
        for (int g = 0; g < n; ++g) if (h[g] != host(g, iters, 1)) { if (first < 0) first = g; ++bad; lanes |= 1u << (g % 16); }
        printf("{\"waves_per_simd\": %d, \"grid\": %d, \"bad\": %d, \"first_bad\": %d, \"lane_mod16_mask\": \"0x%x\"}\n", wps, grid, bad, first, lanes);
        CK(hipFree(d));
    }
    return 0;
}
