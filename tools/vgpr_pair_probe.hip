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
typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
// BUF: the initial values come through raw buffer loads (16 bytes per lane and load) from a table the host filled — the form of access
// kernels_ctadd_div.hpp uses for its scratch; else they are computed
template <int WPS, bool BUF>
__global__ void __launch_bounds__(256, WPS) k(uint32_t* out, int iters, int use_lds, const uint32_t* tab, int tab_bytes) {
    extern __shared__ uint32_t lds[];
    uint32_t r[NR];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if constexpr (BUF) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tab), 0, tab_bytes, 0x00020000);
        const uint32_t nslots = gridDim.x * 256;
#pragma unroll
        for (int c = 0; c < NR / 4; ++c) {
            const v4u_ t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(threadIdx.x * 16u), (int)((c * nslots + blockIdx.x * 256u) * 16u), 0);
            r[4 * c] = t.x; r[4 * c + 1] = t.y; r[4 * c + 2] = t.z; r[4 * c + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = gid * 977u + i * 131u + 7u;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = mixone(r[i], r[(i + 1) % NR]);
        if (use_lds) { lds[threadIdx.x] = r[it % 4]; __syncthreads(); r[0] ^= lds[threadIdx.x]; __syncthreads(); }
    }
    if constexpr (BUF) {                                  // ... and a store / reload of every register through the buffer (read after write)
        uint32_t* tw = const_cast<uint32_t*>(tab);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tw, 0, tab_bytes, 0x00020000);
        const uint32_t nslots = gridDim.x * 256;
#pragma unroll
        for (int c = 0; c < NR / 4; ++c) {
            v4u_ t; t.x = r[4 * c]; t.y = r[4 * c + 1]; t.z = r[4 * c + 2]; t.w = r[4 * c + 3];
            __builtin_amdgcn_raw_buffer_store_b128(t, rsrc, (int)(threadIdx.x * 16u), (int)((c * nslots + blockIdx.x * 256u) * 16u), 0);
        }
#pragma unroll
        for (int c = 0; c < NR / 4; ++c) {
            const v4u_ t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(threadIdx.x * 16u), (int)((c * nslots + blockIdx.x * 256u) * 16u), 0);
            r[4 * c] = t.x; r[4 * c + 1] = t.y; r[4 * c + 2] = t.z; r[4 * c + 3] = t.w;
        }
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
    for (int variant = 0; variant < 4; ++variant) {
        const int wps = 1 + (variant & 1);
        const bool buf = variant >= 2;
        const int grid = ncu * wps, n = grid * 256, bytes = 75776;
        uint32_t* d; CK(hipMalloc(&d, n * 4));
        const int tab_bytes = (NR / 4) * n * 16;
        std::vector<uint32_t> tab((size_t)tab_bytes / 4);
        for (int c = 0; c < NR / 4; ++c) for (int g = 0; g < n; ++g) for (int k4 = 0; k4 < 4; ++k4) tab[((size_t)c * n + g) * 4 + k4] = (uint32_t)g * 977u + (4 * c + k4) * 131u + 7u;
        uint32_t* dt; CK(hipMalloc(&dt, tab_bytes)); CK(hipMemcpy(dt, tab.data(), tab_bytes, hipMemcpyHostToDevice));
        auto go = [&](auto kern) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), bytes, 0, d, iters, 1, dt, tab_bytes); };
        if (wps == 1 && !buf) go(k<1, false>); else if (wps == 2 && !buf) go(k<2, false>); else if (wps == 1) go(k<1, true>); else go(k<2, true>);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
        int bad = 0, first = -1; unsigned lanes = 0;
        for (int g = 0; g < n; ++g) if (h[g] != host(g, iters, 1)) { if (first < 0) first = g; ++bad; lanes |= 1u << (g % 16); }
        printf("{\"buffer_ops\": %d, \"waves_per_simd\": %d, \"grid\": %d, \"bad\": %d, \"first_bad\": %d, \"lane_mod16_mask\": \"0x%x\"}\n", (int)buf, wps, grid, bad, first, lanes);
        CK(hipFree(d)); CK(hipFree(dt));
    }
    return 0;
}
