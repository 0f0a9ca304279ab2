// Timing probe for the digit-pair engine (mont_padic.hpp): chains of product-rule multiplications x <- x * x
// on random digit pairs, for the decrypt geometry <36,12> and the (experimental) encrypt geometry <72,8>.
// No host check here (correctness is covered by the Paillier parity tests); prints ns per product per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../pailliercryptolib_python_amd/csrc/kernels_common.hpp"
#include "../pailliercryptolib_python_amd/csrc/mont_padic.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace pai;

// MODE 0: mul (M in LDS), 1: mul_wbuf (M, W global), 2: sqr (M in LDS), 3: sqr_fused, 4: mul_fused (both halves in one pass)
template <int NL, int U, int MODE>
__global__ void __launch_bounds__(BLOCK_THREADS, MODE >= 3 ? 2 : 1)
k_chain(const uint32_t* __restrict__ mod, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint4* scratch, int iters) {
    using E = Padic<NL, U>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int ND = (MODE == 1 || MODE >= 3) ? 2 : 3;
    uint32_t* ldsn = lds + 4 * ND * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = mod[i]; ldsn[NL + i] = mod[i] - (i == 0 ? 1u : 0u); }
    __syncthreads();
#if defined(NM_SGPR) || defined(LEAN_SGPR)
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
#else
    const uint32_t* nm = ldsn;
#endif
    const uint32_t* pm1 = ldsn + NL;
    const uint32_t n0inv = mod[NLMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * ND * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS, slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf Mg{scratch + slot, nslots}, Wg{scratch + (size_t)E::NC * nslots + slot, nslots};
    const typename E::MBuf Ml{B + E::NC * 64, 64};
    for (int c = 0; c < E::NC; ++c) {
        const uint32_t* p = in + ((size_t)slot * 2 * NL) + 4 * c;
        E::st(A, c, make_uint4(p[0] & RMASK, p[1] & RMASK, p[2] & RMASK, p[3] & RMASK));
        E::st(B, c, make_uint4(p[NL] & RMASK, p[NL + 1] & RMASK, p[NL + 2] & RMASK, p[NL + 3] & RMASK));
    }
    wave_lds_fence();
    auto self = [&](const uint4* X) { return [=](int blk, uint32_t (&xv)[U]) { E::digits(X, blk, xv); }; };
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) E::mul(A, B, Ml, self(A), self(B), nm, pm1, n0inv);
        else if constexpr (MODE == 1) E::mul_wbuf(A, B, Mg, Wg, self(A), self(B), nm, pm1, n0inv);
        else if constexpr (MODE == 2) E::sqr(A, B, Ml, nm, pm1, n0inv);
        else if constexpr (MODE == 3) E::sqr_fused(A, B, nm, pm1, n0inv);
        else if constexpr (MODE == 4) E::mul_fused(A, B, self(A), self(B), nm, pm1, n0inv);
        else E::mul_wbuf(A, B, Mg, Wg, self(A), self(B), nm, pm1, n0inv);
    }
    for (int c = 0; c < E::NC; ++c) {
        const uint4 a = E::ld(A, c), b = E::ld(B, c);
        uint32_t* p = out + ((size_t)slot * 2 * NL) + 4 * c;
        p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; p[NL] = b.x; p[NL + 1] = b.y; p[NL + 2] = b.z; p[NL + 3] = b.w;
    }
}

template <int NL, int U, int MODE>
static void run(const char* name, int iters, int ncu) {
    using E = Padic<NL, U>;
    const int grid = ncu * (MODE >= 3 ? 2 : 1), n = grid * BLOCK_THREADS;
    std::mt19937 rng(7);
    std::vector<uint32_t> mod(NLMAX + 1, 0), in((size_t)n * 2 * NL);
    for (int i = 0; i < NL - 1; ++i) mod[i] = rng() & RMASK;
    mod[0] |= 1; mod[NL - 1] = 0;                                     // modulus well below R
    uint32_t inv = mod[0]; for (int i = 0; i < 5; ++i) inv *= 2u - mod[0] * inv;
    mod[NLMAX] = (0u - inv) & RMASK;
    for (auto& w : in) w = rng();
    uint32_t *dmod, *din, *dout; uint4* dscr;
    CK(hipMalloc(&dmod, mod.size() * 4)); CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, in.size() * 4));
    CK(hipMalloc(&dscr, (size_t)2 * E::NC * n * 16));
    CK(hipMemcpy(dmod, mod.data(), mod.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    constexpr int ND = (MODE == 1 || MODE >= 3) ? 2 : 3;
    const int bytes = 4 * ND * E::DIGIT_WORDS * 4 + 2 * NL * 4;
    CK(hipFuncSetAttribute((const void*)k_chain<NL, U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chain<NL, U, MODE>), dim3(grid), dim3(BLOCK_THREADS), bytes, 0, dmod, din, dout, dscr, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<NL, U, MODE>), dim3(grid), dim3(BLOCK_THREADS), bytes, 0, dmod, din, dout, dscr, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = (MODE == 2 || MODE == 3 || MODE == 5) ? (0.5 * NL * (NL + 1) + 3.0 * NL * NL) : 5.0 * NL * NL;
    printf("{\"probe\": \"%s\", \"NL\": %d, \"U\": %d, \"elements\": %d, \"iters\": %d, \"ms\": %.3f, \"ns_per_product_per_elem\": %.4f, "
           "\"exec_TMAC_s\": %.2f}\n", name, NL, U, n, iters, ms, ms * 1e6 / ((double)n * iters), macs * n * iters / (ms * 1e-3) / 1e12);
    fflush(stdout);
    CK(hipFree(dmod)); CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dscr));
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 100;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    run<36, 12, 0>("mul <36,12> M in LDS", iters, ncu);
    run<36, 12, 2>("sqr <36,12> M in LDS", iters, ncu);
    run<72, 8, 1>("mul_wbuf <72,8> M,W global", iters, ncu);
    run<72, 8, 3>("sqr_fused <72,8>", iters, ncu);
    run<72, 8, 4>("mul_fused <72,8>", iters, ncu);
    return 0;
}
