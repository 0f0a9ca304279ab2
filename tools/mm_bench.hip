// Montgomery-multiplication micro-benchmark + self-check for the device row engine (mont_dev.hpp).
// For each geometry (NLL limbs/lane, T lanes/element, U rows/block, modulus in VGPRs or LDS) it runs
//   y = x^(2^ITERS) mod M   as  to_mont -> ITERS Montgomery squarings -> from_mont -> canonicalise
// on N random elements, checks every result against the host big-integer reference (hostbn.hpp)
// and reports time per Montgomery multiplication and the achieved MAC32 rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mm_bench.hip -o tools/mm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../pailliercryptolib_python_amd/csrc/hostbn.hpp"
#include "../pailliercryptolib_python_amd/csrc/kernels_common.hpp"
#include "../pailliercryptolib_python_amd/csrc/kernels_wide.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using namespace pai;

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_sqr_chain(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
            int n, int w32, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t x[G::NLL];
        load_elem<G>(x, in + (size_t)es * w32, w32);
        {
            uint32_t r2[G::NLL];
            load_const_slice<G>(r2, ctx->r2);
            mm_times<G>(x, r2, lds, nm, n0inv);
        }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) mm_square<G>(x, lds, nm, n0inv);
        {
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
        }
        cond_sub<G::NLL, G::T>(x, nm);
        if (live) store_elem<G>(x, out + (size_t)ei * w32, w32, lds);
    }
}

static MontCtx make_ctx(const hbn::Limbs& M, int nl);

// wide engine: y = x^(2^iters) * w mod M   (to_mont by uniform R^2 digits, squarings, multiply by per-lane plain digits)
template <int NL>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_wide_chain(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ in, const uint32_t* __restrict__ in2,
             uint32_t* __restrict__ out, int n, int w32, int iters, int use_sqr) {
    using W = Wide<NL>;
    using IO = WideIO<NL>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t* __restrict__ nm = ctx->n;
    const uint32_t n0inv = ctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* xa = reinterpret_cast<uint4*>(lds + wave * W::WAVE_WORDS) + lane;
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        IO::load_low(xa, in + (size_t)es * w32, w32);
        {
            const uint32_t* __restrict__ r2 = ctx->r2;
            W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) bv[u] = r2[8 * blk + u];
            }, nm, n0inv);
        }
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if (use_sqr == 2) {          // the modexp pattern: 5 squarings + 1 multiplication (same code mix as k_dec_a_wide)
                if (it % 6 != 5) W::sqr(xa, nm, n0inv);
                else {
                    W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
                        const uint4 c0 = W::ld_chunk(xa, 2 * blk), c1 = W::ld_chunk(xa, 2 * blk + 1);
                        bv[0] = c0.x; bv[1] = c0.y; bv[2] = c0.z; bv[3] = c0.w;
                        bv[4] = c1.x; bv[5] = c1.y; bv[6] = c1.z; bv[7] = c1.w;
                    }, nm, n0inv);
                }
            } else if (use_sqr) W::sqr(xa, nm, n0inv);
            else {
                W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
                    const uint4 c0 = W::ld_chunk(xa, 2 * blk), c1 = W::ld_chunk(xa, 2 * blk + 1);
                    bv[0] = c0.x; bv[1] = c0.y; bv[2] = c0.z; bv[3] = c0.w;
                    bv[4] = c1.x; bv[5] = c1.y; bv[6] = c1.z; bv[7] = c1.w;
                }, nm, n0inv);
            }
        }
        {
            const uint32_t* row2 = in2 + (size_t)es * w32;
            W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) bv[u] = row_limb(row2, w32, 8 * blk + u);
            }, nm, n0inv);
        }
        W::cond_sub(xa, nm);
        if (live) IO::store_row(xa, out + (size_t)ei * w32, w32);
        wave_lds_fence();
    }
}

template <int NL>
static int run_wide(int bits, int n, int iters, int ncu, int use_sqr) {
    const int w32 = bits / 32;
    std::mt19937_64 rng(4321 + bits);
    hbn::Limbs M(w32);
    for (auto& w : M) w = (uint32_t)rng();
    M[0] |= 1u; M[w32 - 1] |= 0x80000000u;
    MontCtx hc = make_ctx(M, NL);
    std::vector<uint32_t> in((size_t)n * w32), in2((size_t)n * w32), out((size_t)n * w32);
    for (auto& w : in) w = (uint32_t)rng();
    for (auto& w : in2) w = (uint32_t)rng();
    for (int i = 0; i < n; ++i) { in[(size_t)i * w32 + w32 - 1] &= 0x7fffffffu; in2[(size_t)i * w32 + w32 - 1] &= 0x7fffffffu; }
    MontCtx* dctx; uint32_t *din, *din2, *dout;
    CK(hipMalloc(&dctx, sizeof(MontCtx))); CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&din2, in.size() * 4)); CK(hipMalloc(&dout, out.size() * 4));
    CK(hipMemcpy(dctx, &hc, sizeof(MontCtx), hipMemcpyHostToDevice));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(din2, in2.data(), in2.size() * 4, hipMemcpyHostToDevice));
    const int lds_bytes = NL * 256 * 4;
    CK(hipFuncSetAttribute((const void*)k_wide_chain<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    int grid = std::min(tiles, ncu * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_wide_chain<NL>, dim3(grid), dim3(BLOCK_THREADS), lds_bytes, 0, dctx, din, din2, dout, n, w32, 2, use_sqr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_wide_chain<NL>, dim3(grid), dim3(BLOCK_THREADS), lds_bytes, 0, dctx, din, din2, dout, n, w32, iters, use_sqr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    hbn::Mont32 mt(M);
    for (int c = 0; c < std::min(n, 48); ++c) {
        int i = (c < 24) ? c : n - 1 - (c - 24);
        hbn::Limbs x = hbn::from_u32(&in[(size_t)i * w32], w32), w = hbn::from_u32(&in2[(size_t)i * w32], w32);
        hbn::Limbs y = mt.to_mont(hbn::mod(x, M));
        for (int it = 0; it < iters; ++it) y = mt.mmul(y, y);
        y = hbn::mulmod(mt.from_mont(y), hbn::mod(w, M), M);
        y.resize(w32, 0);
        if (memcmp(y.data(), &out[(size_t)i * w32], w32 * 4) != 0) ++bad;
    }
    double mms = (double)n * (iters + 2);
    double L32 = bits / 32.0, mac_canon = mms * (2.0 * L32 * L32 + L32);
    printf("{\"geo\": \"wide%d %s\", \"bits\": %d, \"n\": %d, \"iters\": %d, \"grid\": %d, \"ms\": %.3f, "
           "\"ns_per_mm_per_elem\": %.3f, \"canon_TMAC32_s\": %.3f, \"mismatches\": %d}\n",
           NL, use_sqr == 2 ? "mix" : use_sqr ? "sqr" : "mul", bits, n, iters, grid, ms, ms * 1e6 / mms, mac_canon / (ms * 1e-3) / 1e12, bad);
    fflush(stdout);
    CK(hipFree(dctx)); CK(hipFree(din)); CK(hipFree(din2)); CK(hipFree(dout));
    return bad;
}

static MontCtx make_ctx(const hbn::Limbs& M, int nl) {
    MontCtx c;
    memset(&c, 0, sizeof(c));
    auto put = [&](uint32_t* dst, const hbn::Limbs& v) { auto r = hbn::to_r29(v, nl); memcpy(dst, r.data(), nl * 4); };
    put(c.n, M);
    hbn::Limbs R = hbn::mod(hbn::shl(hbn::Limbs{1u}, 29 * nl), M);
    put(c.one, R);
    put(c.r2, hbn::mulmod(R, R, M));
    uint32_t inv32 = hbn::neg_inv32(M[0]);
    c.n0inv = inv32 & ((1u << 29) - 1);
    c.nl = nl;
    c.bits = hbn::bitlen(M);
    return c;
}

template <class G>
static int run(const char* name, int bits, int n, int iters, int ncu, bool check) {
    const int nl = G::NL, w32 = bits / 32;
    std::mt19937_64 rng(1234 + bits);
    hbn::Limbs M(w32);
    for (auto& w : M) w = (uint32_t)rng();
    M[0] |= 1u; M[w32 - 1] |= 0x80000000u;
    MontCtx hc = make_ctx(M, nl);
    std::vector<uint32_t> in((size_t)n * w32), out((size_t)n * w32);
    for (auto& w : in) w = (uint32_t)rng();
    for (int i = 0; i < n; ++i) in[(size_t)i * w32 + w32 - 1] &= 0x7fffffffu;   // < M
    MontCtx* dctx; uint32_t *din, *dout;
    CK(hipMalloc(&dctx, sizeof(MontCtx))); CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, out.size() * 4));
    CK(hipMemcpy(dctx, &hc, sizeof(MontCtx), hipMemcpyHostToDevice));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_sqr_chain<G>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    int tiles = (n + G::EPB - 1) / G::EPB;
    int grid = std::min(tiles, ncu * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_sqr_chain<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, 0, dctx, din, dout, n, w32, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_sqr_chain<G>, dim3(grid), dim3(BLOCK_THREADS), G::LDS_BYTES, 0, dctx, din, dout, n, w32, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    if (check) {
        hbn::Mont32 mt(M);
        int nchk = std::min(n, 64);
        for (int c = 0; c < nchk; ++c) {
            int i = (c < 32) ? c : n - 1 - (c - 32);        // first and last elements
            hbn::Limbs x = hbn::from_u32(&in[(size_t)i * w32], w32);
            hbn::Limbs y = mt.to_mont(hbn::mod(x, M));
            for (int it = 0; it < iters; ++it) y = mt.mmul(y, y);
            y = mt.from_mont(y);
            y.resize(w32, 0);
            if (memcmp(y.data(), &out[(size_t)i * w32], w32 * 4) != 0) ++bad;
        }
    }
    double mms = (double)n * (iters + 2);
    double mac_exec = mms * 2.0 * nl * nl;                  // radix-2^29 MACs actually issued per MM
    double L32 = bits / 32.0, mac_canon = mms * (2.0 * L32 * L32 + L32);   // SURVEY §8d canonical count
    printf("{\"geo\": \"%s\", \"bits\": %d, \"NLL\": %d, \"T\": %d, \"U\": %d, \"nm_lds\": %d, \"n\": %d, \"iters\": %d, \"grid\": %d, "
           "\"ms\": %.3f, \"ns_per_mm_per_elem\": %.3f, \"exec_TMAC_s\": %.3f, \"canon_TMAC32_s\": %.3f, \"mismatches\": %d}\n",
           name, bits, G::NLL, G::T, G::U, (int)G::NMLDS, n, iters, grid, ms, ms * 1e6 / mms, mac_exec / (ms * 1e-3) / 1e12,
           mac_canon / (ms * 1e-3) / 1e12, bad);
    fflush(stdout);
    CK(hipFree(dctx)); CK(hipFree(din)); CK(hipFree(dout));
    return bad;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 200;
    int nmul = argc > 2 ? atoi(argv[2]) : 4;     // elements = nmul * resident capacity
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    int bad = 0;
#define RUN(NLL, T, U, LDSF, BITS) bad += run<Geo<NLL, T, U, LDSF>>(#NLL "x" #T " U" #U " lds" #LDSF, BITS, ncu * 2 * (BLOCK_THREADS / T) * nmul - 3, iters, ncu, true)
#ifdef WIDE_ONLY
    bad += run_wide<72>(2048, ncu * 2 * BLOCK_THREADS * nmul - 3, iters, ncu, 1);
    bad += run_wide<72>(2048, ncu * 2 * BLOCK_THREADS * nmul - 3, iters, ncu, 0);
    bad += run_wide<72>(2048, ncu * 2 * BLOCK_THREADS * nmul - 3, iters, ncu, 2);
    bad += run_wide<40>(1024, ncu * 2 * BLOCK_THREADS * nmul - 3, iters, ncu, 1);
    bad += run_wide<40>(1024, ncu * 2 * BLOCK_THREADS * nmul - 3, iters, ncu, 0);
#elif defined(ONLY)
    RUN(ONLY_NLL, ONLY_T, ONLY_U, ONLY_LDS, ONLY_BITS);
#else
    RUN(36, 1, 6, false, 1024);
    RUN(36, 1, 12, false, 1024);
    RUN(36, 2, 6, false, 2048);
    RUN(36, 2, 6, true, 2048);
    RUN(36, 2, 12, true, 2048);
    RUN(36, 2, 4, true, 2048);
    RUN(18, 4, 6, false, 2048);
    RUN(36, 4, 6, true, 4096);
    RUN(36, 4, 6, false, 4096);
    RUN(36, 8, 6, true, 8192);
    RUN(28, 4, 4, true, 3072);
#endif
    printf("{\"total_mismatches\": %d}\n", bad);
    return bad ? 1 : 0;
}
