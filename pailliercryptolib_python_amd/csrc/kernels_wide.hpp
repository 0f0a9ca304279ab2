// Kernels on the wide (one integer per lane) Montgomery engine — see mont_wide.hpp.
//   k_dec_a_wide   CRT-decrypt stage A: u_s = (ct mod s^2)^(s-1) mod s^2 for s in {p, q} (blockIdx.y)
//                  same contract as k_dec_a (kernels_paillier.hpp), used when s^2 fits 72 (or 40) limbs.
#pragma once
#include "kernels_paillier.hpp"
#include "mont_wide.hpp"

namespace pai {

// limb J (29 bits) of a packed little-endian row of W32 words; bits beyond the row read as zero
PAI_DEV uint32_t row_limb(const uint32_t* __restrict__ row, int W32, int J) {
    const int bit = RB * J, k = bit >> 5, s = bit & 31;
    const int k0 = k < W32 ? k : W32 - 1, k1 = k + 1 < W32 ? k + 1 : W32 - 1;
    uint32_t lo = row[k0], hi = row[k1];
    lo = k < W32 ? lo : 0u;
    hi = k + 1 < W32 ? hi : 0u;
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s) & RMASK;
}

template <int NL>
struct WideIO {
    using W = Wide<NL>;
    // limbs 0..NL-1 of a packed row -> LDS x
    PAI_DEV static void load_low(uint4* xa, const uint32_t* __restrict__ row, int W32) {
#pragma unroll 1
        for (int c = 0; c < W::NC; ++c) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = row_limb(row, W32, 4 * c + k);
            W::st_chunk(xa, c, make_uint4(w[0], w[1], w[2], w[3]));
        }
        wave_lds_fence();
    }
    // LDS x (canonical limbs) -> packed row of W32 words (value must fit)
    PAI_DEV static void store_row(const uint4* xa, uint32_t* __restrict__ row, int W32) {
        uint32_t x[NL];
#pragma unroll
        for (int c = 0; c < W::NC; ++c) {
            const uint4 v = W::ld_chunk(xa, c);
            x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
        }
        constexpr int MAXW = (RB * NL + 31) / 32;
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
            uint64_t v = (uint64_t)x[j0] >> s0;
            if (j0 + 1 < NL) v |= (uint64_t)x[j0 + 1] << (RB - s0);
            if (j0 + 2 < NL) v |= (uint64_t)x[j0 + 2] << (2 * RB - s0);
            if (k < W32) row[k] = (uint32_t)v;
        }
    }
};

// One wave per SIMD is enough for this engine (the inner loops are pure ILP: measured 0.375 vs 0.361 ns
// per multiplication at 1 vs 2 waves/SIMD), and the full 512-register budget lets the compiler park
// the kernel's long-lived scalars in AGPRs instead of spilling the accumulator window to scratch.
template <int NL, int WB>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_dec_a_wide(DecAParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out /*[2][n][u_words]*/, int n,
             uint4* __restrict__ table) {
    using W = Wide<NL>;
    using IO = WideIO<NL>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int which = blockIdx.y;
    const MontCtx* ctx = P.sq[which];
    const uint32_t* __restrict__ nm = ctx->n;
    const uint32_t n0inv = ctx->n0inv;
    const uint32_t* __restrict__ expo = P.expo[which];
    const int ewords = P.ewords[which], ebits = P.ebits[which];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* xa = reinterpret_cast<uint4*>(lds + wave * W::WAVE_WORDS) + lane;
    const size_t nslots = (size_t)gridDim.x * gridDim.y * BLOCK_THREADS;
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BLOCK_THREADS + threadIdx.x;
    auto tbl = [&](int entry, int chunk) -> uint4& { return table[((size_t)entry * W::NC + chunk) * nslots + slot]; };
    const int nwin = (ebits + WB - 1) / WB;
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* row = ct + (size_t)es * P.ct_words;
        // x = ct * R mod s^2 :  REDC(ct) = ct R^-1, then * R^3 * R^-1
        IO::load_low(xa, row, P.ct_words);
        W::redc(xa, [&](int blk, uint32_t (&hv)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[u] = row_limb(row, P.ct_words, NL + 8 * blk + u);
        }, nm, n0inv);
        {
            const uint32_t* __restrict__ r3 = P.r3[which];
            W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) bv[u] = r3[8 * blk + u];
            }, nm, n0inv);
        }
        // table[k] = base^k (Montgomery form), k = 1 .. 2^WB - 1
#pragma unroll 1
        for (int c = 0; c < W::NC; ++c) tbl(1, c) = W::ld_chunk(xa, c);
#pragma unroll 1
        for (int k = 2; k < (1 << WB); ++k) {
            W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
                const uint4 c0 = tbl(1, 2 * blk), c1 = tbl(1, 2 * blk + 1);
                bv[0] = c0.x; bv[1] = c0.y; bv[2] = c0.z; bv[3] = c0.w;
                bv[4] = c1.x; bv[5] = c1.y; bv[6] = c1.z; bv[7] = c1.w;
            }, nm, n0inv);
#pragma unroll 1
            for (int c = 0; c < W::NC; ++c) tbl(k, c) = W::ld_chunk(xa, c);
        }
        // top window
        {
            const uint32_t wv = exp_bits(expo, ewords, (nwin - 1) * WB, WB);
            const uint32_t* __restrict__ one = ctx->one;
#pragma unroll 1
            for (int c = 0; c < W::NC; ++c) {
                uint4 v;
                if (wv == 0) v = make_uint4(one[4 * c], one[4 * c + 1], one[4 * c + 2], one[4 * c + 3]);
                else v = tbl((int)wv, c);
                W::st_chunk(xa, c, v);
            }
            wave_lds_fence();
        }
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
            const uint32_t wv = exp_bits(expo, ewords, wi * WB, WB);
#pragma unroll 1
            for (int s = 0; s < WB; ++s) W::sqr(xa, nm, n0inv);
            if (wv != 0) {
                W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
                    const uint4 c0 = tbl((int)wv, 2 * blk), c1 = tbl((int)wv, 2 * blk + 1);
                    bv[0] = c0.x; bv[1] = c0.y; bv[2] = c0.z; bv[3] = c0.w;
                    bv[4] = c1.x; bv[5] = c1.y; bv[6] = c1.z; bv[7] = c1.w;
                }, nm, n0inv);
            }
        }
        // out of Montgomery form, canonical, store
        W::mul(xa, [&](int blk, uint32_t (&bv)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bv[u] = 0;
            if (blk == 0) bv[0] = 1;
        }, nm, n0inv);
        W::cond_sub(xa, nm);
        if (live) IO::store_row(xa, u_out + ((size_t)which * n + ei) * P.u_words, P.u_words);
        wave_lds_fence();
    }
}

}  // namespace pai
