// "Wide" Montgomery engine: one big integer per LANE (no lane groups), used where the modulus is
// small enough for a whole accumulator window to sit in one lane's 256 VGPRs (NL <= 72 limbs of 29
// bits: moduli up to 2086 bits, i.e. the CRT halves p^2, q^2 of keys up to 2048 bits — the dominant
// kernel of the headline benchmark).
//
// Differences from the lane-group engine (mont_dev.hpp):
//  * the running value x lives in LDS, not in registers, in a chunk-major layout: limbs 4c..4c+3 of
//    lane l form one uint4 at xa[c * 64 + l] (conflict-free ds_read_b128 / ds_write_b128).  Registers
//    hold only the 2*(NL+8)-register lazy accumulator window, eight multiplier digits, eight quotient
//    digits and one four-limb chunk of the multiplicand at a time;
//  * the modulus is wave-uniform and is read through scalar loads (SGPR operands of v_mad_u64_u32);
//  * rows are processed eight at a time: the eight quotient digits of a block are produced first
//    from the low eight columns (a short dependent chain), then the block is a pure rank-16 update
//    streamed over the remaining chunks — no cross-lane traffic anywhere;
//  * squaring uses the symmetry x_i x_j = x_j x_i at the granularity of limb classes (thirds of the
//    operand for NL = 72): limbs below the rows' class are skipped, limbs above it are multiplied by
//    the doubled digits.  The class structure is compile-time, so the code stays branch-free with
//    constant register indices (per-chunk uniform branches were tried: the accumulator window then
//    falls out of registers).  A squaring issues 1.67 NL^2 MACs instead of 2 NL^2.
#pragma once
#include "mont_dev.hpp"

namespace pai {

template <int NL>
struct Wide {
    static_assert(NL % 8 == 0, "NL must be a multiple of 8");
    static constexpr int U = 8;
    static constexpr int NC = NL / 4;        // four-limb chunks
    static constexpr int NB = NL / 8;        // eight-row blocks
    static constexpr int NW = NL + U;        // accumulator window
    static constexpr int WAVE_WORDS = NL * 64;

    // ---- LDS access: xa points at this lane's uint4 slot of chunk 0; chunk c is xa[c * 64] -------
    PAI_DEV static uint4 ld_chunk(const uint4* xa, int c) { return xa[c * 64]; }
    PAI_DEV static void st_chunk(uint4* xa, int c, uint4 v) { xa[c * 64] = v; }

    PAI_DEV static void normalize(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = NW - 1; j >= 1; --j) {
            uint64_t keep = (j == NW - 1) ? acc[j] : (acc[j] & RMASK);
            acc[j] = keep + (acc[j - 1] >> RB);
        }
        acc[0] &= RMASK;
    }

    // quotient digits of one block from the low eight columns; `ab8` = contribution of the low eight
    // limbs of the multiplicand is included when HAVE_A8
    template <bool HAVE_A8>
    PAI_DEV static void qchain(uint64_t (&acc)[NW], const uint32_t (&a8)[8], const uint32_t (&bv)[8], uint32_t (&q)[8],
                               const uint32_t* __restrict__ nm, uint32_t n0inv) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (HAVE_A8) {
#pragma unroll
                for (int j = 0; j <= u; ++j) acc[u] += (uint64_t)a8[j] * bv[u - j];
            }
#pragma unroll
            for (int j = 1; j <= u; ++j) acc[u] += (uint64_t)nm[j] * q[u - j];
            q[u] = ((uint32_t)acc[u] * n0inv) & RMASK;
            acc[u] += (uint64_t)nm[0] * q[u];
            acc[u + 1] += acc[u] >> RB;
        }
        // the rest of limbs 1..7 (columns 8..14)
#pragma unroll
        for (int j = 1; j < 8; ++j) {
#pragma unroll
            for (int u = 8 - j; u < 8; ++u) {
                if constexpr (HAVE_A8) acc[j + u] += (uint64_t)a8[j] * bv[u];
                acc[j + u] += (uint64_t)nm[j] * q[u];
            }
        }
    }

    PAI_DEV static void slide(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = acc[j + U];
#pragma unroll
        for (int j = NL; j < NW; ++j) acc[j] = 0;
    }

    // One block of a general multiplication: acc = (acc + x * B_blk + Q_blk * n) / 2^(29*8)
    PAI_DEV static void block_mul(uint64_t (&acc)[NW], const uint4* xa, const uint32_t (&bv)[8],
                                  const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint32_t q[8], a8[8];
        {
            const uint4 c0 = ld_chunk(xa, 0), c1 = ld_chunk(xa, 1);
            a8[0] = c0.x; a8[1] = c0.y; a8[2] = c0.z; a8[3] = c0.w;
            a8[4] = c1.x; a8[5] = c1.y; a8[6] = c1.z; a8[7] = c1.w;
        }
        qchain<true>(acc, a8, bv, q, nm, n0inv);
        // rank-16 update streamed over the remaining chunks, one chunk of x and of n prefetched ahead;
        // the scheduling fences keep the compiler from hoisting every chunk load to the top (which
        // would need 72+ extra registers and spill the accumulator window)
        __builtin_amdgcn_sched_barrier(0);
        uint4 a_cur = ld_chunk(xa, 2);
        uint32_t n_cur[4] = {nm[8], nm[9], nm[10], nm[11]};
#pragma unroll
        for (int c = 2; c < NC; ++c) {
            const int cn = (c + 1 < NC) ? c + 1 : c;
            const uint4 a_nxt = ld_chunk(xa, cn);
            const uint32_t n_nxt[4] = {nm[4 * cn], nm[4 * cn + 1], nm[4 * cn + 2], nm[4 * cn + 3]};
            const uint32_t av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc[4 * c + k + u] += (uint64_t)av[k] * bv[u];
                    acc[4 * c + k + u] += (uint64_t)n_cur[k] * q[u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
#pragma unroll
            for (int k = 0; k < 4; ++k) n_cur[k] = n_nxt[k];
        }
        slide(acc);
    }

    // One block of a squaring for rows inside the limb class [LO, HI) (multiples of 8).  Symmetry is
    // exploited at class granularity with fully static code: limbs below LO are skipped (their
    // products with these rows were issued, doubled, by the earlier rows), limbs of the own class are
    // multiplied normally, limbs at or above HI are multiplied by the doubled digits.  No branches and
    // only constant register indices, so the accumulator window stays in VGPRs and the window slide
    // is absorbed by register renaming exactly as in block_mul.
    template <int LO, int HI>
    PAI_DEV static void block_sqr(uint64_t (&acc)[NW], const uint4* xa, const uint32_t (&bv)[8],
                                  const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint32_t bv2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv2[u] = bv[u] << 1;
        uint32_t q[8], a8[8];
        if constexpr (LO == 0) {
            const uint4 c0 = ld_chunk(xa, 0), c1 = ld_chunk(xa, 1);
            a8[0] = c0.x; a8[1] = c0.y; a8[2] = c0.z; a8[3] = c0.w;
            a8[4] = c1.x; a8[5] = c1.y; a8[6] = c1.z; a8[7] = c1.w;
            if constexpr (HI > 8) qchain<true>(acc, a8, bv, q, nm, n0inv);
            else qchain<true>(acc, a8, bv, q, nm, n0inv);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = 0;
            qchain<false>(acc, a8, bv, q, nm, n0inv);
        }
        __builtin_amdgcn_sched_barrier(0);
        uint4 a_cur = ld_chunk(xa, 2);
        uint32_t n_cur[4] = {nm[8], nm[9], nm[10], nm[11]};
#pragma unroll
        for (int c = 2; c < NC; ++c) {
            const int cn = (c + 1 < NC) ? c + 1 : c;
            const uint4 a_nxt = ld_chunk(xa, cn);
            const uint32_t n_nxt[4] = {nm[4 * cn], nm[4 * cn + 1], nm[4 * cn + 2], nm[4 * cn + 3]};
            const uint32_t av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (4 * c >= HI) acc[4 * c + k + u] += (uint64_t)av[k] * bv2[u];
                    else if (4 * c >= LO) acc[4 * c + k + u] += (uint64_t)av[k] * bv[u];
                    acc[4 * c + k + u] += (uint64_t)n_cur[k] * q[u];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
#pragma unroll
            for (int k = 0; k < 4; ++k) n_cur[k] = n_nxt[k];
        }
        slide(acc);
    }

    template <int LO, int HI>
    PAI_DEV static void sqr_class(uint64_t (&acc)[NW], const uint4* xa, const uint32_t* __restrict__ nm, uint32_t n0inv) {
        static_assert(LO % 8 == 0 && HI % 8 == 0 && LO < HI && HI <= NL, "class bounds");
#pragma unroll 1
        for (int blk = LO / 8; blk < HI / 8; ++blk) {
            uint32_t bv[8];
            {
                const uint4 c0 = ld_chunk(xa, 2 * blk), c1 = ld_chunk(xa, 2 * blk + 1);
                bv[0] = c0.x; bv[1] = c0.y; bv[2] = c0.z; bv[3] = c0.w;
                bv[4] = c1.x; bv[5] = c1.y; bv[6] = c1.z; bv[7] = c1.w;
            }
            block_sqr<LO, HI>(acc, xa, bv, nm, n0inv);
            if ((blk & 1) == 1 && blk != NB - 1) normalize(acc);          // doubled products: every 16 rows
        }
    }

    // carry-propagate the window's low NL columns and write the canonical limbs into LDS
    PAI_DEV static void finish_to_lds(const uint64_t (&acc)[NW], uint4* xa) {
        uint64_t c = 0;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t t = acc[4 * ch + k] + c;
                w[k] = (uint32_t)t & RMASK;
                c = t >> RB;
            }
            st_chunk(xa, ch, make_uint4(w[0], w[1], w[2], w[3]));
        }
    }

    // x <- x * B * R^-1 mod M with the multiplier digits supplied per block by `bsrc(blk, bv)`
    template <class BSrc>
    PAI_DEV static void mul(uint4* xa, BSrc&& bsrc, const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint64_t acc[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[j] = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t bv[8];
            bsrc(blk, bv);
            block_mul(acc, xa, bv, nm, n0inv);
            if ((blk % 3) == 2 && blk != NB - 1) normalize(acc);          // every 24 rows
        }
        wave_lds_fence();
        finish_to_lds(acc, xa);
        wave_lds_fence();
    }

    // x <- x * x * R^-1 mod M.  Rows are grouped into limb classes (thirds for NL = 72) so that a
    // squaring issues  sum_k |class_k| * (NL - LO_k)  multiplicand MACs instead of NL^2
    // (NL = 72: 3456 instead of 5184; with the reduction half: 8640 instead of 10368 per squaring).
    PAI_DEV static void sqr(uint4* xa, const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint64_t acc[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[j] = 0;
        if constexpr (NL == 72) {
            sqr_class<0, 24>(acc, xa, nm, n0inv);
            sqr_class<24, 48>(acc, xa, nm, n0inv);
            sqr_class<48, 72>(acc, xa, nm, n0inv);
        } else if constexpr (NL == 40) {
            sqr_class<0, 16>(acc, xa, nm, n0inv);
            sqr_class<16, 40>(acc, xa, nm, n0inv);
        } else {
            sqr_class<0, NL>(acc, xa, nm, n0inv);
        }
        wave_lds_fence();
        finish_to_lds(acc, xa);
        wave_lds_fence();
    }

    // Montgomery reduction of a 2*NL-limb value whose low half is in LDS (xa) and whose high half is
    // supplied eight limbs per block by `hsrc(blk, hv)`:  x <- t * R^-1 mod M
    template <class HSrc>
    PAI_DEV static void redc(uint4* xa, HSrc&& hsrc, const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint64_t acc[NW];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const uint4 v = ld_chunk(xa, ch);
            acc[4 * ch + 0] = v.x; acc[4 * ch + 1] = v.y; acc[4 * ch + 2] = v.z; acc[4 * ch + 3] = v.w;
        }
#pragma unroll
        for (int j = NL; j < NW; ++j) acc[j] = 0;
        uint32_t zero8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) zero8[j] = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t hv[8], q[8];
            hsrc(blk, hv);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[NL + u] += hv[u];
            qchain<false>(acc, zero8, zero8, q, nm, n0inv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 2; c < NC; ++c) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[4 * c + k + u] += (uint64_t)nm[4 * c + k] * q[u];
                }
                if ((c & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            slide(acc);
            if ((blk % 3) == 2 && blk != NB - 1) normalize(acc);
        }
        wave_lds_fence();
        finish_to_lds(acc, xa);
        wave_lds_fence();
    }

    // canonicalise x (< 2M) in LDS into [0, M)
    PAI_DEV static void cond_sub(uint4* xa, const uint32_t* __restrict__ nm) {
        // pass 1: does x - M borrow?
        int32_t borrow = 0;
#pragma unroll 1
        for (int ch = 0; ch < NC; ++ch) {
            const uint4 v = ld_chunk(xa, ch);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int32_t t = (int32_t)w[k] - (int32_t)nm[4 * ch + k] + borrow;
                borrow = t >> RB;
            }
        }
        const bool ge = (borrow == 0);
        // pass 2: write x - M where x >= M
        int32_t b2 = 0;
#pragma unroll 1
        for (int ch = 0; ch < NC; ++ch) {
            const uint4 v = ld_chunk(xa, ch);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int32_t t = (int32_t)w[k] - (int32_t)nm[4 * ch + k] + b2;
                b2 = t >> RB;
                w[k] = ge ? ((uint32_t)t & RMASK) : w[k];
            }
            st_chunk(xa, ch, make_uint4(w[0], w[1], w[2], w[3]));
        }
        wave_lds_fence();
    }
};

}  // namespace pai
