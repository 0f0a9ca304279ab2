// Arithmetic modulo p^2 on base-p digit pairs — the engine behind CRT-decrypt stage A.
//
// An element x of Z/p^2 is kept in "Montgomery digit form": two integers (a, b), each of NL limbs of
// 29 bits (lazy: < 2p + eps, never canonicalised inside the exponentiation), with
//        a + b p  ==  x R   (mod p^2),        R = 2^(29 NL)  >>  p   (R/p >= 2^20).
// Product rule.  For (a, b) ~ x and (c, d) ~ y let
//        w  = (a c + m p) / R                     -- Montgomery product mod p, quotient digits m
//        v  = (a d + b c - m + R p + m' p) / R    -- second Montgomery reduction mod p, quotient m'
// then  w + v p == (a + b p)(c + d p) R^-1 (mod p^2), i.e. (w, v) ~ x y.   Proof: a c = R w - m p and
// R v = a d + b c - m + (R + m') p, hence (a + b p)(c + d p) == a c + (a d + b c) p == R w + R v p.
// The b d p^2 term vanishes and every reduction is modulo p (NL limbs) instead of p^2 (2 NL limbs):
// a multiplication costs 5 NL^2 limb products instead of 8 NL^2, a squaring (every limb pair of a^2
// once, 2 a b once) 3.5 NL^2 instead of ~6.7 NL^2.  There is no division anywhere: the input is
// brought into digit form by applying the rule to its base-R digits and host-precomputed digit
// pairs of R^(i+2) mod p^2, and at the end the second digit of x^(p-1) IS Paillier's L function.
//
// Layout: one element per lane; the digit pair lives in LDS (chunk-major, see mont_wide.hpp); the
// accumulator window (NL + U lazy 64-bit columns, U = 12), the quotient digits m and the first
// result digit w live in VGPRs; p and p - 1 are wave-uniform (scalar loads).
#pragma once
#include "mont_dev.hpp"


namespace pai {

// XLDS: digit-buffer accesses through LDS-qualified pointers instead of leaving it to address-space inference.  One
// kernel shape (wide digits with quotient digits in scratch) otherwise compiles digit accesses to flat_load /
// flat_store; measured per kernel: 72-limb decrypt 364 -> 324 ms with it, 36-limb decrypt 476 -> 490 and 72-limb
// encrypt 67 -> 70 ms — hence a per-kernel switch.
template <int NL, int U, bool XLDS = false>
struct Padic {
    static_assert(NL % 4 == 0 && U % 4 == 0 && NL % U == 0 && U <= 16, "geometry");
    static constexpr int NC = NL / 4;        // four-limb chunks per digit
    static constexpr int UC = U / 4;         // chunks per row block
    static constexpr int NB = NL / U;        // row blocks
    static constexpr int NW = NL + U;        // accumulator window
    static constexpr int DIGIT_WORDS = NL * 64;   // LDS words of one digit for one wave
    static constexpr int P1 = (24 / U) * U;       // rows between normalisations at 2^59 per row
    static constexpr int P2 = (16 / U) * U;       // ... at 1.5 * 2^59 per row (doubled or three products)

    typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u_ lds_v4u_;
    PAI_DEV static uint4 ld(const uint4* x, int c) {
        if constexpr (XLDS) {
            const v4u_ v = *((const lds_v4u_*)(x + c * 64));
            return make_uint4(v.x, v.y, v.z, v.w);
        } else {
            return x[c * 64];
        }
    }
    PAI_DEV static void st(uint4* x, int c, uint4 v) {
        if constexpr (XLDS) {
            v4u_ t;
            t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
            *((lds_v4u_*)(x + c * 64)) = t;
        } else {
            x[c * 64] = v;
        }
    }

    PAI_DEV static void zero(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[j] = 0;
    }
    PAI_DEV static void normalize(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = NW - 1; j >= 1; --j) {
            const uint64_t keep = (j == NW - 1) ? acc[j] : (acc[j] & RMASK);
            acc[j] = keep + (acc[j - 1] >> RB);
        }
        acc[0] &= RMASK;
    }
    PAI_DEV static void slide(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = acc[j + U];
#pragma unroll
        for (int j = NL; j < NW; ++j) acc[j] = 0;
    }
    // U digits of an LDS operand for row block `blk`
    PAI_DEV static void digits(const uint4* x, int blk, uint32_t (&v)[U]) {
#pragma unroll
        for (int c = 0; c < UC; ++c) {
            const uint4 t = ld(x, UC * blk + c);
            v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
        }
    }
    PAI_DEV static void digits_uniform(const uint32_t* __restrict__ k, int blk, uint32_t (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = k[U * blk + u];
    }

    // One block of U rows:  acc += X * xv  (limbs of X below LO skipped, limbs >= HI taken twice)
    //                           (+ Y * yv)  + p * q ,  then the window slides by U columns.
    // q receives the block's quotient digits.  FEED: U limbs of a constant enter at the window top.
    // SYM (squaring, xv = limbs LO .. HI-1 of X itself): inside the diagonal class [LO, HI) the pair
    // (limb LO + d, row u) is taken once — skipped for d < u, single for d == u, doubled for d > u.
    template <bool HAS_X, int LO, int HI, bool HAS_Y, bool FEED, bool SYM = false>
    PAI_DEV static void block(uint64_t (&acc)[NW], const uint4* X, const uint32_t (&xv)[U], const uint4* Y,
                              const uint32_t (&yv)[U], const uint32_t* __restrict__ nm, uint32_t n0inv,
                              const uint32_t* __restrict__ feed, int blk, uint32_t (&q)[U]) {
        uint32_t xv2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv2[u] = xv[u] << 1;
        if constexpr (FEED) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[NL + u] += feed[U * blk + u];
        }
        // low U limbs of the operands (the quotient chain needs them)
        uint32_t x0[U], y0[U];
#pragma unroll
        for (int c = 0; c < UC; ++c) {
            if (HAS_X && LO < 4 * (c + 1)) {
                const uint4 t = ld(X, c);
                x0[4 * c] = t.x; x0[4 * c + 1] = t.y; x0[4 * c + 2] = t.z; x0[4 * c + 3] = t.w;
            } else {
                x0[4 * c] = x0[4 * c + 1] = x0[4 * c + 2] = x0[4 * c + 3] = 0;
            }
            if constexpr (HAS_Y) {
                const uint4 t = ld(Y, c);
                y0[4 * c] = t.x; y0[4 * c + 1] = t.y; y0[4 * c + 2] = t.z; y0[4 * c + 3] = t.w;
            } else {
                y0[4 * c] = y0[4 * c + 1] = y0[4 * c + 2] = y0[4 * c + 3] = 0;
            }
        }
        // multiplicity of the product (limb j of X) * (row u): 0, 1 or 2
        auto xmult = [](int j, int u) -> int {
            if (!HAS_X || j < LO) return 0;
            if (j >= HI) return 2;
            if (!SYM) return 1;
            return (j - LO < u) ? 0 : (j - LO == u ? 1 : 2);
        };
        auto xterm = [&](int j, int u) -> uint64_t {           // contribution of limb j (< U) of X in row u
            const int k = xmult(j, u);
            return k == 0 ? 0 : (uint64_t)x0[j] * (k == 2 ? xv2[u] : xv[u]);
        };
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j <= u; ++j) {
                if (xmult(j, u - j)) acc[u] += xterm(j, u - j);
                if constexpr (HAS_Y) acc[u] += (uint64_t)y0[j] * yv[u - j];
            }
#pragma unroll
            for (int j = 1; j <= u; ++j) acc[u] += (uint64_t)nm[j] * q[u - j];
            q[u] = ((uint32_t)acc[u] * n0inv) & RMASK;
            acc[u] += (uint64_t)nm[0] * q[u];
            acc[u + 1] += acc[u] >> RB;
        }
#pragma unroll
        for (int j = 1; j < U; ++j) {
#pragma unroll
            for (int u = U - j; u < U; ++u) {
                if (xmult(j, u)) acc[j + u] += xterm(j, u);
                if constexpr (HAS_Y) acc[j + u] += (uint64_t)y0[j] * yv[u];
                acc[j + u] += (uint64_t)nm[j] * q[u];
            }
        }
        // remaining chunks: a pure rank-(2U or 3U) update
        if constexpr (NL <= 48) {
#pragma unroll
            for (int c = UC; c < NC; ++c) {
                uint32_t xa[4] = {0, 0, 0, 0}, ya[4] = {0, 0, 0, 0};
                if constexpr (HAS_X) {
                    if (4 * c >= LO) { const uint4 t = ld(X, c); xa[0] = t.x; xa[1] = t.y; xa[2] = t.z; xa[3] = t.w; }
                }
                if constexpr (HAS_Y) { const uint4 t = ld(Y, c); ya[0] = t.x; ya[1] = t.y; ya[2] = t.z; ya[3] = t.w; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int mult = xmult(4 * c + k, u);
                        if (mult) acc[4 * c + k + u] += (uint64_t)xa[k] * (mult == 2 ? xv2[u] : xv[u]);
                        if constexpr (HAS_Y) acc[4 * c + k + u] += (uint64_t)ya[k] * yv[u];
                        acc[4 * c + k + u] += (uint64_t)nm[4 * c + k] * q[u];
                    }
                }
            }
        } else {
            // wide windows (NL = 72): stream the operands one chunk ahead and fence the scheduler per chunk so
            // that the chunk loads are not all hoisted to the top (which would spill the accumulator window)
            __builtin_amdgcn_sched_barrier(0);
            uint4 x_cur = ld(X, UC), y_cur = ld(Y, UC);
            uint32_t n_cur[4] = {nm[4 * UC], nm[4 * UC + 1], nm[4 * UC + 2], nm[4 * UC + 3]};
#pragma unroll
            for (int c = UC; c < NC; ++c) {
                const int cn = (c + 1 < NC) ? c + 1 : c;
                const uint4 x_nxt = ld(X, cn), y_nxt = ld(Y, cn);
                const uint32_t n_nxt[4] = {nm[4 * cn], nm[4 * cn + 1], nm[4 * cn + 2], nm[4 * cn + 3]};
                const uint32_t xa[4] = {x_cur.x, x_cur.y, x_cur.z, x_cur.w};
                const uint32_t ya[4] = {y_cur.x, y_cur.y, y_cur.z, y_cur.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int mult = xmult(4 * c + k, u);
                        if (mult) acc[4 * c + k + u] += (uint64_t)xa[k] * (mult == 2 ? xv2[u] : xv[u]);
                        if constexpr (HAS_Y) acc[4 * c + k + u] += (uint64_t)ya[k] * yv[u];
                        acc[4 * c + k + u] += (uint64_t)n_cur[k] * q[u];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                x_cur = x_nxt;
                y_cur = y_nxt;
#pragma unroll
                for (int k = 0; k < 4; ++k) n_cur[k] = n_nxt[k];
            }
        }
        slide(acc);
    }

    // carry-propagate the low NL columns into canonical 29-bit limbs (registers)
    PAI_DEV static void finish(const uint64_t (&acc)[NW], uint32_t (&r)[NL]) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint64_t t = acc[j] + c;
            r[j] = (uint32_t)t & RMASK;
            c = t >> RB;
        }
    }
    PAI_DEV static void store_digit(uint4* x, const uint32_t (&r)[NL]) {
#pragma unroll
        for (int c = 0; c < NC; ++c) st(x, c, make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]));
    }

    // ---- first half of the product rule: w = (X * C + m p) / R, quotient digits m -----------------
    // CSrc: functor (blk, xv) giving the U digits of C for a row block.
    // The quotient digits m are parked in LDS (digit buffer M) so that the row-block loop can stay rolled
    // (a rolled loop cannot index a register array) — the hot loop must fit the 64 KB instruction cache.
    // M may be an LDS digit buffer (stride 64 uint4) or a global scratch column (stride = number of slots)
    struct MBuf {
        uint4* p;
        size_t stride;
    };
    PAI_DEV static void store_q(MBuf M, int blk, const uint32_t (&q)[U]) {
#pragma unroll
        for (int c = 0; c < UC; ++c)
            M.p[(size_t)(UC * blk + c) * M.stride] = make_uint4(q[4 * c], q[4 * c + 1], q[4 * c + 2], q[4 * c + 3]);
    }
    template <class CSrc>
    PAI_DEV static void mm1_mul(uint32_t (&w)[NL], MBuf M, const uint4* X, CSrc&& csrc,
                                const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint64_t acc[NW];
        zero(acc);
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
        // the digits of the next row block are fetched while the current block is computed (they may come
        // from global memory: table entries)
        uint32_t xn[U];
        csrc(0, xn);
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t xv[U], q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = xn[u];
            csrc(blk + 1 < NB ? blk + 1 : blk, xn);
            block<true, 0, NL, false, false>(acc, X, xv, X, dummy, nm, n0inv, nm, blk, q);
            store_q(M, blk, q);
            if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc);    // 2^59 per row: at most 24 rows
        }
        finish(acc, w);
    }
    // squaring: C = X, symmetric by limb classes of U limbs (row block b multiplies limbs >= U b)
    // Column bound: a column of X^2 holds at most NL/2 doubled pairs (2^59 each) and one square over the WHOLE
    // pass, plus one p*q product (2^58) per row.  For NL <= 40 a single normalisation after about half the rows
    // keeps both halves below 2^64: (NL/2 + 1/2) 2^59 + r 2^58 < 2^64 for r <= 24.  Wider digits use the
    // per-row rule (1.5 * 2^59 per row, at most 16 rows).
    PAI_DEV static constexpr bool sqr1_normalize_after(int B) {
        if (B == NB - 1) return false;
        if (NL <= 40) return (B + 1) * U <= 24 && (B + 2) * U > 24;
        return ((B + 1) * U) % P2 == 0;
    }
    template <int B>
    PAI_DEV static void mm1_sqr_blocks(uint64_t (&acc)[NW], MBuf M, const uint4* X,
                                       const uint32_t* __restrict__ nm, uint32_t n0inv) {
        if constexpr (B < NB) {
            uint32_t xv[U], q[U], dummy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dummy[u] = 0;
            digits(X, B, xv);
            block<true, U * B, U * (B + 1), false, false, true>(acc, X, xv, X, dummy, nm, n0inv, nm, B, q);
            store_q(M, B, q);
            if (sqr1_normalize_after(B)) normalize(acc);
            mm1_sqr_blocks<B + 1>(acc, M, X, nm, n0inv);
        }
    }
    PAI_DEV static void mm1_sqr(uint32_t (&w)[NL], MBuf M, const uint4* X,
                                const uint32_t* __restrict__ nm, uint32_t n0inv) {
        uint64_t acc[NW];
        zero(acc);
        mm1_sqr_blocks<0>(acc, M, X, nm, n0inv);
        finish(acc, w);
    }

    // ---- second half: v = (X * D + Y * C - m + R p + m' p) / R --------------------------------------
    // initial window = (R - 1 - m) + 1 in the low NL columns; p - 1 enters at the top (pm1 = limbs of p - 1)
    PAI_DEV static void mm2_init(uint64_t (&acc)[NW], MBuf M) {
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint4 t = M.p[(size_t)c * M.stride];
            acc[4 * c] = (uint64_t)(RMASK - t.x); acc[4 * c + 1] = (uint64_t)(RMASK - t.y);
            acc[4 * c + 2] = (uint64_t)(RMASK - t.z); acc[4 * c + 3] = (uint64_t)(RMASK - t.w);
        }
        acc[0] += 1;
#pragma unroll
        for (int j = NL; j < NW; ++j) acc[j] = 0;
    }
    template <class DSrc, class CSrc>
    PAI_DEV static void mm2_mul(uint32_t (&v)[NL], MBuf M, const uint4* X, const uint4* Y, DSrc&& dsrc,
                                CSrc&& csrc, const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        uint64_t acc[NW];
        uint32_t xn[U], yn[U];
        dsrc(0, xn);
        csrc(0, yn);
        mm2_init(acc, M);
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t xv[U], yv[U], q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { xv[u] = xn[u]; yv[u] = yn[u]; }
            dsrc(blk + 1 < NB ? blk + 1 : blk, xn);
            csrc(blk + 1 < NB ? blk + 1 : blk, yn);
            block<true, 0, NL, true, true>(acc, X, xv, Y, yv, nm, n0inv, pm1, blk, q);
            if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc);     // 1.5 * 2^59 per row: at most 16 rows
        }
        finish(acc, v);
    }
    // squaring: v = (2 X * Y - m + R p + m' p) / R   (X = first digit, Y = second digit of the same element)
    PAI_DEV static void mm2_sqr(uint32_t (&v)[NL], MBuf M, const uint4* X, const uint4* Y,
                                const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        uint64_t acc[NW];
        mm2_init(acc, M);
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
        // unrolled (three 12-row blocks at 36 limbs): 2 % faster inside the decrypt kernel (476 vs 486 ms); unrolling the
        // product's loops as well overflows the instruction cache (511 ms)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t xv[U], q[U];
            digits(Y, blk, xv);
            block<true, 0, 0, false, true>(acc, X, xv, X, dummy, nm, n0inv, pm1, blk, q);   // HI = 0: every limb doubled
            if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc);
        }
        finish(acc, v);
    }

    // Variant for wide digits (NL = 72): the first result digit w is parked in a strided scratch buffer (same
    // shape as M) instead of 72 registers, so that nothing but the accumulator window is live inside the loops.
    PAI_DEV static void finish_to_buf(const uint64_t (&acc)[NW], MBuf Wb) {
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
            Wb.p[(size_t)ch * Wb.stride] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    template <class CSrc, class DSrc>
    PAI_DEV static void mul_wbuf(uint4* A, uint4* B, MBuf M, MBuf Wb, CSrc&& csrc, DSrc&& dsrc,
                                 const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        {   // first half: w -> Wb, quotient digits -> M
            uint64_t acc[NW];
            zero(acc);
            uint32_t dummy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dummy[u] = 0;
            uint32_t xn[U];
            csrc(0, xn);
#pragma unroll 1
            for (int blk = 0; blk < NB; ++blk) {
                uint32_t xv[U], q[U];
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = xn[u];
                csrc(blk + 1 < NB ? blk + 1 : blk, xn);
                block<true, 0, NL, false, false>(acc, A, xv, A, dummy, nm, n0inv, nm, blk, q);
                store_q(M, blk, q);
                if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc);
            }
            finish_to_buf(acc, Wb);
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // second half: v -> B (after the last read of B), then w -> A
            uint64_t acc[NW];
            uint32_t xn[U], yn[U];
            dsrc(0, xn);
            csrc(0, yn);
            mm2_init(acc, M);
#pragma unroll 1
            for (int blk = 0; blk < NB; ++blk) {
                uint32_t xv[U], yv[U], q[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { xv[u] = xn[u]; yv[u] = yn[u]; }
                dsrc(blk + 1 < NB ? blk + 1 : blk, xn);
                csrc(blk + 1 < NB ? blk + 1 : blk, yn);
                block<true, 0, NL, true, true>(acc, A, xv, B, yv, nm, n0inv, pm1, blk, q);
                if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc);
            }
            wave_lds_fence();
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
                st(B, ch, make_uint4(w[0], w[1], w[2], w[3]));
                st(A, ch, Wb.p[(size_t)ch * Wb.stride]);
            }
            wave_lds_fence();
        }
    }

    // second result digit over B, the parked first digit back from the scratch column over A
    PAI_DEV static void finish_into(const uint64_t (&acc)[NW], uint4* Bdst, uint4* Adst, MBuf Wb) {
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
            st(Bdst, ch, make_uint4(w[0], w[1], w[2], w[3]));
            st(Adst, ch, Wb.p[(size_t)ch * Wb.stride]);
        }
    }

    // Squaring for wide digits with BOTH halves in rolled loops (nothing to overflow the instruction cache): the first
    // half is the product loop with the element's own first digit as multiplier (a * a, every limb pair twice — the
    // limb-class symmetric first half is fully unrolled code), the second half is the squaring's own
    // v = (2 a b - m + R p + m' p) / R with doubled multiplier digits: ONE pass over a instead of the two (a d + b c)
    // of the product rule.  4 NL^2 limb products per squaring instead of the 5 NL^2 of mul_wbuf(x, x).
    PAI_DEV static void sqr_rolled_wbuf(uint4* A, uint4* B, MBuf M, MBuf Wb, const uint32_t* __restrict__ nm,
                                        const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        {
            uint64_t acc[NW];
            zero(acc);
            uint32_t dummy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dummy[u] = 0;
            uint32_t xn[U];
            digits(A, 0, xn);
#pragma unroll 1
            for (int blk = 0; blk < NB; ++blk) {
                uint32_t xv[U], q[U];
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = xn[u];
                digits(A, blk + 1 < NB ? blk + 1 : blk, xn);
                block<true, 0, NL, false, false>(acc, A, xv, A, dummy, nm, n0inv, nm, blk, q);
                store_q(M, blk, q);
                if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc);
            }
            finish_to_buf(acc, Wb);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            uint64_t acc[NW];
            mm2_init(acc, M);
            uint32_t dummy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dummy[u] = 0;
#pragma unroll 1
            for (int blk = 0; blk < NB; ++blk) {
                uint32_t xv[U], q[U];
                digits(B, blk, xv);
                block<true, 0, 0, false, true>(acc, A, xv, A, dummy, nm, n0inv, pm1, blk, q);   // HI = 0: every limb doubled
                if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc);
            }
            wave_lds_fence();
            finish_into(acc, B, A, Wb);
            wave_lds_fence();
        }
    }

    // The same with a limb-class symmetric first half that still fits the instruction cache: per row block, the
    // a-part (the element's own limbs >= U b against its digits of class b: diagonal class once per pair, classes above
    // doubled) is one of NB specialised straight-line routines selected by a wave-uniform switch — NL (NL + U) / 2 limb
    // products of code in total — and the reduction part (quotient chain and p * q, identical for every block) is the
    // shared rolled body.  3.5 NL^2 limb products per squaring.
    template <int B>
    PAI_DEV static void sqr_apart(uint64_t (&acc)[NW], const uint4* X, const uint32_t (&xv)[U]) {
        constexpr int LO = U * B, HI = U * (B + 1);
        uint32_t xv2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv2[u] = xv[u] << 1;
#pragma unroll
        for (int c = LO / 4; c < NC; ++c) {
            const uint4 t = ld(X, c);
            const uint32_t xa[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = 4 * c + k;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int mult = j >= HI ? 2 : (j - LO < u ? 0 : (j - LO == u ? 1 : 2));
                    if (mult) acc[j + u] += (uint64_t)xa[k] * (mult == 2 ? xv2[u] : xv[u]);
                }
            }
        }
    }
    template <int B>
    PAI_DEV static void sqr_apart_dispatch(uint64_t (&acc)[NW], const uint4* X, const uint32_t (&xv)[U], int blk) {
        if (blk == B) sqr_apart<B>(acc, X, xv);
        else if constexpr (B + 1 < NB) sqr_apart_dispatch<B + 1>(acc, X, xv, blk);
    }

    // ---- both halves of the product rule in ONE pass over the row blocks -------------------------------------------
    // Row block k of the first half produces the quotient digits m[U k .. U k + U) — exactly the columns the second
    // half's window starts at in ITS row block k — so the two accumulator windows advance together: the quotient digits
    // go from registers straight into the second window as (2^29 - 1 - m_j) and never exist as a stored number, and the
    // first result digit w stays in its window until v is finished as well, so nothing is parked either.  Wide digits
    // (NL = 56 / 72) otherwise bounce both through strided HBM scratch (MBuf): 2 NL limbs written and read per product.
    // Price: two windows of NL + U lazy columns live at once (4 (NL + U) VGPRs; one wave per SIMD has 512).
    PAI_DEV static void finish_pair(const uint64_t (&acc1)[NW], const uint64_t (&acc2)[NW], uint4* Adst, uint4* Bdst) {
        uint64_t c1 = 0, c2 = 0;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            uint32_t w[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t t1 = acc1[4 * ch + k] + c1;
                w[k] = (uint32_t)t1 & RMASK;
                c1 = t1 >> RB;
                const uint64_t t2 = acc2[4 * ch + k] + c2;
                v[k] = (uint32_t)t2 & RMASK;
                c2 = t2 >> RB;
            }
            st(Adst, ch, make_uint4(w[0], w[1], w[2], w[3]));
            st(Bdst, ch, make_uint4(v[0], v[1], v[2], v[3]));
        }
    }
    template <class CSrc, class DSrc>
    PAI_DEV static void mul_fused(uint4* A, uint4* B, CSrc&& csrc, DSrc&& dsrc, const uint32_t* __restrict__ nm,
                                  const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        uint64_t acc1[NW], acc2[NW];
        zero(acc1);
        zero(acc2);
        acc2[0] = 1;                          // (R - 1 - m) + 1
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
        uint32_t cn[U], dn[U];
        csrc(0, cn);
        dsrc(0, dn);
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t cv[U], dv[U], q1[U], q2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { cv[u] = cn[u]; dv[u] = dn[u]; }
            csrc(blk + 1 < NB ? blk + 1 : blk, cn);
            dsrc(blk + 1 < NB ? blk + 1 : blk, dn);
            block<true, 0, NL, false, false>(acc1, A, cv, A, dummy, nm, n0inv, nm, blk, q1);
#pragma unroll
            for (int u = 0; u < U; ++u) acc2[u] += (uint64_t)(RMASK - q1[u]);
            block<true, 0, NL, true, true>(acc2, A, dv, B, cv, nm, n0inv, pm1, blk, q2);
            if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc1);
            if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc2);
        }
        wave_lds_fence();
        finish_pair(acc1, acc2, A, B);
        wave_lds_fence();
    }
    // (A, B) <- (A, B) * (C, 0): the product rule with a right operand whose SECOND digit is zero — w = (a c + m p) / R,
    // v = (b c - m + R p + m' p) / R: 4 NL^2 limb products instead of 5.  The fixed-base tables store such operands
    // (g-factored entries, kernels_padic_enc.hpp): x = (c, 0) * (1 + p)^t with the exponents t summed on the side.
    template <class CSrc>
    PAI_DEV static void mul_fused_c0(uint4* A, uint4* B, CSrc&& csrc, const uint32_t* __restrict__ nm,
                                     const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        uint64_t acc1[NW], acc2[NW];
        zero(acc1);
        zero(acc2);
        acc2[0] = 1;                          // (R - 1 - m) + 1
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
        uint32_t cn[U];
        csrc(0, cn);
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t cv[U], q1[U], q2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cv[u] = cn[u];
            csrc(blk + 1 < NB ? blk + 1 : blk, cn);
            block<true, 0, NL, false, false>(acc1, A, cv, A, dummy, nm, n0inv, nm, blk, q1);
#pragma unroll
            for (int u = 0; u < U; ++u) acc2[u] += (uint64_t)(RMASK - q1[u]);
            block<false, 0, NL, true, true>(acc2, A, dummy, B, cv, nm, n0inv, pm1, blk, q2);
            if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) { normalize(acc1); normalize(acc2); }   // two products per row and column each
        }
        wave_lds_fence();
        finish_pair(acc1, acc2, A, B);
        wave_lds_fence();
    }
    // squaring: first half a * a (every limb pair twice), second half 2 a b with doubled multiplier digits — 4 NL^2
    PAI_DEV static void sqr_fused(uint4* A, uint4* B, const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1,
                                  uint32_t n0inv) {
        uint64_t acc1[NW], acc2[NW];
        zero(acc1);
        zero(acc2);
        acc2[0] = 1;
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t av[U], bv[U], q1[U], q2[U];
            digits(A, blk, av);
            digits(B, blk, bv);
            block<true, 0, NL, false, false>(acc1, A, av, A, dummy, nm, n0inv, nm, blk, q1);
#pragma unroll
            for (int u = 0; u < U; ++u) acc2[u] += (uint64_t)(RMASK - q1[u]);
            block<true, 0, 0, false, true>(acc2, A, bv, A, dummy, nm, n0inv, pm1, blk, q2);   // HI = 0: every limb doubled
            if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc1);
            if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc2);
        }
        wave_lds_fence();
        finish_pair(acc1, acc2, A, B);
        wave_lds_fence();
    }

    // fused counterpart of sqr_sym_wbuf: limb-class symmetric first half (3.5 NL^2)
    PAI_DEV static void sqr_sym_fused(uint4* A, uint4* B, const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1,
                                      uint32_t n0inv) {
        uint64_t acc1[NW], acc2[NW];
        zero(acc1);
        zero(acc2);
        acc2[0] = 1;
        uint32_t dummy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dummy[u] = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t av[U], bv[U], q1[U], q2[U];
            digits(A, blk, av);
            digits(B, blk, bv);
            sqr_apart_dispatch<0>(acc1, A, av, blk);
            __builtin_amdgcn_sched_barrier(0);
            block<false, 0, NL, false, false>(acc1, A, dummy, A, dummy, nm, n0inv, nm, blk, q1);
#pragma unroll
            for (int u = 0; u < U; ++u) acc2[u] += (uint64_t)(RMASK - q1[u]);
            block<true, 0, 0, false, true>(acc2, A, bv, A, dummy, nm, n0inv, pm1, blk, q2);
            if (sqr1_normalize_after(blk)) normalize(acc1);
            if (blk != NB - 1 && ((blk + 1) * U) % P2 == 0) normalize(acc2);
        }
        wave_lds_fence();
        finish_pair(acc1, acc2, A, B);
        wave_lds_fence();
    }
    // compile-time choice between the scratch-parked and the fused forms (per kernel: the fused form trades the scratch
    // round trips for register pressure)
    template <bool FUSED, class CSrc, class DSrc>
    PAI_DEV static void mul_w(uint4* A, uint4* B, MBuf M, MBuf Wb, CSrc&& csrc, DSrc&& dsrc, const uint32_t* __restrict__ nm,
                              const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        if constexpr (FUSED) mul_fused(A, B, csrc, dsrc, nm, pm1, n0inv);
        else mul_wbuf(A, B, M, Wb, csrc, dsrc, nm, pm1, n0inv);
    }
    template <bool FUSED>
    PAI_DEV static void sqr_rolled_w(uint4* A, uint4* B, MBuf M, MBuf Wb, const uint32_t* __restrict__ nm,
                                     const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        if constexpr (FUSED) sqr_fused(A, B, nm, pm1, n0inv);
        else sqr_rolled_wbuf(A, B, M, Wb, nm, pm1, n0inv);
    }

    // (A, B) <- (A, B)^2
    PAI_DEV static void sqr(uint4* A, uint4* B, MBuf M, const uint32_t* __restrict__ nm, const uint32_t* __restrict__ pm1,
                            uint32_t n0inv) {
        uint32_t w[NL], v[NL];
        mm1_sqr(w, M, A, nm, n0inv);
        mm2_sqr(v, M, A, B, nm, pm1, n0inv);
        wave_lds_fence();
        store_digit(A, w);
        store_digit(B, v);
        wave_lds_fence();
    }
    // (A, B) <- (A, B) * (C, D), the second operand's digits supplied per row block
    template <class CSrc, class DSrc>
    PAI_DEV static void mul(uint4* A, uint4* B, MBuf M, CSrc&& csrc, DSrc&& dsrc, const uint32_t* __restrict__ nm,
                            const uint32_t* __restrict__ pm1, uint32_t n0inv) {
        uint32_t w[NL], v[NL];
        mm1_mul(w, M, A, csrc, nm, n0inv);
        mm2_mul(v, M, A, B, dsrc, csrc, nm, pm1, n0inv);
        wave_lds_fence();
        store_digit(A, w);
        store_digit(B, v);
        wave_lds_fence();
    }

    // Plain product with addend, no reduction:  X * D + W  (X in LDS, D digits per block, W < 2^(29 NL) in registers).
    // The low NL limbs are written to the LDS digit buffer LO (chunk-major), the high NL limbs returned in hi.
    template <class DSrc>
    PAI_DEV static void mul_plain(uint32_t (&hi)[NL], uint4* LO, const uint32_t (&W)[NL], const uint4* X, DSrc&& dsrc) {
        uint64_t acc[NW];
#pragma unroll
        for (int j = 0; j < NL; ++j) acc[j] = W[j];
#pragma unroll
        for (int j = NL; j < NW; ++j) acc[j] = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            uint32_t xv[U], low[U];
            dsrc(blk, xv);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint4 t = ld(X, c);
                const uint32_t xa[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[4 * c + k + u] += (uint64_t)xa[k] * xv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                low[u] = (uint32_t)acc[u] & RMASK;
                acc[u + 1] += acc[u] >> RB;
            }
#pragma unroll
            for (int c = 0; c < UC; ++c)
                st(LO, UC * blk + c, make_uint4(low[4 * c], low[4 * c + 1], low[4 * c + 2], low[4 * c + 3]));
            slide(acc);
            if (blk != NB - 1 && ((blk + 1) * U) % P1 == 0) normalize(acc);
        }
        finish(acc, hi);
    }

    // x (NL limbs, < 4p say) -> canonical [0, p) by up to `times` conditional subtractions, in registers
    PAI_DEV static void cond_sub(uint32_t (&x)[NL], const uint32_t* __restrict__ nm) {
        uint32_t d[NL];
        int32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int32_t t = (int32_t)x[j] - (int32_t)nm[j] + borrow;
            d[j] = (uint32_t)t & RMASK;
            borrow = t >> RB;
        }
        const bool ge = (borrow == 0);
#pragma unroll
        for (int j = 0; j < NL; ++j) x[j] = ge ? d[j] : x[j];
    }
};

}  // namespace pai
