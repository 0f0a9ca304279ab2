// ciphertext + ciphertext in the WIRE form by true division, one element per lane (round 6).
//
// CipherText::operator+ (bindings/ipcl_bindings_classes.cpp:318-321) is a b mod n^2 on canonical residues.  On lane groups it is two
// Montgomery products modulo n^2 (a b R^-1, then R^2: 16 units of 72^2 limb products at 2048-bit keys); kernels_paillier.hpp).
// Here the same residue comes from base-n digits and Barrett division with NO Montgomery factor anywhere:
//     (a1, a0) = divmod(a, n)   (b1, b0) = divmod(b, n)                         a = a0 + a1 n,  b = b0 + b1 n
//     (p1, p0) = divmod(a0 b0, n)                                               a b == p0 + (a0 b1 + a1 b0 + p1) n   (mod n^2)
//     s0 = (a0 b1 + a1 b0) mod n ;   h = (s0 + p1) mod n ;   out = p0 + h n     (< n^2: canonical without a final subtraction)
// Barrett (HAC 14.42, radix B = 2^29, K = NL - 1 limbs of n: B^(K-1) <= n < B^K, dividends below B^(2K)):
//     q1 = limbs K-1 .. 2K of x  (NL limbs) ;  q3 = floor(q1 mu / B^NL),  mu = floor(B^(2K) / n)  (NL limbs) — the HIGH half of one product
//     r  = (x mod B^NL) - (q3 n mod B^NL)  — the LOW half of one product, half its limb products — ;  r -= n at most three times (the high half
//          is cut two limbs below its first column, which can cost one more unit of q3), q3 follows
// On the one-element-per-lane engine a half product costs about half (0.63 / 0.58 at chunk granularity): 8.8 units wire -> wire (4 divisions of
// 1.21, products 1 + 2 + 1) against 16 — measured 3.2 ms per 2^20 against 3.5-3.7 (67 % multiplies, one wave per SIMD)
// — and on lane groups it would not (the lanes that own the unused columns idle), which is why this lives here.  Serves the key sizes
// whose n fills 71 limbs (2031 .. 2059 bits: the 2048-bit keys), large batches; every other case keeps the Montgomery kernels.
// Buffers per lane: two NL-limb digit buffers in LDS (A, B: the operand of the running product and the addend / remainder), six in a
// per-slot global scratch (coalesced uint4 columns, resident in the Infinity Cache), the accumulator window and one NL-limb value in registers.
#pragma once
#include "kernels_padic_enc.hpp"

namespace pai {

struct CtAddDivParams {
    const uint32_t* n29;         // n, NL limbs radix 29
    const uint32_t* mu29;        // floor(B^(2 (NL-1)) / n), NL limbs
    uint4* scratch;              // [6][NC][nslots]
    int ct_words;
};

enum DvHalf { DV_FULL = 0, DV_LOW = 1, DV_HIGH = 2 };

// acc-window product  X * d1 (+ Y * d2) + W, row blocks of U digits; the U limbs a block retires go to `sink(blk, low)`; the limbs above NL
// come back in hi.  DV_LOW: only the columns below NL are formed (the chunks of X that reach them) — hi is meaningless then.  DV_HIGH: only
// the chunks that reach column NL - 2 or above (two guard limbs): what is dropped is below 72 B^(NL-1), so hi is the high half of the
// product or one less (HAC 14.42's truncated q2); the retired low limbs are meaningless.
template <class E, int HALF, bool TWO, bool HAS_W, class D1, class D2, class Sink>
PAI_DEV void dv_product(uint32_t (&hi)[E::NC * 4], const uint32_t (&W)[E::NC * 4], const uint4* X, D1&& d1, const uint4* Y, D2&& d2,
                        Sink&& sink) {
    constexpr int NL = E::NC * 4, U = E::UC * 4;
    uint64_t acc[E::NW];
#pragma unroll
    for (int j = 0; j < NL; ++j) acc[j] = HAS_W ? W[j] : 0u;
#pragma unroll
    for (int j = NL; j < E::NW; ++j) acc[j] = 0;
    uint32_t xn[U], yn[U];                  // the NEXT block's digits: in flight while this block multiplies (one wave per SIMD hides nothing)
    d1(0, xn);
    if constexpr (TWO) d2(0, yn);
#pragma unroll 1
    for (int blk = 0; blk < E::NB; ++blk) {
        uint32_t xv[U], yv[U], low[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xv[u] = xn[u]; if constexpr (TWO) yv[u] = yn[u]; }
        if (blk + 1 < E::NB) {
            d1(blk + 1, xn);
            if constexpr (TWO) d2(blk + 1, yn);
        }
        const int climit = HALF == DV_LOW ? (NL - U * blk + 3) / 4 : E::NC;       // chunks of X that reach a column below NL
        const int hgap = NL - U * blk - U - 4;                                     // DV_HIGH: chunks below cfirst end under column NL - 2
        const int cfirst = (HALF == DV_HIGH && hgap > 0) ? (hgap + 3) / 4 : 0;
        // the operand chunks stream one ahead, with the scheduler fenced per chunk: left alone it hoists all 18 (36) chunk loads to the
        // top of the block and the accumulator window spills (mont_padic.hpp: block, wide windows)
        __builtin_amdgcn_sched_barrier(0);
        uint4 x_cur = E::ld(X, cfirst), y_cur = E::ld(TWO ? Y : X, cfirst);
#pragma unroll
        for (int c = 0; c < E::NC; ++c) {
            if (HALF == DV_FULL || (HALF == DV_LOW && c < climit) || (HALF == DV_HIGH && c >= cfirst)) {      // wave-uniform
                const int cn = (c + 1 < E::NC) ? c + 1 : c;
                const uint4 x_nxt = E::ld(X, cn), y_nxt = E::ld(TWO ? Y : X, cn);
                const uint32_t xa[4] = {x_cur.x, x_cur.y, x_cur.z, x_cur.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[4 * c + k + u] += (uint64_t)xa[k] * xv[u];
                }
                if constexpr (TWO) {
                    const uint32_t ya[4] = {y_cur.x, y_cur.y, y_cur.z, y_cur.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[4 * c + k + u] += (uint64_t)ya[k] * yv[u];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                x_cur = x_nxt;
                y_cur = y_nxt;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            low[u] = (uint32_t)acc[u] & RMASK;
            acc[u + 1] += acc[u] >> RB;
        }
        sink(blk, low);
        E::slide(acc);
        // a column takes U products of < 2^58 per block (2 U with TWO): 36 of them (three blocks; two with TWO) stay below 2^63.2 on top of a
        // normalised column — one carry sweep of the window per product instead of the engine's conservative cadence
        if (blk != E::NB - 1 && (blk + 1) % (TWO ? 2 : 3) == 0) E::normalize(acc);
    }
    if constexpr (HALF != DV_LOW) E::finish(acc, hi);
}

template <class E>
struct DvOps {
    static constexpr int NL = E::NC * 4, U = E::UC * 4;
    // a column of the per-slot global scratch through a buffer descriptor: the lane's part of the address is ONE 32-bit byte offset
    // (slot * 16), the chunk / buffer part a scalar offset — 64-bit per-chunk addresses, which the compiler keeps live across the
    // products, cost two registers per chunk and buffer (216 of them: spills)
    typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
    struct GBuf {
        __amdgpu_buffer_rsrc_t rsrc;
        uint32_t voff, sbase, sstride;
        PAI_DEV uint4 get(int c) const {
            const v4u_ t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(sbase + (uint32_t)c * sstride), 0);
            return make_uint4(t.x, t.y, t.z, t.w);
        }
        PAI_DEV void put(int c, uint4 v) const {
            v4u_ t;
            t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
            __builtin_amdgcn_raw_buffer_store_b128(t, rsrc, (int)voff, (int)(sbase + (uint32_t)c * sstride), 0);
        }
    };
    using MBuf = GBuf;
    // digits of a row block from a global scratch column / from a wave-uniform limb array
    static PAI_DEV auto from_buf(MBuf G) {
        return [=](int blk, uint32_t (&xv)[U]) {
#pragma unroll
            for (int c = 0; c < E::UC; ++c) {
                const uint4 t = G.get(E::UC * blk + c);
                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
            }
        };
    }
    static PAI_DEV auto uniform(const uint32_t* __restrict__ k) {
        return [=](int blk, uint32_t (&xv)[U]) { E::digits_uniform(k, blk, xv); };
    }
    static PAI_DEV void to_buf(MBuf G, const uint32_t (&r)[NL]) {
#pragma unroll
        for (int c = 0; c < E::NC; ++c) G.put(c, make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]));
    }
    static PAI_DEV void buf_to_lds(uint4* X, MBuf G) {       // every load in flight before the first LDS write (a lone wave hides no latency)
        uint4 t[E::NC];
#pragma unroll
        for (int c = 0; c < E::NC; ++c) t[c] = G.get(c);
#pragma unroll
        for (int c = 0; c < E::NC; ++c) E::st(X, c, t[c]);
    }
    static PAI_DEV void lds_to_regs(uint32_t (&r)[NL], const uint4* X) {
#pragma unroll
        for (int c = 0; c < E::NC; ++c) {
            const uint4 t = E::ld(X, c);
            r[4 * c] = t.x; r[4 * c + 1] = t.y; r[4 * c + 2] = t.z; r[4 * c + 3] = t.w;
        }
    }
    // x >= n ? x - n : x ; returns 1 when it subtracted
    static PAI_DEV uint32_t cond_sub_cnt(uint32_t (&x)[NL], const uint32_t* __restrict__ nm) {
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
        return ge ? 1u : 0u;
    }

    static PAI_DEV bool ge_mod(const uint32_t (&x)[NL], const uint32_t* __restrict__ nm) {       // x >= n
        int32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) borrow = ((int32_t)x[j] - (int32_t)nm[j] + borrow) >> RB;
        return borrow == 0;
    }
    // Barrett division of the 2 NL-limb dividend whose limbs K-1 .. 2K sit in A and whose low NL limbs sit in B (K = NL - 1).
    // Returns the remainder in r (registers) and leaves q3 (the quotient BEFORE its correction) in A; *corr = 0, 1 or 2 is to be added to it.
    static PAI_DEV void divmod(uint32_t (&r)[NL], uint32_t* corr, uint4* A, uint4* B, const uint32_t* __restrict__ nm,
                               const uint32_t* __restrict__ mu) {
        uint32_t none[NL];
        {
            uint32_t q3[NL];
            auto drop = [](int, const uint32_t (&)[U]) {};
            dv_product<E, DV_HIGH, false, false>(q3, none, A, uniform(mu), A, uniform(mu), drop);       // q3 = floor(q1 mu / B^NL), or one less
            wave_lds_fence();
            E::store_digit(A, q3);
            wave_lds_fence();
        }
        int32_t borrow = 0;
        auto sub_into_b = [&](int blk, const uint32_t (&low)[U]) {                                       // B <- B - (q3 n mod B^NL), block by block
#pragma unroll
            for (int c = 0; c < E::UC; ++c) {
                const uint4 t = E::ld(B, E::UC * blk + c);
                uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t d = (int32_t)w[k] - (int32_t)low[4 * c + k] + borrow;
                    borrow = d >> RB;
                    w[k] = (uint32_t)d & RMASK;
                }
                E::st(B, E::UC * blk + c, make_uint4(w[0], w[1], w[2], w[3]));
            }
        };
        dv_product<E, DV_LOW, false, false>(none, none, A, uniform(nm), A, uniform(nm), sub_into_b);
        wave_lds_fence();
        lds_to_regs(r, B);
        // q - 3 <= q3 <= q (HAC 14.42 with the truncated high product): up to three subtractions of n — the second and the third only when
        // some lane of the wave still needs one (a 72-limb compare instead of a compare-subtract-select; both are rare)
        uint32_t c = cond_sub_cnt(r, nm);
        if (__any(ge_mod(r, nm))) {
            c += cond_sub_cnt(r, nm);
            if (__any(ge_mod(r, nm))) c += cond_sub_cnt(r, nm);
        }
        *corr = c;
    }
    // q (registers) <- the quotient: q3 from A plus its correction
    static PAI_DEV void quotient(uint32_t (&q)[NL], const uint4* A, uint32_t corr) {
        lds_to_regs(q, A);
        uint32_t c = corr;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint32_t t = q[j] + c;
            q[j] = t & RMASK;
            c = t >> RB;
        }
    }
};

// out_i = a_i * b_i mod n^2, wire form in and out.  One workgroup per CU, one element per lane.
template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_ctadd_div(CtAddDivParams P, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, int n) {
    using E = Padic<NL, U, false>;
    using D = DvOps<E>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.n29[i]; ldsn[NL + i] = P.mu29[i]; }
    __syncthreads();
    const uint32_t* nm = ldsn;
    const uint32_t* mu = ldsn + NL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const uint32_t nslots = gridDim.x * BLOCK_THREADS;
    const uint32_t slot = blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(P.scratch, 0, (int)(6u * E::NC * nslots * 16u), 0x00020000);
    auto G = [&](int k) { return typename D::GBuf{rsrc, slot * 16u, (uint32_t)k * E::NC * nslots * 16u, nslots * 16u}; };
    constexpr int K = NL - 1;
    // the dividend of an operand row: limbs K-1 .. 2K into A, limbs 0 .. NL-1 into B.  The row (ROW_WORDS packed words, one row per lane:
    // 16-byte pieces 512 bytes apart) is read with all its loads in flight at once and cut into limbs in registers — limb by limb through
    // row_limb it was 288 dependent pairs of 4-byte loads, a third of the kernel's time.
    constexpr int ROW_WORDS = 128;
    static_assert(NL == 72, "row I/O is cut for 2048-bit keys (128-word ciphertexts; the host checks ct_words)");
    auto load_dividend = [&](const uint32_t* row) {
        uint32_t w[ROW_WORDS + 1];
        const uint4* r4 = reinterpret_cast<const uint4*>(row);
#pragma unroll
        for (int i = 0; i < ROW_WORDS / 4; ++i) {
            const uint4 t = r4[i];
            w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
        }
        w[ROW_WORDS] = 0;
        auto limb = [&](int J) -> uint32_t {                      // J, and with it every index and shift, is a compile-time constant after unrolling
            const int bit = RB * J, k = bit >> 5, sh = bit & 31;
            if (k >= ROW_WORDS) return 0u;
            const uint32_t lo = w[k] >> sh;
            const uint32_t hi = sh > 32 - RB ? (w[k + 1] << (32 - sh)) : 0u;
            return (lo | hi) & RMASK;
        };
#pragma unroll
        for (int c = 0; c < E::NC; ++c) {
            E::st(A, c, make_uint4(limb(K - 1 + 4 * c), limb(K - 1 + 4 * c + 1), limb(K - 1 + 4 * c + 2), limb(K - 1 + 4 * c + 3)));
            E::st(B, c, make_uint4(limb(4 * c), limb(4 * c + 1), limb(4 * c + 2), limb(4 * c + 3)));
        }
        wave_lds_fence();
    };
    // the dividend (lo in B already, hi in registers): limbs K-1 .. 2K into A;  lo70 / lo71 = limbs NL-2, NL-1 of the low half
    auto dividend_from = [&](const uint32_t (&hi)[NL], uint32_t lo70, uint32_t lo71) {
        uint32_t q1[NL];
        q1[0] = lo70; q1[1] = lo71;
#pragma unroll
        for (int j = 2; j < NL; ++j) q1[j] = hi[j - 2];
        wave_lds_fence();
        E::store_digit(A, q1);
        wave_lds_fence();
    };
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t r[NL], q[NL], none[NL];
        uint32_t corr;
        // ---- (b1, b0) = divmod(b, n): b0 -> G2, b1 -> G3 -------------------------------------------------------
        load_dividend(b + (size_t)es * P.ct_words);
        D::divmod(r, &corr, A, B, nm, mu);
        D::to_buf(G(2), r);
        D::quotient(q, A, corr);
        D::to_buf(G(3), q);
        // ---- (a1, a0) = divmod(a, n): a0 -> G0 and LDS A, a1 -> G1 ---------------------------------------------
        wave_lds_fence();
        load_dividend(a + (size_t)es * P.ct_words);
        D::divmod(r, &corr, A, B, nm, mu);
        D::quotient(q, A, corr);
        D::to_buf(G(1), q);
        D::to_buf(G(0), r);
        wave_lds_fence();
        E::store_digit(A, r);
        wave_lds_fence();
        // ---- P = a0 b0: low half -> B, high half in registers; (p1, p0) = divmod(P, n): p0 -> G4, p1 -> G5 -------
        {
            uint32_t hi[NL];
            auto lo_to_b = [&](int blk, const uint32_t (&low)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c) E::st(B, E::UC * blk + c, make_uint4(low[4 * c], low[4 * c + 1], low[4 * c + 2], low[4 * c + 3]));
            };
            dv_product<E, DV_FULL, false, false>(hi, none, A, D::from_buf(G(2)), A, D::from_buf(G(2)), lo_to_b);
            wave_lds_fence();
            const uint4 top = E::ld(B, E::NC - 1);
            dividend_from(hi, top.z, top.w);
        }
        D::divmod(r, &corr, A, B, nm, mu);
        D::to_buf(G(4), r);
        D::quotient(q, A, corr);
        D::to_buf(G(5), q);
        // ---- S = a0 b1 + a1 b0 (< 2 n^2): low half -> G0 (a0's copy is in LDS by then), high half in registers; s0 = S mod n ----
        wave_lds_fence();
        D::buf_to_lds(A, G(0));
        D::buf_to_lds(B, G(1));
        wave_lds_fence();
        {
            uint32_t hi[NL];
            const typename D::GBuf g0 = G(0);
            auto lo_to_g0 = [&](int blk, const uint32_t (&low)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c)
                    g0.put(E::UC * blk + c, make_uint4(low[4 * c], low[4 * c + 1], low[4 * c + 2], low[4 * c + 3]));
            };
            dv_product<E, DV_FULL, true, false>(hi, none, A, D::from_buf(G(3)), B, D::from_buf(G(2)), lo_to_g0);
            wave_lds_fence();
            D::buf_to_lds(B, g0);                                     // the low half is the remainder's minuend
            const uint4 top = g0.get(E::NC - 1);
            dividend_from(hi, top.z, top.w);
        }
        D::divmod(r, &corr, A, B, nm, mu);                            // r = s0
        // ---- h = (s0 + p1) mod n -------------------------------------------------------------------------------
        {
            const typename D::GBuf g5 = G(5);
            uint32_t c = 0;
#pragma unroll
            for (int cc = 0; cc < E::NC; ++cc) {
                const uint4 t = g5.get(cc);
                const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t v = r[4 * cc + k] + w[k] + c;
                    r[4 * cc + k] = v & RMASK;
                    c = v >> RB;
                }
            }
            (void)D::cond_sub_cnt(r, nm);
        }
        // ---- out = p0 + h n: low half -> B, high half -> A, packed words out -------------------------------------
        wave_lds_fence();
        E::store_digit(A, r);
        wave_lds_fence();
        {
            uint32_t hi[NL], w[NL];
            const typename D::GBuf g4 = G(4);
#pragma unroll
            for (int cc = 0; cc < E::NC; ++cc) {
                const uint4 t = g4.get(cc);
                w[4 * cc] = t.x; w[4 * cc + 1] = t.y; w[4 * cc + 2] = t.z; w[4 * cc + 3] = t.w;
            }
            auto lo_to_b = [&](int blk, const uint32_t (&low)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c) E::st(B, E::UC * blk + c, make_uint4(low[4 * c], low[4 * c + 1], low[4 * c + 2], low[4 * c + 3]));
            };
            dv_product<E, DV_FULL, false, true>(hi, w, A, D::uniform(nm), A, D::uniform(nm), lo_to_b);
            wave_lds_fence();
            // packed words out: words 0 .. 63 come from limbs 0 .. 70 (the low half, in B), words 64 .. 127 from limbs 70 .. 141
            uint32_t lo[NL];
            D::lds_to_regs(lo, B);
            auto word = [&](int k, auto&& limb_at) -> uint32_t {     // bits 32 k .. 32 k + 31 of the limb string
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)limb_at(j0) >> s0;
                t |= (uint64_t)limb_at(j0 + 1) << (RB - s0);
                if (2 * RB - s0 < 32) t |= (uint64_t)limb_at(j0 + 2) << (2 * RB - s0);
                return (uint32_t)t;
            };
            uint4* o4 = reinterpret_cast<uint4*>(out + (size_t)es * P.ct_words);
            auto lo_at = [&](int j) -> uint32_t { return j < NL ? lo[j] : 0u; };       // (words 0 .. 63 end inside limb 70)
            if (live) {
#pragma unroll
                for (int i = 0; i < ROW_WORDS / 8; ++i) o4[i] = make_uint4(word(4 * i, lo_at), word(4 * i + 1, lo_at), word(4 * i + 2, lo_at), word(4 * i + 3, lo_at));
            }
            auto all_at = [&](int j) -> uint32_t { return j < NL ? lo[j] : (j < 2 * NL ? hi[j - NL] : 0u); };
            if (live) {
#pragma unroll
                for (int i = ROW_WORDS / 8; i < ROW_WORDS / 4; ++i)
                    o4[i] = make_uint4(word(4 * i, all_at), word(4 * i + 1, all_at), word(4 * i + 2, all_at), word(4 * i + 3, all_at));
            }
        }
        wave_lds_fence();
    }
}

}  // namespace pai
