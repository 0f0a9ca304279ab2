// CRT-decrypt stage A on the p-adic digit engine (mont_padic.hpp):
//   for s in {p, q} (blockIdx.y):  L_s = L_s( ct^(s-1) mod s^2 ) = ((ct^(s-1) mod s^2) - 1) / s      (< s)
// written as packed words into u_out[which][i][0 .. u_words) (upper words zero).  Stage B then only has
// to multiply by h_s and recombine (DecBParams::u_is_L).  Replaces, for keys whose primes fit NL limbs,
// the modexp modulo s^2 of k_dec_a / k_dec_a_wide with arithmetic modulo s on digit pairs.
#pragma once
#include "kernels_wide.hpp"
#include "mont_padic.hpp"

namespace pai {

struct DecPadicParams {
    const MontCtx* pr[2];        // moduli p, q (NL limbs, R = 2^(29 NL))
    const uint32_t* pm1[2];      // p - 1, q - 1 as radix-29 limbs (NLMAX padded)
    const uint32_t* kdig[2];     // [ND][2][NL]: digit pairs of R^(i+2) mod s^2, i = 0 .. ND-1
    const uint16_t* ops[2];      // sliding-window schedule of s - 1: entries (squarings | table index << 8), 0xFF = no multiply
    int nops[2];
    int tbl_entries;             // odd powers base^(2i+1), i < tbl_entries; slot tbl_entries holds base^2
    int nd;                      // base-R digits of a ciphertext
    uint4* wscratch;             // PADIC_WBUF: [NC][nslots] quotient digits of the digit-form entry and the exit
    int ct_words, u_words;
};

// (A, B) <- Montgomery digit form of the packed integer `row` (row_words 32-bit words, any value < s^2 R-ish):
// sum_i (c_i, 0) * digits(R^(i+2) mod s^2) over its base-R digits c_i (kdig = [nd][2][NL] host-precomputed pairs).
// M is the quotient-digit buffer of the engine (LDS or strided scratch).
template <class E>
PAI_DEV void padic_to_digit_form(uint4* A, uint4* B, typename E::MBuf M, const uint32_t* __restrict__ row, int row_words,
                                 const uint32_t* __restrict__ kdig, int nd, const uint32_t* __restrict__ nm,
                                 const uint32_t* __restrict__ pm1, uint32_t n0inv) {
    constexpr int NL = E::NC * 4, U = E::UC * 4;
    uint32_t sw[NL], sv[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) { sw[j] = 0; sv[j] = 0; }
#pragma unroll 1
    for (int i = 0; i < nd; ++i) {
        wave_lds_fence();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) {
            uint32_t t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = row_limb(row, row_words, NL * i + 4 * c + k);
            E::st(A, c, make_uint4(t[0], t[1], t[2], t[3]));
        }
        wave_lds_fence();
        const uint32_t* __restrict__ ka = kdig + (size_t)(2 * i) * NL;
        const uint32_t* __restrict__ kb = ka + NL;
        uint32_t w[NL], v[NL];
        E::mm1_mul(w, M, A, [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(ka, blk, xv); }, nm, n0inv);
        {   // v = (c_i * kb - m + R p + m' p) / R   (the element's second digit is zero)
            uint64_t acc[E::NW];
            E::mm2_init(acc, M);
            uint32_t dummy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dummy[u] = 0;
#pragma unroll 1
            for (int blk = 0; blk < E::NB; ++blk) {
                uint32_t xv[U], q[U];
                E::digits_uniform(kb, blk, xv);
                E::template block<true, 0, NL, false, true>(acc, A, xv, A, dummy, nm, n0inv, pm1, blk, q);
                if (blk != E::NB - 1) E::normalize(acc);
            }
            E::finish(acc, v);
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) { sw[j] += w[j]; sv[j] += v[j]; }
    }
    // limb sums (< 2^32) back to 29-bit limbs
    uint32_t cw = 0, cv = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const uint64_t tw = (uint64_t)sw[j] + cw, tv = (uint64_t)sv[j] + cv;
        sw[j] = (uint32_t)tw & RMASK; cw = (uint32_t)(tw >> RB);
        sv[j] = (uint32_t)tv & RMASK; cv = (uint32_t)(tv >> RB);
    }
    wave_lds_fence();
    E::store_digit(A, sw);
    E::store_digit(B, sv);
    wave_lds_fence();
}

// MODE PADIC_LDS_M: digit pair + quotient digits in LDS (36-limb primes: 3 x 36 KB per workgroup).
// MODE PADIC_WBUF:  LDS holds only the digit pair (wider primes: 2 x 56 / 2 x 72 KB per workgroup); the two halves of the
//                   product rule run fused (mont_padic.hpp: mul_fused / sqr_fused / sqr_sym_fused — quotient digits go from
//                   registers straight into the second window), only the digit-form entry and the exit park their
//                   quotient digits in a strided global scratch column.
// Both run one wave per SIMD.  Measured and dropped (profiles/r04/: bench_dec_regm.json, bench_dec_comba.jsonl,
// dec72_variants.jsonl; the code is in the history): two workgroups per CU with the quotient digits in registers (527.9 vs
// 476.6 ms per 2^20), product-scanning squarings entirely in registers (468-478 ms), 4-row blocks (506.9 ms), the unfused
// scratch-parked forms at 56 / 72 limbs.
constexpr int PADIC_XLDS_FROM = 56;          // LDS-qualified digit accesses from this limb count on (mont_padic.hpp: XLDS)
// Wide digits square through rolled loops (a fully unrolled limb-class symmetric squaring is 62 KB of code at 72 limbs,
// beyond the instruction cache): both halves in one pass, 4 NL^2 limb products instead of the product rule's 5.  Up to 56
// limbs the first half is additionally limb-class symmetric (specialised a-parts behind a wave-uniform switch, 3.5 NL^2:
// 121.0 -> 112.7 ms per 65 536 at 3072-bit keys); at 72 limbs the same code LOSES (282.6 -> 383.4 ms: the nine specialised
// a-parts next to the 160-register window no longer fit), so 4096-bit keys keep the plain rolled first half.
constexpr int PADIC_SQR_SYM_MAX_NL = 56;
constexpr int PADIC_LDS_M = 0, PADIC_WBUF = 1;
template <int NL, int U, int WB, int MODE>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_dec_a_padic(DecPadicParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out, int n,
              uint4* __restrict__ table) {
    using E = Padic<NL, U, (NL >= PADIC_XLDS_FROM)>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int which = blockIdx.y;
    const MontCtx* ctx = P.pr[which];
    // modulus and s - 1 from LDS (see kernels_padic_enc.hpp)
    constexpr int LDS_DIGITS = MODE == PADIC_LDS_M ? 3 : 2;
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * LDS_DIGITS * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = ctx->n[i]; ldsn[NL + i] = P.pm1[which][i]; }
    __syncthreads();
    // the modulus limbs are wave-uniform multiplier operands: pin them in SGPRs (every use is statically
    // indexed, so the array never leaves the register file) instead of letting the compiler hoist LDS
    // loads into 36 VGPRs/AGPRs (per 65 536 decryptions, SGPRs vs LDS: 36 limbs 488 vs 508 ms (x16), 56 limbs 142 vs 151 ms,
    // 72 limbs 364 vs 391 ms)
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* pm1 = ldsn + NL;
    const uint32_t n0inv = ctx->n0inv;
    const uint32_t* __restrict__ kdig = P.kdig[which];
    const uint16_t* __restrict__ ops = P.ops[which];
    const int nops = P.nops[which];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * LDS_DIGITS * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * gridDim.y * BLOCK_THREADS;
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BLOCK_THREADS + threadIdx.x;
    // M: quotient digits of the first half of the product rule: an LDS digit buffer, or (PADIC_WBUF) a strided global column
    const typename E::MBuf M = MODE != PADIC_LDS_M ? typename E::MBuf{P.wscratch + slot, nslots} : typename E::MBuf{B + E::NC * 64, 64};
    auto SQR = [&]() {
        if constexpr (MODE == PADIC_WBUF) {
            if constexpr (NL <= PADIC_SQR_SYM_MAX_NL) E::sqr_sym_fused(A, B, nm, pm1, n0inv);
            else E::sqr_fused(A, B, nm, pm1, n0inv);
        } else E::sqr(A, B, M, nm, pm1, n0inv);
    };
    auto MUL = [&](auto&& csrc, auto&& dsrc) {
        if constexpr (MODE == PADIC_WBUF) E::mul_fused(A, B, csrc, dsrc, nm, pm1, n0inv);
        else E::mul(A, B, M, csrc, dsrc, nm, pm1, n0inv);
    };
    auto SQRN = [&](int nsq) __attribute__((always_inline)) {
#pragma unroll 1
        for (int s = 0; s < nsq; ++s) SQR();
    };
    // table entry e: digit d (0 = first, 1 = second), chunk c
    auto tbl = [&](int e, int d, int c) -> uint4& { return table[(((size_t)e * 2 + d) * E::NC + c) * nslots + slot]; };
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* row = ct + (size_t)es * P.ct_words;
        // ---- digit form of ct:  sum_i  (c_i, 0) * digits(R^(i+2) mod s^2) -----------------------------
        padic_to_digit_form<E>(A, B, M, row, P.ct_words, kdig, P.nd, nm, pm1, n0inv);
        // ---- table of odd powers: T[i] = base^(2i+1); slot `tbl_entries` keeps base^2 -----------------------
        const int NT = P.tbl_entries;
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { tbl(0, 0, c) = E::ld(A, c); tbl(0, 1, c) = E::ld(B, c); }
        auto from_table = [&](int e, int d) {
            return [&, e, d](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c) {
                    const uint4 t = tbl(e, d, E::UC * blk + c);
                    xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                }
            };
        };
        SQRN(1);
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { tbl(NT, 0, c) = E::ld(A, c); tbl(NT, 1, c) = E::ld(B, c); }
        wave_lds_fence();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { E::st(A, c, tbl(0, 0, c)); E::st(B, c, tbl(0, 1, c)); }
        wave_lds_fence();
#pragma unroll 1
        for (int k = 1; k < NT; ++k) {
            MUL(from_table(NT, 0), from_table(NT, 1));
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { tbl(k, 0, c) = E::ld(A, c); tbl(k, 1, c) = E::ld(B, c); }
        }
        // ---- sliding-window schedule (wave-uniform, compiled on the host from s - 1) --------------------
        {
            const int i0 = (int)(ops[0] >> 8);
            wave_lds_fence();
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { E::st(A, c, tbl(i0, 0, c)); E::st(B, c, tbl(i0, 1, c)); }
            wave_lds_fence();
        }
#pragma unroll 1
        for (int k = 1; k < nops; ++k) {
            const int op = (int)ops[k];
            const int nsq = op & 0xFF, idx = op >> 8;
            SQRN(nsq);
            if (idx != 0xFF) {
                MUL(from_table(idx, 0), from_table(idx, 1));
            }
        }
        // ---- leave Montgomery form: multiply by the plain element 1 = (1, 0) ----------------------------
        uint32_t w[NL], v[NL];
        {
            auto one = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = 0;
                if (blk == 0) xv[0] = 1;
            };
            auto zero = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = 0;
            };
            E::mm1_mul(w, M, A, one, nm, n0inv);
            E::mm2_mul(v, M, A, B, zero, one, nm, pm1, n0inv);
        }
        // u = w + v s with w == 1 (mod s), w < 2s: L = v (+1 if w = s + 1), reduced into [0, s)
        {
            uint32_t wc[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) wc[j] = w[j];
            E::cond_sub(wc, nm);
            bool same = true;
#pragma unroll
            for (int j = 0; j < NL; ++j) same = same && (wc[j] == w[j]);
            uint32_t carry = same ? 0u : 1u;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const uint32_t t = v[j] + carry;
                v[j] = t & RMASK;
                carry = t >> RB;
            }
            E::cond_sub(v, nm);
            E::cond_sub(v, nm);
        }
        if (live) {
            uint32_t* orow = u_out + ((size_t)which * n + ei) * P.u_words;
            constexpr int MAXW = (RB * NL + 31) / 32;
#pragma unroll
            for (int k = 0; k < MAXW; ++k) {
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)v[j0] >> s0;
                if (j0 + 1 < NL) t |= (uint64_t)v[j0 + 1] << (RB - s0);
                if (j0 + 2 < NL) t |= (uint64_t)v[j0 + 2] << (2 * RB - s0);
                if (k < P.u_words) orow[k] = (uint32_t)t;
            }
            for (int k = MAXW; k < P.u_words; ++k) orow[k] = 0;
        }
        wave_lds_fence();
    }
}

}  // namespace pai
