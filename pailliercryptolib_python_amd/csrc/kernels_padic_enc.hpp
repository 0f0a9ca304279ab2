// Encryption on the digit engine (mont_padic.hpp) with the public modulus n as the digit base:
// an element of Z/n^2 is a pair (a, b), a + b n == x R (mod n^2), and every reduction is modulo n.
//   k_fb_table_padic   per-key fixed-base table T[j][d] = digits( hs^(d 2^(8 j)) R mod n^2 )
//   k_encrypt_padic    mode 0: ct = 1 + m n                (raw_encrypt: the plain digit pair (1, m) itself)
//                      mode 1: ct = (1 + m n) hs^r          (DJN encrypt: prod_j T[j][r_j], then * (1, m))
//                      mode 2: ct <- ct hs^r               (apply_obfuscator: the ciphertext enters digit form as in stage A)
//   k_pow_padic        base^E mod n^2 for a wave-uniform E (standard-scheme obfuscator r^n), sliding windows
//   k_ctmul_padic      ct^e mod n^2 with per-element (or broadcast) exponents: ciphertext * plaintext
//                      (CipherText::operator*, classes.cpp:324-325), fixed windows over a per-slot table
// Same contracts as k_encrypt (kernels_paillier.hpp); ciphertexts are returned as canonical packed words.
#pragma once
#include "kernels_wide.hpp"
#include "mont_padic.hpp"
#include "kernels_padic.hpp"

namespace pai {

// Per-kernel forms, each measured at 72 limbs (profiles/r02 .. r04):
//  * LDS-qualified digit accesses (mont_padic.hpp: XLDS) only in the r^n kernel (161 vs 167 ms per 65 536 with them; ct * pt
//    92.8 vs 89.4 ms per 2^20 and encryption 70 vs 67 ms per 2^20 WITHOUT them);
//  * the fused product rule (Padic::mul_fused / sqr_fused: both halves in one pass, no quotient digits or parked digit in
//    HBM scratch) everywhere except r^n at 36 limbs (1024-bit keys: 25.1 ms unfused vs 29.0 fused per 65 536);
//  * ct * pt squares with the plain rolled first half (the limb-class symmetric one measured slower there,
//    profiles/r04/ctops_ctmul_sym.jsonl).
constexpr bool PAI_XLDS_POW = true;
constexpr bool pai_fused_pow(int nl) { return nl > 36; }

struct EncPadicParams {
    const MontCtx* nctx;         // modulus n (NL limbs, R = 2^(29 NL))
    const uint32_t* nm1;         // n - 1 limbs
    const uint32_t* nsq;         // n^2 limbs (2 NL, radix 29)
    const uint4* fb_table;       // [J][2^fb_wbits][2][NC] uint4
    uint4* mscratch;             // [2 NC][nslots]: quotient digits, then the parked first result digit
    const uint32_t* kdig;        // mode 2: [nd][2][NL] digit pairs of R^(i+2) mod n^2
    int nd;
    int fb_windows, fb_wbits;
    int pt_words, ct_words, r_words;
    int fb_gform;                // g-factored table (round 4): entry = (a, t) with x R == a (1 + t n) (mod n^2), see k_fb_g_*
};

// the g-factored product runs on the fused product rule of both encryption kernels
constexpr bool PAI_ENC_GFORM_OK = true;

// ---- table construction: one lane per window j ------------------------------------------------------
template <int NL, int U>
__global__ void __launch_bounds__(64, 1)
k_fb_table_padic(const MontCtx* __restrict__ nctx, const uint32_t* nm1, const uint32_t* __restrict__ hs_dig,
                 const uint32_t* __restrict__ one_dig, uint4* __restrict__ table, int J, int wb,
                 const uint32_t* __restrict__ bases_plain, int base_words, const uint32_t* __restrict__ kdig, int nd) {
    using E = Padic<NL, U>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + 3 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += 64) { ldsn[i] = nctx->n[i]; ldsn[NL + i] = nm1[i]; }
    __syncthreads();
    const uint32_t* nm = ldsn;
    nm1 = ldsn + NL;
    const uint32_t n0inv = nctx->n0inv;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane;
    const int js = j < J ? j : J - 1;
    uint4* A = reinterpret_cast<uint4*>(lds) + lane;
    uint4* B = A + E::NC * 64;
    const typename E::MBuf M{B + E::NC * 64, 64};
    const int ENT = 1 << wb;
    auto ent = [&](int d, int dg, int c) -> uint4& { return table[((((size_t)js << wb) + d) * 2 + dg) * E::NC + c]; };
    auto self = [&](const uint4* X) {
        return [=](int blk, uint32_t (&xv)[U]) { E::digits(X, blk, xv); };
    };
    if (bases_plain != nullptr) {
        // the window bases B_j = hs^(2^(wb j)) arrive as plain residues modulo n^2 (k_sq_chain: ONE chain of squarings on
        // an integer-per-wavefront geometry, microseconds per product) and enter digit form the way ciphertexts do
        padic_to_digit_form<E>(A, B, M, bases_plain + (size_t)js * base_words, base_words, kdig, nd, nm, nm1, n0inv);
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { ent(1, 0, c) = E::ld(A, c); ent(1, 1, c) = E::ld(B, c); }
    } else {
    // B_j = hs^(2^(wb j)): every lane walks the same squaring chain and snapshots its own window base
#pragma unroll 1
    for (int c = 0; c < E::NC; ++c) {
        E::st(A, c, make_uint4(hs_dig[4 * c], hs_dig[4 * c + 1], hs_dig[4 * c + 2], hs_dig[4 * c + 3]));
        E::st(B, c, make_uint4(hs_dig[NL + 4 * c], hs_dig[NL + 4 * c + 1], hs_dig[NL + 4 * c + 2], hs_dig[NL + 4 * c + 3]));
    }
    wave_lds_fence();
    const int jmax = min(J - 1, blockIdx.x * 64 + 63);
#pragma unroll 1
    for (int s = 0;; ++s) {
        if (s == wb * js) {
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { ent(1, 0, c) = E::ld(A, c); ent(1, 1, c) = E::ld(B, c); }
        }
        if (s == wb * jmax) break;
        E::mul(A, B, M, self(A), self(B), nm, nm1, n0inv);            // x <- x^2
    }
    }
    __threadfence();
    // T[j][0] = 1 (Montgomery digit form), T[j][d] = T[j][d-1] * B_j
#pragma unroll 1
    for (int c = 0; c < E::NC; ++c) {
        ent(0, 0, c) = make_uint4(one_dig[4 * c], one_dig[4 * c + 1], one_dig[4 * c + 2], one_dig[4 * c + 3]);
        ent(0, 1, c) = make_uint4(one_dig[NL + 4 * c], one_dig[NL + 4 * c + 1], one_dig[NL + 4 * c + 2], one_dig[NL + 4 * c + 3]);
    }
    wave_lds_fence();
#pragma unroll 1
    for (int c = 0; c < E::NC; ++c) { E::st(A, c, ent(1, 0, c)); E::st(B, c, ent(1, 1, c)); }
    wave_lds_fence();
    auto from_ent = [&](int dg) {
        return [&, dg](int blk, uint32_t (&xv)[U]) {
#pragma unroll
            for (int c = 0; c < E::UC; ++c) {
                const uint4 t = ent(1, dg, E::UC * blk + c);
                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
            }
        };
    };
#pragma unroll 1
    for (int d = 2; d < ENT; ++d) {
        E::mul(A, B, M, from_ent(0), from_ent(1), nm, nm1, n0inv);
        if (j < J) {
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { ent(d, 0, c) = E::ld(A, c); ent(d, 1, c) = E::ld(B, c); }
        }
    }
}

// ---- wide windows (2 h bits): T[j][d1 2^h + d0] = S[2 j + 1][d1] * S[2 j][d0] ----------------------------
// S is a table of h-bit windows at twice the density (k_fb_table_padic with 2 J windows of h bits), so that
// S[2 j] covers the low half and S[2 j + 1] the high half of window j.  One lane per output entry: the
// J 2^(2 h) products are independent, which turns the sequential d -> d + 1 chain (2^(2 h) steps per lane)
// into one parallel pass.
template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_fb_expand_padic(const MontCtx* __restrict__ nctx, const uint32_t* nm1g, const uint4* __restrict__ S,
                  uint4* __restrict__ T, int J, int h, uint4* __restrict__ mscratch) {
    using E = Padic<NL, U>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = nctx->n[i]; ldsn[NL + i] = nm1g[i]; }
    __syncthreads();
    const uint32_t* nm = ldsn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{mscratch + slot, nslots};
    const typename E::MBuf Wb{mscratch + (size_t)E::NC * nslots + slot, nslots};
    const size_t total = (size_t)J << (2 * h);
    const size_t tiles = (total + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t idx = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = idx < total;
        const size_t is = live ? idx : total - 1;
        const size_t j = is >> (2 * h);
        const uint32_t d = (uint32_t)(is & (((size_t)1 << (2 * h)) - 1));
        const uint4* lo = S + ((((2 * j) << h) + (d & ((1u << h) - 1u))) * 2) * E::NC;
        const uint4* hi = S + ((((2 * j + 1) << h) + (d >> h)) * 2) * E::NC;
        wave_lds_fence();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { E::st(A, c, lo[c]); E::st(B, c, lo[E::NC + c]); }
        wave_lds_fence();
        auto from_hi = [&](int dg) {
            return [&, dg](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c) {
                    const uint4 t = hi[dg * E::NC + E::UC * blk + c];
                    xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                }
            };
        };
        E::mul_fused(A, B, from_hi(0), from_hi(1), nm, nm1, n0inv);
        if (live) {
            uint4* out = T + is * 2 * E::NC;
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { out[c] = E::ld(A, c); out[E::NC + c] = E::ld(B, c); }
        }
    }
}

// ---- g-factoring of a finished fixed-base table (round 4) ---------------------------------------------------------------
// A table entry is the Montgomery digit pair (a, d) of x = hs^(digit 2^(w j)):  a + d n == x R (mod n^2).  With g = 1 + n,
// a (1 + t n) == a + a t n, so x R == a g^t for t = d a^-1 mod n: every entry factors into an element WITHOUT a second
// digit and a power of g, and g^t g^t' = g^(t + t').  The encryption kernels then multiply by (a, 0) — 4 NL^2 limb
// products instead of 5 — and add the exponents t (kernels above).  The conversion needs a^-1 mod n for every entry:
// Montgomery's simultaneous inversion over chunks of K entries, one chunk per lane (chunk c = entries c, c + nchunks,
// c + 2 nchunks, ...: any grouping serves the trick, and this one makes the lanes of a wave walk consecutive entries) —
//   pass 1 (k_fb_g_prefix): prefix products P_i = a_0 ... a_i R^-i (one Montgomery product per entry, R = 2^(29 NL), modulo n)
//           into `pref`, the chunk total as a packed canonical residue into `tot`;
//   the totals are inverted by the wave-parallel extended GCD (inv_eea.hip: launch_inv_eea);
//   pass 2 (k_fb_g_finish): back sweep with the running inverse (a_0 ... a_i)^-1 R^(i+1): a_i^-1 R = running * P_(i-1) R^-1,
//           t_i = d_i a_i^-1 written over d_i, running <- running * a_i R^-1.
// Four single-digit Montgomery products per entry (1 + 3): the powers of R of the prefix products and of the running
// inverse cancel step by step.  X0 / X1 are the two LDS digit buffers of the lane; the quotient
// digits of mm1_mul are not needed and go to the lane's scratch column.
template <class E>
PAI_DEV void gf_load_digit(uint4* X, const uint4* __restrict__ src) {
    wave_lds_fence();
#pragma unroll 1
    for (int c = 0; c < E::NC; ++c) E::st(X, c, src[c]);
    wave_lds_fence();
}
template <class E, int NL>
PAI_DEV void gf_put_digit(uint4* X, const uint32_t (&w)[NL]) {
    wave_lds_fence();
    E::store_digit(X, w);
    wave_lds_fence();
}

template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_fb_g_prefix(const MontCtx* __restrict__ nctx, const uint4* __restrict__ table, size_t count, int K,
              uint4* __restrict__ pref, uint32_t* __restrict__ tot, int tw, uint4* __restrict__ mscratch) {
    using E = Padic<NL, U>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = nctx->n[i]; ldsn[NL + i] = nctx->r2[i]; }
    __syncthreads();
    const uint32_t* nm = ldsn;
    const uint32_t* r2 = ldsn + NL;
    const uint32_t n0inv = nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* X0 = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{mscratch + slot, nslots};
    auto r2dig = [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(r2, blk, xv); };
    auto one = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = 0;
        if (blk == 0) xv[0] = 1;
    };
    const size_t nchunks = count / (size_t)K;                           // the host makes K divide count
    const size_t tiles = (nchunks + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ch = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ch < nchunks;
        const size_t cs = live ? ch : nchunks - 1;
        uint32_t w[NL];
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const size_t g = (size_t)i * nchunks + cs;      // interleaved chunks: at step i the lanes of a wave touch consecutive entries
            gf_load_digit<E>(X0, table + g * 2 * E::NC);                 // a_i
            if (i == 0) {
#pragma unroll
                for (int c = 0; c < E::NC; ++c) { const uint4 t = E::ld(X0, c); w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w; }
            }
            if (i > 0) {
                const uint4* prev = pref + (g - nchunks) * E::NC;
                auto pdig = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                    for (int c = 0; c < E::UC; ++c) {
                        const uint4 t = prev[E::UC * blk + c];
                        xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                    }
                };
                E::mm1_mul(w, M, X0, pdig, nm, n0inv);                    // P_i = P_(i-1) a_i R^-1 = a_0 ... a_i R^-i
            }
            if (live) {
                uint4* dst = pref + g * E::NC;
#pragma unroll
                for (int c = 0; c < E::NC; ++c) dst[c] = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
            }
            __threadfence_block();
        }
        // the chunk total a_0 ... a_(K-1) R^-(K-1) as a canonical residue, packed words (its inverse carries R^(K-1))
        E::cond_sub(w, nm);
        E::cond_sub(w, nm);
        if (live) {
            uint32_t* orow = tot + ch * (size_t)tw;
            constexpr int MAXW = (RB * NL + 31) / 32;
#pragma unroll
            for (int k = 0; k < MAXW; ++k) {
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)w[j0] >> s0;
                if (j0 + 1 < NL) t |= (uint64_t)w[j0 + 1] << (RB - s0);
                if (j0 + 2 < NL) t |= (uint64_t)w[j0 + 2] << (2 * RB - s0);
                if (k < tw) orow[k] = (uint32_t)t;
            }
            for (int k = MAXW; k < tw; ++k) orow[k] = 0;
        }
    }
}

template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_fb_g_finish(const MontCtx* __restrict__ nctx, uint4* __restrict__ table, size_t count, int K,
              const uint4* __restrict__ pref, const uint32_t* __restrict__ inv, int tw, uint4* __restrict__ mscratch) {
    using E = Padic<NL, U>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = nctx->n[i]; ldsn[NL + i] = nctx->r2[i]; }
    __syncthreads();
    const uint32_t* nm = ldsn;
    const uint32_t* r2 = ldsn + NL;
    const uint32_t n0inv = nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* X0 = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;      // the running inverse (Montgomery form)
    uint4* X1 = X0 + E::NC * 64;                                                       // a_i^-1 R for the t product
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{mscratch + slot, nslots};
    auto r2dig = [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(r2, blk, xv); };
    const size_t nchunks = count / (size_t)K;
    const size_t tiles = (nchunks + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ch = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ch < nchunks;
        const size_t cs = live ? ch : nchunks - 1;
        uint32_t w[NL];
        {   // the inverse of the chunk total is (a_0 ... a_(K-1))^-1 R^(K-1); one more R: X0 = (a_0 ... a_(K-1))^-1 R^K
            const uint32_t* irow = inv + cs * (size_t)tw;
            wave_lds_fence();
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c)
                E::st(X0, c, make_uint4(row_limb(irow, tw, 4 * c), row_limb(irow, tw, 4 * c + 1), row_limb(irow, tw, 4 * c + 2),
                                        row_limb(irow, tw, 4 * c + 3)));
            wave_lds_fence();
            E::mm1_mul(w, M, X0, r2dig, nm, n0inv);
            gf_put_digit<E, NL>(X0, w);
        }
#pragma unroll 1
        for (int i = K - 1; i >= 0; --i) {
            const size_t g = (size_t)i * nchunks + cs;      // interleaved chunks: at step i the lanes of a wave touch consecutive entries
            uint4* ent = table + g * 2 * E::NC;
            auto from = [&](const uint4* p) {
                return [p](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                    for (int c = 0; c < E::UC; ++c) {
                        const uint4 t = p[E::UC * blk + c];
                        xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                    }
                };
            };
            // u = a_i^-1 R: the running inverse (a_0 ... a_i)^-1 R^(i+1) times P_(i-1) = a_0 ... a_(i-1) R^-(i-1), times R^-1
            // (i = 0: the running inverse itself)
            if (i > 0) {
                E::mm1_mul(w, M, X0, from(pref + (g - nchunks) * E::NC), nm, n0inv);
                gf_put_digit<E, NL>(X1, w);
            } else {
                wave_lds_fence();
#pragma unroll 1
                for (int c = 0; c < E::NC; ++c) E::st(X1, c, E::ld(X0, c));
                wave_lds_fence();
            }
            // t_i = d_i a_i^-1 (plain), canonical, over d_i
            E::mm1_mul(w, M, X1, from(ent + E::NC), nm, n0inv);
            E::cond_sub(w, nm);
            E::cond_sub(w, nm);
            if (live) {
#pragma unroll
                for (int c = 0; c < E::NC; ++c) ent[E::NC + c] = make_uint4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
            }
            // running inverse <- running inverse * a_i R^-1: (a_0 ... a_(i-1))^-1 R^i, one power of R less per step, as the
            // prefix products lose one per step
            if (i > 0) {
                E::mm1_mul(w, M, X0, from(ent), nm, n0inv);
                gf_put_digit<E, NL>(X0, w);
            }
        }
    }
}

// ---- (lo in LDS digit A, hi in LDS digit B) as one 2 NL-limb integer: conditional subtraction of n^2 ----
template <class E>
PAI_DEV void cond_sub_2nl(uint4* A, uint4* B, const uint32_t* __restrict__ nsq) {
    int32_t borrow = 0;
#pragma unroll 1
    for (int c = 0; c < 2 * E::NC; ++c) {
        const uint4 t = c < E::NC ? E::ld(A, c) : E::ld(B, c - E::NC);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) borrow = ((int32_t)w[k] - (int32_t)nsq[4 * c + k] + borrow) >> RB;
    }
    const bool ge = borrow == 0;
    int32_t b2 = 0;
#pragma unroll 1
    for (int c = 0; c < 2 * E::NC; ++c) {
        uint4* dst = c < E::NC ? A : B;
        const int cc = c < E::NC ? c : c - E::NC;
        const uint4 t = E::ld(dst, cc);
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int32_t d = (int32_t)w[k] - (int32_t)nsq[4 * c + k] + b2;
            b2 = d >> RB;
            w[k] = ge ? ((uint32_t)d & RMASK) : w[k];
        }
        E::st(dst, cc, make_uint4(w[0], w[1], w[2], w[3]));
    }
    wave_lds_fence();
}

// limb J of the 2 NL-limb integer held in the LDS digit buffers (A = low half, B = high half)
template <class E>
PAI_DEV uint32_t lds_limb(const uint4* A, const uint4* B, int J) {
    if (J >= 2 * E::NC * 4) return 0u;
    const uint4* base = J < 4 * E::NC ? A : B;
    const int jj = J < 4 * E::NC ? J : J - 4 * E::NC;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (jj >> 2) * 64);
    return p[jj & 3];
}

// OBF = true is the apply_obfuscator instantiation (mode 2 only): kept out of the encryption kernel, whose register
// allocation suffers from the extra digit-form conversion (k_encrypt 67 -> 78 ms per 2^20 when both shared one kernel)
template <int NL, int U, bool OBF, bool GFORM>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_encrypt_padic(EncPadicParams P, const uint32_t* __restrict__ m, const uint32_t* __restrict__ r,
                const uint32_t* __restrict__ ct_in, uint32_t* __restrict__ ct_out, int n, int mode) {
    using E = Padic<NL, U, false>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // the modulus and n - 1 are read from LDS (broadcast reads): through the kernel-argument struct the
    // compiler cannot prove them unclobbered and would fetch them with vector global loads inside the loops
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.nctx->n[i]; ldsn[NL + i] = P.nm1[i]; }
    __syncthreads();
    // modulus limbs pinned in SGPRs for the statically indexed uses (see kernels_padic.hpp); the rolled
    // loop of mul_plain indexes them dynamically and keeps reading the LDS copy
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* nm_lds = ldsn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = P.nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{P.mscratch + slot, nslots};
    const typename E::MBuf Wb{P.mscratch + (size_t)E::NC * nslots + slot, nslots};
    auto one = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = 0;
        if (blk == 0) xv[0] = 1;
    };
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t w[NL], v[NL];
        if (mode == 0) {
            // the plain digit pair of 1 + m n is (1, m)
#pragma unroll
            for (int j = 0; j < NL; ++j) { w[j] = 0; v[j] = row_limb(m + (size_t)es * P.pt_words, P.pt_words, j); }
            w[0] = 1;
        } else {
            // x = prod_j T[j][r_j]  (Montgomery digit form of hs^r)
            const uint32_t* rrow = r + (size_t)es * P.r_words;
            // mode 2 (apply_obfuscator): start from the existing ciphertext in digit form instead of the first entry
            if constexpr (OBF) padic_to_digit_form<E>(A, B, M, ct_in + (size_t)es * P.ct_words, P.ct_words, P.kdig, P.nd, nm, nm1, n0inv);
            // g-factored tables: the entries are (a, t) with x = (a, 0) g^t, g = 1 + n: the product takes the 4 NL^2 rule for a
            // right operand without a second digit and the exponents t are summed lazily in registers (limbs < 2^29, at
            // most 7 additions between carry passes); g^(m + sum t) joins in the final product with the plain pair (1, s)
            uint32_t tsum[GFORM ? NL : 1];
            if constexpr (GFORM) {
#pragma unroll
                for (int j = 0; j < NL; ++j) tsum[j] = 0;
            }
            int since_norm = 0;
#pragma unroll 1
            for (int jw = 0; jw < P.fb_windows; ++jw) {
                const int bit = jw * P.fb_wbits, k = bit >> 5;
                uint64_t bits2 = rrow[k];
                if (k + 1 < P.r_words) bits2 |= (uint64_t)rrow[k + 1] << 32;
                const uint32_t d = (uint32_t)(bits2 >> (bit & 31)) & ((1u << P.fb_wbits) - 1u);
                const uint4* ent = P.fb_table + (((size_t)jw << P.fb_wbits) + d) * 2 * E::NC;
                if constexpr (GFORM) {
                    {
#pragma unroll
                        for (int c = 0; c < E::NC; ++c) {
                            const uint4 t = ent[E::NC + c];
                            tsum[4 * c] += t.x; tsum[4 * c + 1] += t.y; tsum[4 * c + 2] += t.z; tsum[4 * c + 3] += t.w;
                        }
                        if (++since_norm == 6) {               // limbs < 7 * 2^29 + carry: no 32-bit wrap
                            since_norm = 0;
                            uint32_t cy = 0;
#pragma unroll
                            for (int j = 0; j < NL; ++j) {
                                const uint32_t tj = tsum[j] + cy;       // (tsum[j] < 7 * 2^29, cy < 8)
                                cy = tj >> RB;
                                tsum[j] = tj & RMASK;
                            }                                           // (the sum stays far below 2^(29 NL): cy == 0 at the top)
                        }
                        if (jw == 0 && !OBF) {
                            wave_lds_fence();
#pragma unroll 1
                            for (int c = 0; c < E::NC; ++c) { E::st(A, c, ent[c]); E::st(B, c, make_uint4(0u, 0u, 0u, 0u)); }
                            wave_lds_fence();
                        } else {
                            auto from_a = [&](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                                for (int c = 0; c < E::UC; ++c) {
                                    const uint4 t = ent[E::UC * blk + c];
                                    xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                                }
                            };
                            E::mul_fused_c0(A, B, from_a, nm, nm1, n0inv);
                        }
                        continue;
                    }
                }
                if (jw == 0 && !OBF) {
                    wave_lds_fence();
#pragma unroll 1
                    for (int c = 0; c < E::NC; ++c) { E::st(A, c, ent[c]); E::st(B, c, ent[E::NC + c]); }
                    wave_lds_fence();
                } else {
                    auto from_ent = [&](int dg) {
                        return [&, dg](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                            for (int c = 0; c < E::UC; ++c) {
                                const uint4 t = ent[dg * E::NC + E::UC * blk + c];
                                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                            }
                        };
                    };
                    E::mul_fused(A, B, from_ent(0), from_ent(1), nm, nm1, n0inv);
                }
            }
            // times the plain digit pair (1, m) of 1 + m n  => plain digit pair of the ciphertext
            {
                // mode 2 leaves Montgomery form with the plain pair (1, 0)
                const uint32_t* mrow = OBF ? nullptr : m + (size_t)es * P.pt_words;
                constexpr bool gf = GFORM;
                if constexpr (GFORM) {
                    {
                        // s = m + sum t (lazy: < 2^7 n, well inside the digit), parked in this lane's Wb column for the row-block reads
                        uint32_t cy = 0;
#pragma unroll
                        for (int c = 0; c < E::NC; ++c) {
                            uint32_t sl[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int j = 4 * c + k;
                                const uint64_t tj = (uint64_t)tsum[j] + (OBF ? 0u : row_limb(mrow, P.pt_words, j)) + cy;
                                cy = (uint32_t)(tj >> RB);
                                sl[k] = (uint32_t)tj & RMASK;
                            }
                            Wb.p[(size_t)c * Wb.stride] = make_uint4(sl[0], sl[1], sl[2], sl[3]);
                        }
                    }
                }
                auto mdig = [&](int blk, uint32_t (&xv)[U]) {
                    if constexpr (gf) {
#pragma unroll
                        for (int c = 0; c < E::UC; ++c) {
                            const uint4 t = Wb.p[(size_t)(E::UC * blk + c) * Wb.stride];
                            xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u) xv[u] = OBF ? 0u : row_limb(mrow, P.pt_words, U * blk + u);
                    }
                };
                E::mm1_mul(w, M, A, one, nm, n0inv);
                E::mm2_mul(v, M, A, B, mdig, one, nm, nm1, n0inv);
            }
        }
        // ct = w + v n  (w, v lazy < 2n + eps)  ->  canonical residue modulo n^2, packed words
        wave_lds_fence();
        E::store_digit(B, v);
        wave_lds_fence();
        uint32_t hi[NL];
        E::mul_plain(hi, A, w, B, [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(nm_lds, blk, xv); });
        wave_lds_fence();
        E::store_digit(B, hi);
        wave_lds_fence();
        if (mode != 0) {
            cond_sub_2nl<E>(A, B, P.nsq);
            cond_sub_2nl<E>(A, B, P.nsq);
        }
        if (live) {
            uint32_t* orow = ct_out + (size_t)ei * P.ct_words;
#pragma unroll 1
            for (int k = 0; k < P.ct_words; ++k) {
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)lds_limb<E>(A, B, j0) >> s0;
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 1) << (RB - s0);
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 2) << (2 * RB - s0);
                orow[k] = (uint32_t)t;
            }
        }
        wave_lds_fence();
    }
}

// ---- ciphertext * plaintext: out_i = ct_i ^ e_i mod n^2 on base-n digit pairs ------------------------------------------
struct CtMulPadicParams {
    const MontCtx* nctx;         // modulus n (NL limbs)
    const uint32_t* nm1;         // n - 1 limbs
    const uint32_t* nsq;         // n^2 limbs (2 NL, radix 29)
    const uint32_t* kdig;        // [nd][2][NL] digit pairs of R^(i+2) mod n^2
    const uint32_t* one_dig;     // [2][NL] digit pair of R mod n^2
    uint4* mscratch;             // [2 NC][nslots]
    uint4* table;                // [2^wbits][2][NC][nslots] per-slot powers x^0 .. x^(2^wbits - 1)
    int nd, wbits;
    int ct_words, e_words, ebits_max, e_bcast;
};

// Fixed windows of `wbits` bits, most significant first; every lane looks its own digit up in its own table column.
// A window in which every lane's digit is zero is skipped (wave-uniform); otherwise lanes with a zero digit
// multiply by the table's entry 0 (= 1).  Squarings run as products (see kernels_padic.hpp on the 72-limb squaring).
template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_ctmul_padic(CtMulPadicParams P, const uint32_t* __restrict__ ct, const uint32_t* __restrict__ e, uint32_t* __restrict__ out, int n) {
    using E = Padic<NL, U, false>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.nctx->n[i]; ldsn[NL + i] = P.nm1[i]; }
    __syncthreads();
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* nm_lds = ldsn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = P.nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{P.mscratch + slot, nslots};
    const typename E::MBuf Wb{P.mscratch + (size_t)E::NC * nslots + slot, nslots};
    auto tbl = [&](int ent, int d, int c) -> uint4& { return P.table[(((size_t)ent * 2 + d) * E::NC + c) * nslots + slot]; };
    auto from_table = [&](int ent, int d) {
        return [&, ent, d](int blk, uint32_t (&xv)[U]) {
#pragma unroll
            for (int c = 0; c < E::UC; ++c) {
                const uint4 t = tbl(ent, d, E::UC * blk + c);
                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
            }
        };
    };
    auto self = [&](const uint4* X) { return [=](int blk, uint32_t (&xv)[U]) { E::digits(X, blk, xv); }; };
    const int W = P.wbits, NT = 1 << W;
    const int nwin = (P.ebits_max + W - 1) / W;
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* erow = e + (size_t)(P.e_bcast ? 0 : es) * P.e_words;
        auto window = [&](int wi) -> uint32_t {
            const int bit = wi * W, k = bit >> 5;
            uint64_t bits2 = k < P.e_words ? erow[k] : 0u;
            if (k + 1 < P.e_words) bits2 |= (uint64_t)erow[k + 1] << 32;
            return (uint32_t)(bits2 >> (bit & 31)) & (uint32_t)(NT - 1);
        };
        // x in Montgomery digit form; table[d] = x^d
        padic_to_digit_form<E>(A, B, M, ct + (size_t)es * P.ct_words, P.ct_words, P.kdig, P.nd, nm, nm1, n0inv);
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) {
            tbl(0, 0, c) = make_uint4(P.one_dig[4 * c], P.one_dig[4 * c + 1], P.one_dig[4 * c + 2], P.one_dig[4 * c + 3]);
            tbl(0, 1, c) = make_uint4(P.one_dig[NL + 4 * c], P.one_dig[NL + 4 * c + 1], P.one_dig[NL + 4 * c + 2], P.one_dig[NL + 4 * c + 3]);
            tbl(1, 0, c) = E::ld(A, c);
            tbl(1, 1, c) = E::ld(B, c);
        }
#pragma unroll 1
        for (int k = 2; k < NT; ++k) {
            E::mul_fused(A, B, from_table(1, 0), from_table(1, 1), nm, nm1, n0inv);
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { tbl(k, 0, c) = E::ld(A, c); tbl(k, 1, c) = E::ld(B, c); }
        }
        // top window
        {
            const int d0 = (int)window(nwin - 1);
            wave_lds_fence();
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { E::st(A, c, tbl(d0, 0, c)); E::st(B, c, tbl(d0, 1, c)); }
            wave_lds_fence();
        }
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
            for (int sq = 0; sq < W; ++sq) {
                E::sqr_fused(A, B, nm, nm1, n0inv);              // 4 NL^2 instead of the product rule's 5
            }
            const int d = (int)window(wi);
            if (__any(d != 0)) E::mul_fused(A, B, from_table(d, 0), from_table(d, 1), nm, nm1, n0inv);
        }
        // leave Montgomery form (times the plain pair (1, 0)), then ct = w + v n as one integer, canonical
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
            E::mm2_mul(v, M, A, B, zero, one, nm, nm1, n0inv);
        }
        wave_lds_fence();
        E::store_digit(B, v);
        wave_lds_fence();
        uint32_t hi[NL];
        E::mul_plain(hi, A, w, B, [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(nm_lds, blk, xv); });
        wave_lds_fence();
        E::store_digit(B, hi);
        wave_lds_fence();
        cond_sub_2nl<E>(A, B, P.nsq);
        cond_sub_2nl<E>(A, B, P.nsq);
        if (live) {
            uint32_t* orow = out + (size_t)ei * P.ct_words;
#pragma unroll 1
            for (int k = 0; k < P.ct_words; ++k) {
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)lds_limb<E>(A, B, j0) >> s0;
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 1) << (RB - s0);
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 2) << (2 * RB - s0);
                orow[k] = (uint32_t)t;
            }
        }
        wave_lds_fence();
    }
}

// ---- out_i = base_i ^ E mod n^2 for a wave-uniform exponent E (the standard scheme's obfuscator r^n) --------------------
struct PowPadicParams {
    const MontCtx* nctx;
    const uint32_t* nm1;
    const uint32_t* nsq;
    const uint32_t* kdig;        // [>= ceil(in bits / (29 NL))][2][NL]
    const uint16_t* ops;         // sliding-window schedule of E (paillier_capi.hip: compile_sliding_schedule)
    int nops, tbl_entries;
    uint4* mscratch;
    uint4* table;                // [tbl_entries + 1][2][NC][nslots]
    int in_words, ct_words;
};

template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_pow_padic(PowPadicParams P, const uint32_t* __restrict__ base, uint32_t* __restrict__ out, int n) {
    using E = Padic<NL, U, PAI_XLDS_POW>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.nctx->n[i]; ldsn[NL + i] = P.nm1[i]; }
    __syncthreads();
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* nm_lds = ldsn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = P.nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{P.mscratch + slot, nslots};
    const typename E::MBuf Wb{P.mscratch + (size_t)E::NC * nslots + slot, nslots};
    auto tbl = [&](int ent, int d, int c) -> uint4& { return P.table[(((size_t)ent * 2 + d) * E::NC + c) * nslots + slot]; };
    auto from_table = [&](int ent, int d) {
        return [&, ent, d](int blk, uint32_t (&xv)[U]) {
#pragma unroll
            for (int c = 0; c < E::UC; ++c) {
                const uint4 t = tbl(ent, d, E::UC * blk + c);
                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
            }
        };
    };
    auto self = [&](const uint4* X) { return [=](int blk, uint32_t (&xv)[U]) { E::digits(X, blk, xv); }; };
    auto SQR = [&]() { E::template sqr_rolled_w<pai_fused_pow(NL)>(A, B, M, Wb, nm, nm1, n0inv); };      // 4 NL^2 instead of the product rule's 5
    const int NT = P.tbl_entries;
    const int nd = (32 * P.in_words + RB * NL - 1) / (RB * NL);
    const int tiles = (n + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        padic_to_digit_form<E>(A, B, M, base + (size_t)es * P.in_words, P.in_words, P.kdig, nd, nm, nm1, n0inv);
        // table of odd powers T[i] = x^(2i+1); slot NT keeps x^2
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { tbl(0, 0, c) = E::ld(A, c); tbl(0, 1, c) = E::ld(B, c); }
        SQR();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { tbl(NT, 0, c) = E::ld(A, c); tbl(NT, 1, c) = E::ld(B, c); }
        wave_lds_fence();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) { E::st(A, c, tbl(0, 0, c)); E::st(B, c, tbl(0, 1, c)); }
        wave_lds_fence();
#pragma unroll 1
        for (int k = 1; k < NT; ++k) {
            E::template mul_w<pai_fused_pow(NL)>(A, B, M, Wb, from_table(NT, 0), from_table(NT, 1), nm, nm1, n0inv);
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { tbl(k, 0, c) = E::ld(A, c); tbl(k, 1, c) = E::ld(B, c); }
        }
        {
            const int i0 = (int)(P.ops[0] >> 8);
            wave_lds_fence();
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) { E::st(A, c, tbl(i0, 0, c)); E::st(B, c, tbl(i0, 1, c)); }
            wave_lds_fence();
        }
#pragma unroll 1
        for (int k = 1; k < P.nops; ++k) {
            const int op = (int)P.ops[k];
            const int nsq = op & 0xFF, idx = op >> 8;
#pragma unroll 1
            for (int s_ = 0; s_ < nsq; ++s_) SQR();
            if (idx != 0xFF) E::template mul_w<pai_fused_pow(NL)>(A, B, M, Wb, from_table(idx, 0), from_table(idx, 1), nm, nm1, n0inv);
        }
        // leave Montgomery form, w + v n as one canonical integer, packed words (as in k_encrypt_padic)
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
            E::mm2_mul(v, M, A, B, zero, one, nm, nm1, n0inv);
        }
        wave_lds_fence();
        E::store_digit(B, v);
        wave_lds_fence();
        uint32_t hi[NL];
        E::mul_plain(hi, A, w, B, [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(nm_lds, blk, xv); });
        wave_lds_fence();
        E::store_digit(B, hi);
        wave_lds_fence();
        cond_sub_2nl<E>(A, B, P.nsq);
        cond_sub_2nl<E>(A, B, P.nsq);
        if (live) {
            uint32_t* orow = out + (size_t)ei * P.ct_words;
#pragma unroll 1
            for (int k = 0; k < P.ct_words; ++k) {
                const int j0 = (32 * k) / RB, s0 = 32 * k - RB * j0;
                uint64_t t = (uint64_t)lds_limb<E>(A, B, j0) >> s0;
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 1) << (RB - s0);
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 2) << (2 * RB - s0);
                orow[k] = (uint32_t)t;
            }
        }
        wave_lds_fence();
    }
}

// ---- multi-exponentiation: out[r][j] = prod_l base[r][l]^(e[r][l][j]) mod n^2 ---------------------------------------------
// The matrix products of the API (PaillierEncryptedNumber.__matmul__ / __rmatmul__ / dot, ipcl_python.py:829-930) are sums
// of ciphertext * plaintext terms, i.e. products of powers that share their bases across the output columns and their
// squarings across the members of a sum (Straus): every base gets a table of its powers 0 .. 2^w - 1 once
// (k_mexp_table_padic; both the ciphertext and, for negative multipliers, its inverse), and a lane computes the partial
// product over a chunk of members for one output element with ONE chain of squarings: per w-bit window, w squarings,
// then one table product per member.  ct * pt on its own spends 52 squarings + ~28 products per term; here a term costs
// ~14-20 table products.  The partial products leave as canonical residues [chunk][r * M + j] for pai_ct_prod.
struct MexpPadicParams {
    const MontCtx* nctx;
    const uint32_t* nm1;
    const uint32_t* nsq;
    const uint32_t* kdig;        // [nd][2][NL]
    const uint32_t* one_dig;     // [2][NL]
    uint4* mscratch;
    uint4* table;                // [R * K][nsigns][2^wbits][2][NC]
    int nd, ct_words;
    int R, K, M, chunk, nsigns;
    int e_words, ebits_max;
    int by_rows;                 // lane order inside a chunk: 0 = (r, j) with j fastest, 1 = (j, r) with r fastest
    int wbits;                   // window width (2 .. 7): tables of 2^wbits powers per base and sign
};

// one lane per (base, sign): powers 0 .. 15 of the ciphertext (sign 0) or of its inverse (sign 1), digit form
template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_mexp_table_padic(MexpPadicParams P, const uint32_t* __restrict__ ct, const uint32_t* __restrict__ ct_inv, int nlanes) {
    using E = Padic<NL, U, false>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.nctx->n[i]; ldsn[NL + i] = P.nm1[i]; }
    __syncthreads();
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = P.nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{P.mscratch + slot, nslots};
    const typename E::MBuf Wb{P.mscratch + (size_t)E::NC * nslots + slot, nslots};
    const int tiles = (nlanes + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int idx = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = idx < nlanes;
        const int is = live ? idx : nlanes - 1;
        const int b = is / P.nsigns, sg = is - b * P.nsigns;
        const uint32_t* row = (sg ? ct_inv : ct) + (size_t)b * P.ct_words;
        const int NT = 1 << P.wbits;
        uint4* ent = P.table + (size_t)is * NT * 2 * E::NC;               // [d][2][NC]
        padic_to_digit_form<E>(A, B, M, row, P.ct_words, P.kdig, P.nd, nm, nm1, n0inv);
        if (live) {
#pragma unroll 1
            for (int c = 0; c < E::NC; ++c) {
                ent[c] = make_uint4(P.one_dig[4 * c], P.one_dig[4 * c + 1], P.one_dig[4 * c + 2], P.one_dig[4 * c + 3]);
                ent[E::NC + c] = make_uint4(P.one_dig[NL + 4 * c], P.one_dig[NL + 4 * c + 1], P.one_dig[NL + 4 * c + 2], P.one_dig[NL + 4 * c + 3]);
                ent[2 * E::NC + c] = E::ld(A, c);
                ent[3 * E::NC + c] = E::ld(B, c);
            }
        }
        __threadfence();
        const uint4* x1 = ent + 2 * E::NC;            // the base itself, read back as the multiplier (dead lanes: the last live entry)
        auto from_x = [&](int dg) {
            return [=](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                for (int c = 0; c < E::UC; ++c) {
                    const uint4 t = x1[dg * E::NC + E::UC * blk + c];
                    xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                }
            };
        };
#pragma unroll 1
        for (int d = 2; d < NT; ++d) {
            E::mul_fused(A, B, from_x(0), from_x(1), nm, nm1, n0inv);
            if (live) {
#pragma unroll 1
                for (int c = 0; c < E::NC; ++c) { ent[(size_t)(2 * d) * E::NC + c] = E::ld(A, c); ent[(size_t)(2 * d + 1) * E::NC + c] = E::ld(B, c); }
            }
        }
        wave_lds_fence();
    }
}

// lane = (chunk c, output g = r * M + j): partial product over members l in [c * chunk, min(K, (c + 1) * chunk))
// e: [R][K][M][e_words] exponents (|mantissa| << alignment shift), sign: [K][M] bytes (1 = use the inverse's table) or NULL
template <int NL, int U>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_mexp_padic(MexpPadicParams P, const uint32_t* __restrict__ e, const uint8_t* __restrict__ sign, uint32_t* __restrict__ out, int nlanes) {
    using E = Padic<NL, U, false>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* ldsn = lds + (BLOCK_THREADS / 64) * 2 * E::DIGIT_WORDS;
    for (int i = threadIdx.x; i < NL; i += BLOCK_THREADS) { ldsn[i] = P.nctx->n[i]; ldsn[NL + i] = P.nm1[i]; }
    __syncthreads();
    uint32_t sn[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) sn[j] = __builtin_amdgcn_readfirstlane(ldsn[j]);
    const uint32_t* nm = sn;
    const uint32_t* nm_lds = ldsn;
    const uint32_t* nm1 = ldsn + NL;
    const uint32_t n0inv = P.nctx->n0inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* A = reinterpret_cast<uint4*>(lds + wave * 2 * E::DIGIT_WORDS) + lane;
    uint4* B = A + E::NC * 64;
    const size_t nslots = (size_t)gridDim.x * BLOCK_THREADS;
    const size_t slot = (size_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    const typename E::MBuf M{P.mscratch + slot, nslots};
    const typename E::MBuf Wb{P.mscratch + (size_t)E::NC * nslots + slot, nslots};
    const int G = P.R * P.M;
    const int W = P.wbits, NT = 1 << W;
    const int nwin = (P.ebits_max + W - 1) / W;
    const int tiles = (nlanes + BLOCK_THREADS - 1) / BLOCK_THREADS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int idx = tile * BLOCK_THREADS + threadIdx.x;
        const bool live = idx < nlanes;
        const int is = live ? idx : nlanes - 1;
        // lanes of a wave walk the output columns of one row (they share their bases' tables), or — by_rows — the rows
        // of one column (they share the multipliers, so their windows line up when the rows' exponents agree)
        const int ch = is / G, g = is - ch * G;
        const int r = P.by_rows ? g % P.R : g / P.M, j = P.by_rows ? g / P.R : g - (g / P.M) * P.M;
        const int l0 = ch * P.chunk, l1 = min(P.K, l0 + P.chunk);
        // the longest chunk of the wave sets the trip count (chunks are equal except the last)
        const int lcount = P.chunk;
        wave_lds_fence();
#pragma unroll 1
        for (int c = 0; c < E::NC; ++c) {
            E::st(A, c, make_uint4(P.one_dig[4 * c], P.one_dig[4 * c + 1], P.one_dig[4 * c + 2], P.one_dig[4 * c + 3]));
            E::st(B, c, make_uint4(P.one_dig[NL + 4 * c], P.one_dig[NL + 4 * c + 1], P.one_dig[NL + 4 * c + 2], P.one_dig[NL + 4 * c + 3]));
        }
        wave_lds_fence();
        bool started = false;                         // wave-uniform: nothing but ones so far, squarings can be skipped
#pragma unroll 1
        for (int wi = nwin - 1; wi >= 0; --wi) {
            if (started) {
#pragma unroll 1
                for (int sq = 0; sq < W; ++sq) E::sqr_fused(A, B, nm, nm1, n0inv);
            }
            const int bit = wi * W, k = bit >> 5, sh = bit & 31;
#pragma unroll 1
            for (int li = 0; li < lcount; ++li) {
                const int l = l0 + li;
                const bool has = live && l < l1;
                const int ls = l < P.K ? l : P.K - 1;
                const size_t eoff = (((size_t)r * P.K + ls) * P.M + j) * P.e_words;
                uint64_t bits2 = k < P.e_words ? e[eoff + k] : 0u;
                if (k + 1 < P.e_words) bits2 |= (uint64_t)e[eoff + k + 1] << 32;
                const int d = has ? (int)((uint32_t)(bits2 >> sh) & (uint32_t)(NT - 1)) : 0;
                if (__any(d != 0)) {
                    const int sg = (sign && P.nsigns > 1) ? (int)sign[(size_t)ls * P.M + j] : 0;
                    const uint4* ent = P.table + ((((size_t)r * P.K + ls) * P.nsigns + sg) * NT + d) * 2 * E::NC;
                    auto from_ent = [&](int dg) {
                        return [=](int blk, uint32_t (&xv)[U]) {
#pragma unroll
                            for (int c = 0; c < E::UC; ++c) {
                                const uint4 t = ent[dg * E::NC + E::UC * blk + c];
                                xv[4 * c] = t.x; xv[4 * c + 1] = t.y; xv[4 * c + 2] = t.z; xv[4 * c + 3] = t.w;
                            }
                        };
                    };
                    E::mul_fused(A, B, from_ent(0), from_ent(1), nm, nm1, n0inv);
                    started = true;
                }
            }
        }
        // leave Montgomery form (times the plain pair (1, 0)), then ct = w + v n as one integer, canonical
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
            E::mm2_mul(v, M, A, B, zero, one, nm, nm1, n0inv);
        }
        wave_lds_fence();
        E::store_digit(B, v);
        wave_lds_fence();
        uint32_t hi[NL];
        E::mul_plain(hi, A, w, B, [&](int blk, uint32_t (&xv)[U]) { E::digits_uniform(nm_lds, blk, xv); });
        wave_lds_fence();
        E::store_digit(B, hi);
        wave_lds_fence();
        cond_sub_2nl<E>(A, B, P.nsq);
        cond_sub_2nl<E>(A, B, P.nsq);
        if (live) {
            uint32_t* orow = out + ((size_t)ch * G + (size_t)r * P.M + j) * P.ct_words;
#pragma unroll 1
            for (int k2 = 0; k2 < P.ct_words; ++k2) {
                const int j0 = (32 * k2) / RB, s0 = 32 * k2 - RB * j0;
                uint64_t t = (uint64_t)lds_limb<E>(A, B, j0) >> s0;
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 1) << (RB - s0);
                t |= (uint64_t)lds_limb<E>(A, B, j0 + 2) << (2 * RB - s0);
                orow[k2] = (uint32_t)t;
            }
        }
        wave_lds_fence();
    }
}

}  // namespace pai
