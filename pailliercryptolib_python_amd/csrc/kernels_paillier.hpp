// Paillier-specific kernels built on the Montgomery row engine.
//
//   k_encrypt     ct = (1 + m n) * hs^r mod n^2   (ipcl::PublicKey::encrypt reached from
//                 bindings/ipcl_bindings_classes.cpp:53-60; raw form ipcl_python.py:103-106), or
//                 ct <- ct * hs^r (apply_obfuscator, classes.cpp:71-83).  hs^r uses a per-key
//                 fixed-base table T[j][d] = hs^(d * 2^(WB j)) (Montgomery form) so the obfuscator
//                 costs ceil(randbits/WB) multiplications and no squarings.
//   k_dec_a       per (element, prime s in {p,q}): u_s = (ct mod s^2)^(s-1) mod s^2
//   k_dec_b       m = CRT( L_p(u_p) hp mod p , L_q(u_q) hq mod q )     (PrivateKey::decrypt reached
//                 from classes.cpp:127-133; maths in SURVEY.md App. D)
//   k_pow2        ct <- ct^(2^delta_i) for delta_i > 0 (exponent alignment, ipcl_python.py:570-741)
#pragma once
#include "kernels_common.hpp"

namespace pai {

constexpr int FB_WBITS = 8;                  // fixed-base window width
constexpr int FB_ENTRIES = 1 << FB_WBITS;

// limbs [limb_base, limb_base + NL) of a packed row (for operands wider than the modulus)
template <class G>
PAI_DEV void load_elem_off(uint32_t (&x)[G::NLL], const uint32_t* __restrict__ row, int W32, int limb_base) {
    const int t = G::gl();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) {
        const int bit = RB * (limb_base + G::NLL * t + j);
        const int k = bit >> 5, s = bit & 31;
        const int k0 = k < W32 ? k : W32 - 1, k1 = k + 1 < W32 ? k + 1 : W32 - 1;
        uint32_t lo = row[k0], hi = row[k1];
        lo = k < W32 ? lo : 0u;
        hi = k + 1 < W32 ? hi : 0u;
        const uint64_t v = ((uint64_t)hi << 32) | lo;
        x[j] = (uint32_t)(v >> s) & RMASK;
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

// x += y limb-wise then canonical 29-bit limbs again (no modular reduction)
template <class G>
PAI_DEV void add_limbs(uint32_t (&x)[G::NLL], const uint32_t (&y)[G::NLL]) {
    using RW = Rows<G::NLL, G::U, G::T>;
    uint64_t acc[RW::NW];
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) acc[j] = (uint64_t)x[j] + y[j];
#pragma unroll
    for (int u = 0; u < G::U; ++u) acc[G::NLL + u] = 0;
    RW::finish(acc, x);
}

// ---------------------------------------------------------------------------------------------
struct EncParams {
    const MontCtx* nsq;          // modulus n^2
    const uint32_t* nR;          // n * R mod n^2, radix-29, NLMAX-padded (for 1 + m n)
    const uint32_t* fb_table;    // [J][256][NL] radix-29 limbs, Montgomery form (DJN) or NULL
    int fb_windows;              // J = ceil(randbits / fb_wbits)
    int fb_wbits;                // window width (table has 2^fb_wbits entries per window)
    int pt_words, ct_words, r_words;
    const MontCtx* fin = nullptr; // k_encrypt_tree on a minus-one context (nsq = context of n^2 k, table and nR in ITS Montgomery form):
};                                // the context of n^2 itself, for the last reduction

// mode 0: ct = 1 + m n                 (raw_encrypt)
// mode 1: ct = (1 + m n) * hs^r        (encrypt, DJN)
// mode 2: ct = ct_in * hs^r            (apply_obfuscator, DJN)
// mode 3: ct = (1 + m n) * obf         (obf = precomputed r^n mod n^2, standard scheme, read from `r`)
// mode 4: ct = ct_in * obf
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_encrypt(EncParams P, const uint32_t* __restrict__ m, const uint32_t* __restrict__ r,
          const uint32_t* __restrict__ ct_in, uint32_t* __restrict__ ct_out, int n, int mode) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, P.nsq, lds);
    // (minus-one contexts, mode 3 on the wide-group geometries — pai_ct_add_plain: P.nsq is the context of n^2 k, P.nR = n R' in it, its
    // scalar the number of row blocks, P.fin the context of n^2 itself for the way out)
    const uint32_t n0inv = G::M1 ? P.nsq->rows / G::U : P.nsq->n0inv;
    const int t = G::gl();
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t c0[G::NLL];
        if (mode == 0 || mode == 1 || mode == 3) {
            // c0 = m * (n R) * R^-1 + 1 = 1 + m n   (m < n so no reduction is involved)
            uint32_t mm[G::NLL], nr[G::NLL];
            load_elem<G>(mm, m + (size_t)es * P.pt_words, P.pt_words);
            load_const_slice<G>(nr, P.nR);
            mm_times<G>(mm, nr, lds, nm, n0inv);
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            add_limbs<G>(mm, one);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) c0[j] = mm[j];
        } else {
            load_elem<G>(c0, ct_in + (size_t)es * P.ct_words, P.ct_words);
        }
        if (mode == 1 || mode == 2) {
            const uint32_t* rrow = r + (size_t)es * P.r_words;
            uint32_t x[G::NLL];
#pragma unroll 1
            for (int jw = 0; jw < P.fb_windows; ++jw) {
                const int bit = jw * P.fb_wbits, k = bit >> 5;
                uint64_t bits2 = rrow[k];
                if (k + 1 < P.r_words) bits2 |= (uint64_t)rrow[k + 1] << 32;
                const uint32_t d = (uint32_t)(bits2 >> (bit & 31)) & ((1u << P.fb_wbits) - 1u);
                const uint32_t* ent = P.fb_table + ((((size_t)jw << P.fb_wbits) + d) * G::NL) + G::NLL * t;
                if (jw == 0) {
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) x[j] = ent[j];
                } else {
                    uint32_t y[G::NLL];
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) y[j] = ent[j];
                    mm_times<G>(x, y, lds, nm, n0inv);
                }
            }
            // (hs^r R) * c0 * R^-1 = hs^r * c0
            mm_times<G>(x, c0, lds, nm, n0inv);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) c0[j] = x[j];
        } else if (mode == 3 || mode == 4) {
            // obf given in plain form: c0 * obf = MM(MM(c0, obf), R^2)
            uint32_t ob[G::NLL], r2[G::NLL];
            load_elem<G>(ob, r + (size_t)es * P.ct_words, P.ct_words);
            mm_times<G>(c0, ob, lds, nm, n0inv);
            load_const_slice<G>(r2, P.nsq->r2);
            mm_times<G>(c0, r2, lds, nm, n0inv);
        }
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(c0, lds, P.fin);
        else cond_sub<G::NLL, G::T>(c0, nm);
        if (live) store_elem<G>(c0, ct_out + (size_t)ei * P.ct_words, P.ct_words, lds);
    }
}

// ---------------------------------------------------------------------------------------------
// DJN encryption / obfuscation for the smallest batches (wide-group geometries, modes 1 and 2 of k_encrypt): the
// fixed-base product  c0 * prod_j T[j][r_j]  is a chain of fb_windows sequential products on one wave in k_encrypt — every one
// of them latency.  Here the four waves of a workgroup share the integers of ONE wave (64 / T of them): wave w multiplies the
// windows j == w (mod 4) (the next entry's three limbs are fetched under the current product), wave 3 starts its chain
// from c0 = 1 + m n (or the ciphertext to obfuscate), and the four partial products meet through LDS in two levels:
// fb_windows / 4 + 2 products on the critical path instead of fb_windows + 2.  Waves 0-2 hold Montgomery forms (x R), wave 3
// a plain value, so the last product leaves the plain ciphertext: same bits as k_encrypt.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_encrypt_tree(EncParams P, const uint32_t* __restrict__ m, const uint32_t* __restrict__ r,
               const uint32_t* __restrict__ ct_in, uint32_t* __restrict__ ct_out, int n, int mode) {
    static_assert(G::T >= 16 && G::T <= 64 && BLOCK_THREADS == 256, "one to four integers per wave, four waves");
    constexpr int EPW = 64 / G::T;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, P.nsq, lds);
    // minus-one contexts (G::M1: 6 instead of 11 us per product at 2048-bit keys — no quotient multiplication, rows = the limbs
    // n^2 k needs instead of the geometry's capacity): the scalar next to the modulus slice is the number of row blocks
    const uint32_t n0inv = G::M1 ? P.nsq->rows / G::U : P.nsq->n0inv;
    auto times_column = [&](uint32_t (&p)[G::NLL], const uint32_t (&a)[G::NLL], int column) {
        if constexpr (G::M1) mont_mul_m1<G::NLL, G::U, G::T>(p, a, lds + column, G::EPB, nm, (int)n0inv);
        else mont_mul<G::NLL, G::U, G::T>(p, a, lds + column, G::EPB, nm, n0inv);
    };
    const int t = G::gl();
    const int wave = (int)threadIdx.x >> 6;
    const int e = G::elem() - wave * EPW;
    const int tiles = (n + EPW - 1) / EPW;
    auto digit = [&](const uint32_t* rrow, int jw) -> uint32_t {
        const int bit = jw * P.fb_wbits, k = bit >> 5;
        uint64_t bits2 = rrow[k];
        if (k + 1 < P.r_words) bits2 |= (uint64_t)rrow[k + 1] << 32;
        return (uint32_t)(bits2 >> (bit & 31)) & ((1u << P.fb_wbits) - 1u);
    };
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * EPW + e;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* rrow = r + (size_t)es * P.r_words;
        uint32_t x[G::NLL];
        bool have = false;
        // first entry of this wave's chain in flight while wave 3 prepares c0
        uint32_t y[G::NLL];
        int jw = wave;
        if (jw < P.fb_windows) {
            const uint32_t* ent = P.fb_table + ((((size_t)jw << P.fb_wbits) + digit(rrow, jw)) * G::NL) + G::NLL * t;
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) y[j] = ent[j];
        }
        if (wave == 3) {
            if (mode == 1) {
                uint32_t nr[G::NLL], one[G::NLL];
                load_elem<G>(x, m + (size_t)es * P.pt_words, P.pt_words);
                load_const_slice<G>(nr, P.nR);
                mm_times<G>(x, nr, lds, nm, n0inv);                  // 1 + m n = m * (n R) * R^-1 + 1
                set_plain_one<G>(one);
                add_limbs<G>(x, one);
            } else {
                load_elem<G>(x, ct_in + (size_t)es * P.ct_words, P.ct_words);
            }
            have = true;
        }
#pragma unroll 1
        for (; jw < P.fb_windows; jw += 4) {
            uint32_t cur[G::NLL];
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) cur[j] = y[j];
            if (jw + 4 < P.fb_windows) {
                const uint32_t* ent = P.fb_table + ((((size_t)(jw + 4) << P.fb_wbits) + digit(rrow, jw + 4)) * G::NL) + G::NLL * t;
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) y[j] = ent[j];
            }
            if (!have) {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = cur[j];
                have = true;
            } else {
                mm_times<G>(x, cur, lds, nm, n0inv);
            }
        }
        if (!have) load_const_slice<G>(x, P.nsq->one);               // fewer windows than waves: the Montgomery form of 1
        // level 1: waves 0 and 2 take in the partial products of waves 1 and 3
        stage_b<G>(x, lds);
        __syncthreads();
        if (wave == 0 || wave == 2) {
            uint32_t p[G::NLL];
            times_column(p, x, G::elem() + EPW);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = p[j];
        }
        __syncthreads();
        if (wave == 2) stage_b<G>(x, lds);
        __syncthreads();
        if (wave == 0) {
            uint32_t p[G::NLL];
            times_column(p, x, G::elem() + 2 * EPW);
            if constexpr (G::M1) m1_reduce_to_true_modulus<G>(p, lds, P.fin);    // a residue modulo n^2 k -> modulo n^2
            else cond_sub<G::NLL, G::T>(p, nm);
            if (live) store_elem<G>(p, ct_out + (size_t)ei * P.ct_words, P.ct_words, lds);
        }
        __syncthreads();
    }
}

// Fixed-base table of a conventional context (entries x R_c mod n^2, raw radix-29 rows of NL limbs) -> the Montgomery form of
// a minus-one context of n^2 k: one product per entry with c == R'^2 / R_c (mod n^2), out[i] = in[i] c / R' == x R' (lazy
// residue modulo n^2 k, raw rows).  k_encrypt_tree then runs its chain at the minus-one contexts' price.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_fb_to_m1(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n,
           const uint32_t* __restrict__ c29) {
    static_assert(G::M1, "minus-one contexts");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t nblk = ctx->rows / G::U;
    const int t = G::gl();
    uint32_t c[G::NLL];
    load_const_slice<G>(c, c29);
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const size_t es = (size_t)(live ? ei : n - 1);
        uint32_t x[G::NLL];
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) x[j] = in[es * G::NL + G::NLL * t + j];
        mm_times<G>(x, c, lds, nm, nblk);
        if (live) {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) out[es * G::NL + G::NLL * t + j] = x[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Last step of the digit-pair obfuscator (kernels_pair.hpp): the plain pair (w, v) of x = hs^r (1 + m n) [or hs^r]
// arrives as two packed rows of `wv_words` words per element; ct = w + v n (mul_ct = 0) or ct_in (w + v n) mod n^2.
// A kernel of its own: as extra modes of k_encrypt it cost that kernel's fixed-base loop 50 % (36x8: 61 -> 94 ms per 65536).
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_pair_finish(EncParams P, const uint32_t* __restrict__ wv, int wv_words, const uint32_t* __restrict__ ct_in,
              uint32_t* __restrict__ ct_out, int n, int mul_ct) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, P.nsq, lds);
    const uint32_t n0inv = P.nsq->n0inv;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* row = wv + (size_t)es * 2 * wv_words;
        // v n as a Montgomery product with n R: v < 2n + eps, so the result is v n mod n^2 up to one n^2
        uint32_t x[G::NLL], t[G::NLL];
        load_elem<G>(x, row + wv_words, wv_words);
        load_const_slice<G>(t, P.nR);
        mm_times<G>(x, t, lds, nm, n0inv);
        load_elem<G>(t, row, wv_words);
        add_limbs<G>(x, t);
        cond_sub<G::NLL, G::T>(x, nm);
        if (mul_ct) {
            // ct_in * x = MM(MM(ct_in, x), R^2)
            load_elem<G>(t, ct_in + (size_t)es * P.ct_words, P.ct_words);
            mm_times<G>(x, t, lds, nm, n0inv);
            load_const_slice<G>(t, P.nsq->r2);
            mm_times<G>(x, t, lds, nm, n0inv);
        }
        cond_sub<G::NLL, G::T>(x, nm);
        if (live) store_elem<G>(x, ct_out + (size_t)ei * P.ct_words, P.ct_words, lds);
    }
}

// ---------------------------------------------------------------------------------------------
// Second level of the fixed-base table build for the lane-group k_encrypt: with half = 2^h entries per half-width
// window, T[j][hi * half + lo] = S[2 j + 1][hi] * S[2 j][lo] (Montgomery form in and out, raw radix-29 rows of NL
// limbs) — one independent product per entry instead of a binary exponentiation per entry (4096-bit keys: 1.35 s of
// k_modexp_var -> one pass of 2.4 M products).  Wave tiles of 64 / T consecutive entries; the right operand streams
// from the LDS operand buffer as in k_modmul.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_fb_expand(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ S, uint32_t* __restrict__ T, int J, int h) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    const int t = G::gl();
    const size_t half = (size_t)1 << h, per_window = half * half, total = (size_t)J * per_window;
    const size_t tiles = (total + G::EPB - 1) / G::EPB;
    const uint32_t* b_lds = lds + G::elem();
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ei = tile * G::EPB + G::elem();
        const bool live = ei < total;
        const size_t es = live ? ei : total - 1;
        const size_t j = es / per_window, d = es - j * per_window, hi = d >> h, lo = d & (half - 1);
        const uint32_t* ap = S + (((2 * j + 1) << h) + hi) * G::NL + G::NLL * t;
        const uint32_t* bp = S + (((2 * j) << h) + lo) * G::NL + G::NLL * t;
        uint32_t x[G::NLL], y[G::NLL];
        if constexpr (G::NLL % 4 == 0) {                 // limb slices move as 16-byte vectors
            const uint4* a4 = reinterpret_cast<const uint4*>(ap);
            const uint4* b4 = reinterpret_cast<const uint4*>(bp);
#pragma unroll
            for (int c = 0; c < G::NLL / 4; ++c) {
                const uint4 va = a4[c], vb = b4[c];
                x[4 * c] = va.x; x[4 * c + 1] = va.y; x[4 * c + 2] = va.z; x[4 * c + 3] = va.w;
                y[4 * c] = vb.x; y[4 * c + 1] = vb.y; y[4 * c + 2] = vb.z; y[4 * c + 3] = vb.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < G::NLL; ++k) { x[k] = ap[k]; y[k] = bp[k]; }
        }
        stage_b<G>(y, lds);
        uint32_t r[G::NLL];
        mont_mul<G::NLL, G::U, G::T>(r, x, b_lds, G::EPB, nm, n0inv);
        cond_sub<G::NLL, G::T>(r, nm);                   // canonical Montgomery representative, as k_modexp_var stores it
        if (live) {
            uint32_t* op = T + ei * G::NL + G::NLL * t;
            if constexpr (G::NLL % 4 == 0) {
                uint4* o4 = reinterpret_cast<uint4*>(op);
#pragma unroll
                for (int c = 0; c < G::NLL / 4; ++c) o4[c] = make_uint4(r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < G::NLL; ++k) op[k] = r[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decrypt stage A.  blockIdx.y selects the prime (0: p, 1: q).  The ciphertext (2 NL limbs of the
// s^2 geometry) is folded into Montgomery form as lo*R + hi*R^2 = MM(lo, R^2) + MM(hi, R^3).
struct DecAParams {
    const MontCtx* sq[2];        // moduli p^2, q^2
    const uint32_t* r3[2];       // R^3 mod s^2, radix-29, NLMAX-padded
    const uint32_t* expo[2];     // s - 1, packed u32 words
    int ewords[2], ebits[2];
    int ct_words, u_words;       // u_words = words of an s^2 residue
    const MontCtx* fin[2];       // minus-one geometries: sq[] are contexts of s^2 * k (k = -s^-2 mod 2^(29 U)); the result
                                 // is reduced modulo s^2 itself with these conventional contexts at the very end
    const uint16_t* ops[2];      // minus-one geometries: sliding-window schedule of s - 1 (squarings | table index << 8,
    int nops[2];                 // index 0xFF = no multiplication), table of the odd powers base^(2i+1), i < tbl_entries;
    int tbl_entries;             // slot tbl_entries keeps base^2.  NULL: fixed W-bit windows
    int rl = 0;                  // minus-one geometries, smallest batches: k_dec_a_rl (squarings and products on separate waves)
};

template <class G, int W>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_dec_a(DecAParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out /*[2][n][u_words]*/, int n,
        uint32_t* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int which = blockIdx.y;
    const MontCtx* ctx = P.sq[which];
    const uint32_t* expo = P.expo[which];
    const int ewords = P.ewords[which], ebits = P.ebits[which];
    const int t = G::gl();
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;      // minus-one contexts: the number of row blocks instead
    const int nrows = G::M1 ? (int)ctx->rows : G::NL;                   // limbs per chunk of the ciphertext (R = 2^(29 nrows))
    const size_t nslots = (size_t)gridDim.x * gridDim.y * G::EPB;
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * G::EPB + G::elem();
    auto tbl = [&](int entry, int j) -> uint32_t& {
        return table[((size_t)entry * G::NL + (G::NLL * t + j)) * nslots + slot];
    };
    const int nwin = (ebits + W - 1) / W;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* row = ct + (size_t)es * P.ct_words;
        uint32_t x[G::NLL];
        {
            uint32_t bR[G::NLL];
            {
                uint32_t hi[G::NLL], c[G::NLL];
                load_elem_off<G>(hi, row, P.ct_words, nrows);
                load_const_slice<G>(c, P.r3[which]);
                mm_times<G>(hi, c, lds, nm, n0inv);                 // hi * R^2
                load_elem_off<G>(bR, row, P.ct_words, 0);
                if constexpr (G::M1) {                              // the low chunk is nrows limbs, not the geometry's NL
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) bR[j] = (G::NLL * t + j < nrows) ? bR[j] : 0u;
                }
                load_const_slice<G>(c, ctx->r2);
                mm_times<G>(bR, c, lds, nm, n0inv);                 // lo * R
                add_limbs<G>(bR, hi);
                if constexpr (!G::M1) {                             // (minus-one contexts have R > 16 M: < 4M is a valid operand)
                    cond_sub<G::NLL, G::T>(bR, nm);                 // < 4M -> < 2M
                    cond_sub<G::NLL, G::T>(bR, nm);
                }
            }
            if constexpr (G::M1) {
                if (P.ops[which] != nullptr) {
                    // small batches: the host-compiled sliding-window schedule of s - 1 over a table of odd powers (as in
                    // k_dec_a_padic): ~1200 sequential products instead of ~1260, and every one of them is latency here
                    const uint16_t* ops = P.ops[which];
                    const int nops = P.nops[which], NT = P.tbl_entries;
                    uint32_t x2[G::NLL];
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) { x[j] = bR[j]; tbl(0, j) = bR[j]; }
                    mm_square<G>(x, lds, nm, n0inv);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) { x2[j] = x[j]; x[j] = bR[j]; }
#pragma unroll 1
                    for (int k = 1; k < NT; ++k) {
                        mm_times<G>(x, x2, lds, nm, n0inv);
#pragma unroll
                        for (int j = 0; j < G::NLL; ++j) tbl(k, j) = x[j];
                    }
                    {
                        const int i0 = (int)(ops[0] >> 8);
#pragma unroll
                        for (int j = 0; j < G::NLL; ++j) x[j] = tbl(i0, j);
                    }
#pragma unroll 1
                    for (int k = 1; k < nops; ++k) {
                        const int op = (int)ops[k];
                        const int nsq = op & 0xFF, idx = op >> 8;
#pragma unroll 1
                        for (int q = 0; q < nsq; ++q) mm_square<G>(x, lds, nm, n0inv);
                        if (idx != 0xFF) {
                            uint32_t y[G::NLL];
#pragma unroll
                            for (int j = 0; j < G::NLL; ++j) y[j] = tbl(idx, j);
                            mm_times<G>(x, y, lds, nm, n0inv);
                        }
                    }
                    uint32_t one[G::NLL];
                    set_plain_one<G>(one);
                    mm_times<G>(x, one, lds, nm, n0inv);
                    m1_reduce_to_true_modulus<G>(x, lds, P.fin[which]);
                    if (live) store_elem<G>(x, u_out + ((size_t)which * n + ei) * P.u_words, P.u_words, lds);
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { x[j] = bR[j]; tbl(1, j) = bR[j]; }
#pragma unroll 1
            for (int k = 2; k < (1 << W); ++k) {
                mm_times<G>(x, bR, lds, nm, n0inv);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) tbl(k, j) = x[j];
            }
        }
        {
            const uint32_t wv = exp_bits(expo, ewords, (nwin - 1) * W, W);
            if (wv == 0) load_const_slice<G>(x, ctx->one);
            else {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = tbl((int)wv, j);
            }
        }
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
            const uint32_t wv = exp_bits(expo, ewords, wi * W, W);
#pragma unroll 1
            for (int s = 0; s < W; ++s) mm_square<G>(x, lds, nm, n0inv);
            if (wv != 0) {
                uint32_t y[G::NLL];
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) y[j] = tbl((int)wv, j);
                mm_times<G>(x, y, lds, nm, n0inv);
            }
        }
        {
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(x, one, lds, nm, n0inv);
            if constexpr (G::M1) {
                // x is the power modulo s^2 k (lazy): reduce it modulo s^2 itself
                m1_reduce_to_true_modulus<G>(x, lds, P.fin[which]);
            } else {
                cond_sub<G::NLL, G::T>(x, nm);
            }
        }
        if (live) store_elem<G>(x, u_out + ((size_t)which * n + ei) * P.u_words, P.u_words, lds);
    }
}

// ---------------------------------------------------------------------------------------------
// Decrypt stage A for the smallest batches (minus-one geometries): RIGHT-TO-LEFT exponentiation with the squarings and the
// products on SEPARATE waves.  Left to right, every product of the window method sits on the critical path between two
// squarings (~1 030 squarings + ~180 products at 1024-bit primes, each ~3 us when an integer is spread over a wavefront).
// Right to left, the chain s_i = base^(2^i) does not depend on the accumulator: wave A (waves 0 and 1 of the workgroup)
// runs the e_bits - 1 squarings back to back, publishing every s_i in a ring of LDS operand buffers — the staging a squaring
// does anyway —, and wave B (waves 2 and 3, on the other SIMDs of the CU) multiplies the s_i at the set bits of s - 1 into
// the accumulator straight from the ring, half a product per squaring on average: the critical path is the squarings plus
// one product.  Hand-over through two LDS words per wave pair (head: slots published by A; tail: first slot B still needs),
// release / acquire at workgroup scope; a wait that does not end traps instead of hanging the device.
constexpr int RL_RING = 16;
PAI_DEV void rl_publish(uint32_t* flag, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// waits until *flag >= need; returns the value seen (the flag is wave-uniform: the loop branches on a scalar)
// s_sleep units (64 cycles) between wave B's polls of `head`: a polling wave costs the squaring wave of its CU scalar issue
// slots and LDS cycles (3.48 vs 3.37 ms with B polling all the time)
constexpr int RL_SLEEP_B = 8;
template <int SLEEP = 1>
PAI_DEV uint32_t rl_wait(uint32_t* flag, uint32_t need) {
    int spins = 0;
    uint32_t v;
    while ((v = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < need) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1 << 22)) __builtin_trap();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return v;
}
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_dec_a_rl(DecAParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out /*[2][n][u_words]*/, int n) {
    static_assert(G::M1 && G::EPB >= 2 && BLOCK_THREADS == 256, "minus-one geometries, two wave pairs per workgroup");
    constexpr int HALF = G::EPB / 2;                                   // integers per workgroup: waves 0, 1 square, waves 2, 3 multiply
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];     // staging buffer, RL_RING operand buffers, flags
    uint32_t* stage = lds;
    uint32_t* ring = lds + G::LDS_WORDS;
    uint32_t* flags = lds + (RL_RING + 1) * G::LDS_WORDS;
    const int which = blockIdx.y;
    const MontCtx* ctx = P.sq[which];
    const uint32_t* expo = P.expo[which];
    const int ebits = P.ebits[which];
    const int t = G::gl();
    const int wave = (int)threadIdx.x >> 6;
    const bool is_a = wave < 2;
    const int col = G::elem() - (is_a ? 0 : HALF);                     // the pair's column in the ring buffers
    uint32_t* head = flags + 2 * (wave & 1);
    uint32_t* tail = head + 1;
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t nblk = ctx->rows / G::U;
    const int nrows = (int)ctx->rows;
    auto bit_of = [&](int i) -> uint32_t { return (expo[i >> 5] >> (i & 31)) & 1u; };
    auto next_set = [&](int i) -> int {                                // first set bit at or above i (ebits if none)
        while (i < ebits && !bit_of(i)) ++i;
        return i;
    };
    const int tiles = (n + HALF - 1) / HALF;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        if (threadIdx.x < 4) flags[threadIdx.x] = 0;
        __syncthreads();
        const int ei = tile * HALF + col;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        if (is_a) {
            const uint32_t* row = ct + (size_t)es * P.ct_words;
            uint32_t x[G::NLL];
            {
                uint32_t hi[G::NLL], c[G::NLL];
                load_elem_off<G>(hi, row, P.ct_words, nrows);
                load_const_slice<G>(c, P.r3[which]);
                mm_times<G>(hi, c, stage, nm, nblk);                   // hi * R^2
                load_elem_off<G>(x, row, P.ct_words, 0);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) x[j] = (G::NLL * t + j < nrows) ? x[j] : 0u;
                load_const_slice<G>(c, ctx->r2);
                mm_times<G>(x, c, stage, nm, nblk);                    // lo * R
                add_limbs<G>(x, hi);
            }
            uint32_t tail_seen = 0;                                     // B only moves it forward: re-read when the cached value is too old
#pragma unroll 1
            for (int i = 0; i < ebits; ++i) {
                if (i >= RL_RING && tail_seen < (uint32_t)(i - RL_RING + 1)) tail_seen = rl_wait(tail, (uint32_t)(i - RL_RING + 1));
                uint32_t* slot = ring + (i % RL_RING) * G::LDS_WORDS;
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) slot[(G::NLL * t + j) * G::EPB + col] = x[j];
                rl_publish(head, (uint32_t)(i + 1));
                if (i + 1 < ebits) {
                    uint32_t r[G::NLL];
                    mont_mul_m1<G::NLL, G::U, G::T>(r, x, slot + col, G::EPB, nm, (int)nblk);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
                }
            }
        } else {
            uint32_t acc[G::NLL];
            int i = next_set(0);
            rl_publish(tail, (uint32_t)i);
            bool first = true;
#pragma unroll 1
            while (i < ebits) {
                rl_wait<RL_SLEEP_B>(head, (uint32_t)(i + 1));
                const uint32_t* slot = ring + (i % RL_RING) * G::LDS_WORDS;
                if (first) {
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) acc[j] = slot[(G::NLL * t + j) * G::EPB + col];
                    first = false;
                } else {
                    uint32_t r[G::NLL];
                    mont_mul_m1<G::NLL, G::U, G::T>(r, acc, slot + col, G::EPB, nm, (int)nblk);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) acc[j] = r[j];
                }
                i = next_set(i + 1);
                rl_publish(tail, (uint32_t)i);
            }
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(acc, one, stage, nm, nblk);
            m1_reduce_to_true_modulus<G>(acc, stage, P.fin[which]);
            if (live) store_elem<G>(acc, u_out + ((size_t)which * n + ei) * P.u_words, P.u_words, stage);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// base_i ^ e_i mod M for the smallest batches, right to left on wave pairs as k_dec_a_rl (small-batch ct * pt): wave A
// squares ebits_max - 1 times, wave B multiplies base^(2^i) into the accumulators whose exponent has bit i (one product
// for the whole wave wherever some integer of it needs one, kept per integer), starting from the Montgomery form of 1.
// No table of powers: ~ebits_max + 8 sequential products instead of ~1.5 ebits_max + 2^w.  ctx: minus-one context of M k,
// fin: M's own context (as k_modexp_var_win on these geometries).
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_modexp_rl(const MontCtx* __restrict__ ctx, const uint32_t* __restrict__ base, int base_w32, const uint32_t* __restrict__ expo,
            int ew, int ebits_max, int exp_bcast, uint32_t* __restrict__ out, int out_w32, int n, const MontCtx* __restrict__ fin) {
    static_assert(G::M1 && G::EPB >= 2 && BLOCK_THREADS == 256, "minus-one geometries, two wave pairs per workgroup");
    constexpr int HALF = G::EPB / 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stage = lds;
    uint32_t* ring = lds + G::LDS_WORDS;
    uint32_t* flags = lds + (RL_RING + 1) * G::LDS_WORDS;
    const int t = G::gl();
    const int wave = (int)threadIdx.x >> 6;
    const bool is_a = wave < 2;
    const int col = G::elem() - (is_a ? 0 : HALF);
    uint32_t* head = flags + 2 * (wave & 1);
    uint32_t* tail = head + 1;
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t nblk = ctx->rows / G::U;
    const int tiles = (n + HALF - 1) / HALF;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        if (threadIdx.x < 4) flags[threadIdx.x] = 0;
        __syncthreads();
        const int ei = tile * HALF + col;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        if (is_a) {
            uint32_t x[G::NLL], c[G::NLL];
            load_elem<G>(x, base + (size_t)es * base_w32, base_w32);
            load_const_slice<G>(c, ctx->r2);
            mm_times<G>(x, c, stage, nm, nblk);
            uint32_t tail_seen = 0;
#pragma unroll 1
            for (int i = 0; i < ebits_max; ++i) {
                if (i >= RL_RING && tail_seen < (uint32_t)(i - RL_RING + 1)) tail_seen = rl_wait(tail, (uint32_t)(i - RL_RING + 1));
                uint32_t* slot = ring + (i % RL_RING) * G::LDS_WORDS;
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) slot[(G::NLL * t + j) * G::EPB + col] = x[j];
                rl_publish(head, (uint32_t)(i + 1));
                if (i + 1 < ebits_max) {
                    uint32_t r[G::NLL];
                    mont_mul_m1<G::NLL, G::U, G::T>(r, x, slot + col, G::EPB, nm, (int)nblk);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
                }
            }
        } else {
            const uint32_t* erow = expo + (size_t)(exp_bcast ? 0 : es) * ew;
            auto my_bit = [&](int i) -> bool { return (i >> 5) < ew && ((erow[i >> 5] >> (i & 31)) & 1u) != 0u; };
            auto next_any = [&](int i) -> int {                    // first bit position at or above i some integer of this wave has set
                while (i < ebits_max && !__any(my_bit(i) ? 1 : 0)) ++i;
                return i;
            };
            uint32_t acc[G::NLL];
            load_const_slice<G>(acc, ctx->one);
            int i = next_any(0);
            rl_publish(tail, (uint32_t)i);
#pragma unroll 1
            while (i < ebits_max) {
                rl_wait<RL_SLEEP_B>(head, (uint32_t)(i + 1));
                const uint32_t* slot = ring + (i % RL_RING) * G::LDS_WORDS;
                uint32_t r[G::NLL];
                mont_mul_m1<G::NLL, G::U, G::T>(r, acc, slot + col, G::EPB, nm, (int)nblk);
                const bool mine = my_bit(i);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) acc[j] = mine ? r[j] : acc[j];
                i = next_any(i + 1);
                rl_publish(tail, (uint32_t)i);
            }
            uint32_t one[G::NLL];
            set_plain_one<G>(one);
            mm_times<G>(acc, one, stage, nm, nblk);
            m1_reduce_to_true_modulus<G>(acc, stage, fin);
            if (live) store_elem<G>(acc, out + (size_t)ei * out_w32, out_w32, stage);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Decrypt stage B in the geometry of the primes themselves.
struct DecBParams {
    const MontCtx* pr[2];        // moduli p, q
    const uint32_t* sinv2[2];    // s^-1 mod 2^(29 NL)
    const uint32_t* nsinv2[2];   // 2^(29 NL) - s^-1   (so that (u-1) s^-1 = u s^-1 + nsinv2 mod 2^(29 NL))
    const uint32_t* hR[2];       // hp R mod p, hq R mod q
    const uint32_t* pinvqR;      // (p^-1 mod q) R mod q
    int u_words, pt_words;
    int u_is_L;                  // stage A already produced L_s(u_s) (p-adic engine) instead of u_s
};

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAI_LG_WAVES(G))
k_dec_b(DecBParams P, const uint32_t* __restrict__ u_in /*[2][n][u_words]*/, uint32_t* __restrict__ m_out, int n) {
    static_assert(!G::NMLDS, "stage B keeps the (small) prime moduli in registers");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];       // 2 operand buffers
    uint32_t* ldsA = lds;
    uint32_t* ldsB = lds + G::LDS_WORDS;
    const int t = G::gl(), e = G::elem();
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + e;
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        uint32_t ms[2][G::NLL];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            NmRegs<G::NLL> nm;
            load_const_slice<G>(nm.v, P.pr[which]->n);
            const uint32_t n0inv = P.pr[which]->n0inv;
            // L = (u - 1) / s = low half of (u_low * sinv2 + nsinv2)
            uint32_t ulo[G::NLL], c[G::NLL], init[G::NLL], hi[G::NLL];
            load_elem_off<G>(ulo, u_in + ((size_t)which * n + es) * P.u_words, P.u_words, 0);
            if (P.u_is_L) {
                stage_b<G>(ulo, ldsB);                                   // the input already is L
            } else {
                stage_b<G>(ulo, ldsA);
                load_const_slice<G>(c, P.sinv2[which]);
                load_const_slice<G>(init, P.nsinv2[which]);
                mul_plain<G::NLL, G::U, G::T>(hi, init, c, ldsA + e, G::EPB, ldsB + e, G::EPB);
                wave_lds_fence();
            }
            // ms = L * h mod s  with L read back from LDS as the multiplier
            load_const_slice<G>(c, P.hR[which]);
            mont_mul<G::NLL, G::U, G::T>(ms[which], c, ldsB + e, G::EPB, nm, n0inv);
            cond_sub<G::NLL, G::T>(ms[which], nm);
            wave_lds_fence();
        }
        // t = (mq - mp + q) * pinvq mod q ;  m = mp + p * t
        NmRegs<G::NLL> nq;
        load_const_slice<G>(nq.v, P.pr[1]->n);
        uint32_t d[G::NLL];
        {
            using RW = Rows<G::NLL, G::U, G::T>;
            int64_t sd[G::NLL];
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) sd[j] = (int64_t)ms[1][j] - (int64_t)ms[0][j] + (int64_t)nq.v[j];
            RW::finish_signed(sd, d);
        }
        uint32_t c[G::NLL];
        load_const_slice<G>(c, P.pinvqR);
        mm_times<G>(d, c, ldsA, nq, P.pr[1]->n0inv);
        cond_sub<G::NLL, G::T>(d, nq);
        stage_b<G>(d, ldsA);
        uint32_t pl[G::NLL], hi[G::NLL];
        load_const_slice<G>(pl, P.pr[0]->n);
        mul_plain<G::NLL, G::U, G::T>(hi, ms[0], pl, ldsA + e, G::EPB, ldsB + e, G::EPB);
        wave_lds_fence();
        // assemble m = lo (LDS B, NL limbs) + hi (registers) * 2^(29 NL) into packed words
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) ldsA[(G::NLL * t + j) * G::EPB + e] = hi[j];
        wave_lds_fence();
        if (live) {
            uint32_t* row = m_out + (size_t)ei * P.pt_words;
            auto limb = [&](int J) -> uint64_t {
                if (J < G::NL) return ldsB[J * G::EPB + e];
                if (J < 2 * G::NL) return ldsA[(J - G::NL) * G::EPB + e];
                return 0;
            };
            for (int k = t; k < P.pt_words; k += G::T) {
                const int j0 = (32 * k) / RB;
                const int s0 = 32 * k - RB * j0;
                uint64_t v = limb(j0) >> s0;
                v |= limb(j0 + 1) << (RB - s0);
                v |= limb(j0 + 2) << (2 * RB - s0);
                row[k] = (uint32_t)v;
            }
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// ct_i <- ct_i^(2^delta_i) mod n^2 for delta_i > 0; other elements are left untouched.  Wave tiles as in k_modmul
// (load_tile / store_tile, no workgroup barrier); a wave tile without any positive delta is skipped.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_pow2(const MontCtx* __restrict__ ctx, uint32_t* ct, const int32_t* __restrict__ delta, int delta_bcast,
       int n, int w32, const MontCtx* __restrict__ fin) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using WT = WaveTile<G>;
    uint32_t* stage = lds + G::LDS_WORDS + G::NL;
    uint32_t* r2_lds = stage + G::STAGE_WORDS;           // R^2 mod M, one copy per workgroup
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) r2_lds[i] = ctx->r2[i];
    __syncthreads();
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;      // minus-one contexts (small batches, as k_add_aligned): row blocks
    constexpr int WPB = BLOCK_THREADS / 64;
    const int wtiles = (n + WT::EPW - 1) / WT::EPW;
    clear_stage<G>(stage);
    const int per_wave = (wtiles + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    const int wt_begin = ((int)blockIdx.x * WPB + WT::wave()) * per_wave;
    const int wt_end = min(wtiles, wt_begin + per_wave);
    for (int wt = wt_begin; wt < wt_end; ++wt) {
        const int row0 = wt * WT::EPW;
        const int rows = min(WT::EPW, n - row0);
        const int ei = row0 + (WT::lane() / G::T);
        const bool live = ei < n;
        int dl = live ? delta[delta_bcast ? 0 : ei] : 0;
        if (dl < 0) dl = 0;
        int dmax = dl;
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(dmax, off, 64); dmax = o > dmax ? o : dmax; }
        if (dmax == 0) continue;                                     // wave-uniform
        __builtin_amdgcn_s_setprio(2);
        load_tile<G>(stage, ct + (size_t)row0 * w32, rows, w32);
        uint32_t x[G::NLL];
        unpack_row<G>(x, stage);
        __builtin_amdgcn_s_setprio(0);
        // one rolled loop body for every product of the tile: step -1 enters the Montgomery domain (* R^2), steps
        // 0 .. dmax-1 square (kept only by the elements that still need it), step dmax leaves the domain (* 1)
#pragma unroll 1
        for (int s = -1; s <= dmax; ++s) {
            uint32_t y[G::NLL], z[G::NLL];
            const int gl = G::gl();
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) {
                const uint32_t c = s < 0 ? r2_lds[G::NLL * gl + j] : ((gl == 0 && j == 0) ? 1u : 0u);
                y[j] = (s >= 0 && s < dmax) ? x[j] : c;
                z[j] = x[j];
            }
            mm_times<G>(z, y, lds, nm, n0inv);
            const bool keep = s < 0 || s >= dmax || s < dl;
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = keep ? z[j] : x[j];
        }
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(x, lds, fin);
        else cond_sub<G::NLL, G::T>(x, nm);
        __builtin_amdgcn_s_setprio(2);
        pack_row<G>(x, stage);                                       // rows with delta <= 0 come back unchanged
        store_tile<G>(stage, ct + (size_t)row0 * w32, rows, w32);
        __builtin_amdgcn_s_setprio(0);
    }
}

// ---------------------------------------------------------------------------------------------
// out[j] = base^(2^(h j)) mod M for j < nsnap, as plain packed rows: ONE chain of squarings — the window bases of a
// fixed-base table.  Meant for the integer-per-wavefront geometries (a product takes microseconds there; the digit
// engine's table kernel used to walk this chain on one lane per window, 50 us per squaring: 51 ms of every first
// obfuscating call at 2048-bit keys).  Every element of the workgroup computes the same chain; element 0 stores.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_sq_chain(const MontCtx* __restrict__ ctx, const MontCtx* __restrict__ fin, const uint32_t* __restrict__ base, int w32,
           uint32_t* __restrict__ out, int h, int nsnap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;
    uint32_t x[G::NLL], c[G::NLL];
    load_elem<G>(x, base, w32);
    load_const_slice<G>(c, ctx->r2);
    mm_times<G>(x, c, lds, nm, n0inv);                               // base R
#pragma unroll 1
    for (int j = 0; j < nsnap; ++j) {
        uint32_t y[G::NLL], one[G::NLL];
#pragma unroll
        for (int k = 0; k < G::NLL; ++k) y[k] = x[k];
        set_plain_one<G>(one);
        mm_times<G>(y, one, lds, nm, n0inv);                         // leave Montgomery form
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(y, lds, fin);
        else cond_sub<G::NLL, G::T>(y, nm);
        if (G::elem() == 0) store_elem<G>(y, out + (size_t)j * w32, w32, lds);
        if (j + 1 < nsnap) {
#pragma unroll 1
            for (int s = 0; s < h; ++s) mm_square<G>(x, lds, nm, n0inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exponent alignment fused with the addition: PaillierEncryptedNumber.__raw_add (ipcl_python.py:490-526) raises the
// operand with the LOWER fixed-point exponent by ct^(2^delta) (:570-741) and then multiplies the two ciphertexts.
//   delta_i = exponent(a_i) - exponent(b_i);   out_i = delta_i > 0 ? a_i * b_i^(2^delta_i) : a_i^(2^-delta_i) * b_i   (mod n^2)
// One pass over the data and dmax + 2 Montgomery products per wave tile (domain entry of the operand to be raised,
// dmax squarings kept only by the elements that still need them, the product with the other operand, which stays
// plain and thereby leaves the domain) instead of two k_pow2 passes and a k_modmul: 2 dmax + 6.  The bits are the
// same: the same integers are multiplied.  b_bcast: one ciphertext b for every i.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_add_aligned(const MontCtx* __restrict__ ctx, const uint32_t* a, const uint32_t* b, int b_bcast,
              const int32_t* __restrict__ delta, uint32_t* out, int n, int w32, const uint32_t* __restrict__ entry,
              const MontCtx* __restrict__ fin) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using WT = WaveTile<G>;
    uint32_t* stage = lds + G::LDS_WORDS + G::NL;
    uint32_t* r2_lds = stage + G::STAGE_WORDS;           // the domain-entry constant (R^2 mod M), one copy per workgroup
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    // minus-one contexts (small batches on one integer per wavefront, round 6): ctx is the context of n^2 k, its scalar the number
    // of row blocks, and `fin` the context of n^2 itself for the way out (m1_reduce_to_true_modulus) — a product without the
    // dependent quotient digit per row: 51 -> 33 us for the reference's BM_Add_CTCT shifts (profiles/r06/lat_add_m1.jsonl)
    const uint32_t n0inv = G::M1 ? ctx->rows / G::U : ctx->n0inv;
    constexpr int WPB = BLOCK_THREADS / 64;
    const int wtiles = (n + WT::EPW - 1) / WT::EPW;
    clear_stage<G>(stage);
    if (entry == nullptr) {
        for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) r2_lds[i] = ctx->r2[i];
    } else {
        // operands stored as x R^k (pai_ct_add_aligned_dom): the caller's constant R^(2-k), one packed row, replaces R^2
        uint32_t cc[G::NLL];
        load_tile<G>(stage, entry, 1, w32, true);
        unpack_row<G>(cc, stage);
        if (threadIdx.x < G::T) {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) r2_lds[G::NLL * G::gl() + j] = cc[j];
        }
    }
    __syncthreads();
    const uint32_t* o_lds = lds + G::elem();             // column of this element in the [limb][element] operand buffer (staged x)
    const int per_wave = (wtiles + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    const int wt_begin = ((int)blockIdx.x * WPB + WT::wave()) * per_wave;
    const int wt_end = min(wtiles, wt_begin + per_wave);
    for (int wt = wt_begin; wt < wt_end; ++wt) {
        const int row0 = wt * WT::EPW;
        const int rows = min(WT::EPW, n - row0);
        const int ei = row0 + (WT::lane() / G::T);
        const int dl = ei < n ? delta[ei] : 0;
        const bool raise_b = dl > 0;                     // a has the larger exponent: b is the one to be raised
        const int cnt = dl > 0 ? dl : -dl;
        int dmax = cnt;
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(dmax, off, 64); dmax = o > dmax ? o : dmax; }
        __builtin_amdgcn_s_setprio(2);
        uint32_t x[G::NLL], o[G::NLL];
        {
            uint32_t xa[G::NLL], xb[G::NLL];
            load_tile<G>(stage, a + (size_t)row0 * w32, rows, w32);
            unpack_row<G>(xa, stage);
            load_tile<G>(stage, b + (size_t)(b_bcast ? 0 : row0) * w32, b_bcast ? 1 : rows, w32, b_bcast != 0);
            unpack_row<G>(xb, stage);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { x[j] = raise_b ? xb[j] : xa[j]; o[j] = raise_b ? xa[j] : xb[j]; }
        }
        __builtin_amdgcn_s_setprio(0);
        // one rolled loop body for every product of the tile: step -1 enters the Montgomery domain (x * R^2), steps
        // 0 .. dmax-1 square x (kept only by the elements that still need it), step dmax multiplies the plain other
        // operand o by x (Montgomery form), which leaves the domain: o * x R * R^-1 = o * x
#pragma unroll 1
        for (int s = -1; s <= dmax; ++s) {
            uint32_t lhs[G::NLL], r[G::NLL];
            const bool last = s == dmax;
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) lhs[j] = last ? o[j] : x[j];
            if (s >= 0) stage_b<G>(x, lds);              // the right operand of squarings and of the last product: x itself
            if constexpr (G::M1) mont_mul_m1<G::NLL, G::U, G::T>(r, lhs, s < 0 ? r2_lds : o_lds, s < 0 ? 1 : G::EPB, nm, (int)n0inv);
            else mont_mul<G::NLL, G::U, G::T>(r, lhs, s < 0 ? r2_lds : o_lds, s < 0 ? 1 : G::EPB, nm, n0inv);
            const bool keep = s < 0 || last || s < cnt;
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) x[j] = keep ? r[j] : x[j];
        }
        if constexpr (G::M1) m1_reduce_to_true_modulus<G>(x, lds, fin);      // a residue modulo n^2 k -> the canonical one modulo n^2
        else cond_sub<G::NLL, G::T>(x, nm);
        __builtin_amdgcn_s_setprio(2);
        pack_row<G>(x, stage);
        store_tile<G>(stage, out + (size_t)row0 * w32, rows, w32);
        __builtin_amdgcn_s_setprio(0);
    }
}

// ---------------------------------------------------------------------------------------------
// n-ary ciphertext sum (round 5): out_i = prod_{j < k} op_j[i]^(2^raise_j[i]) mod n^2 in ONE pass — the aggregation
// sum_j E(x_j) over k parties' arrays, which the reference spells as a chain of k - 1 __add__ calls with their exponent
// alignments (ipcl_python.py:365-381, 490-526, 570-741; tests/ipcl_python_test.py:21-38).  A wave tile loads the k operand
// rows in turn and keeps the running product in registers: k - 1 Montgomery products per element, one tile store, no
// intermediate arrays — against k - 1 launches of k_modmul, each with two tile loads, a store and its pack / unpack.
//
// Montgomery bookkeeping: operand 0 holds x R^tag0, the others x R^tag; a product of the accumulator (x R^c) with an operand
// (y R^t) is (x y) R^(c + t - 1).  c is wave-uniform and tracked per tile.  An operand that some element of the tile has to
// raise (raise_j > 0: the target exponent of the sum is the per-element maximum, computed by the caller) enters the domain
// first (one product with R^(2 - t), `conv`), is squared max(raise) times (kept only by the elements that still need it)
// and then joins the product as y R, so such a tile ends at a different c than the plain path; the last step brings every
// tile to the caller's dom_out through one product with R^(1 + dom_out - c) from the key's table of powers of R (skipped when
// c already is dom_out — the common case when dom_out is the natural tag tag0 + (k - 1)(tag - 1) and nothing is raised).
// All products of a tile run through ONE rolled loop body (a second inlined copy of the row engine spills: k_modmul).
constexpr int ADDN_MAX = 16;                 // operands per launch
constexpr int RPOW_SPAN = 48;                // the key's table holds R^m mod n^2 for |m| <= RPOW_SPAN (limb form, NL limbs per row)
struct AddnArgs {
    const uint32_t* op[ADDN_MAX];
    const int32_t* raise[ADDN_MAX];          // per operand: int32 [n] of squarings per element (>= 0), or NULL
    int k, tag0, tag, dom_out;
};

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 2)
k_addn(const MontCtx* __restrict__ ctx, AddnArgs A, uint32_t* out, int n, int w32, const uint32_t* __restrict__ rpow) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    using WT = WaveTile<G>;
    uint32_t* stage = lds + G::LDS_WORDS + G::NL;
    uint32_t* conv_lds = stage + G::STAGE_WORDS;         // R^(2 - tag): the domain entry of an operand that is to be raised
    typename G::NM nm;
    load_modulus<G>(nm, ctx, lds);
    const uint32_t n0inv = ctx->n0inv;
    constexpr int WPB = BLOCK_THREADS / 64;
    const int wtiles = (n + WT::EPW - 1) / WT::EPW;
    clear_stage<G>(stage);
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) conv_lds[i] = rpow[(size_t)(RPOW_SPAN + 2 - A.tag) * G::NL + i];
    __syncthreads();
    const uint32_t* o_lds = lds + G::elem();             // column of this element in the [limb][element] operand buffer
    const int per_wave = (wtiles + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    const int wt_begin = ((int)blockIdx.x * WPB + WT::wave()) * per_wave;
    const int wt_end = min(wtiles, wt_begin + per_wave);
    for (int wt = wt_begin; wt < wt_end; ++wt) {
        const int row0 = wt * WT::EPW;
        const int rows = min(WT::EPW, n - row0);
        const int ei = row0 + (WT::lane() / G::T);
        uint32_t acc[G::NLL];
        int c = 0;                                       // the accumulator holds (product so far) R^c
        // operands 0 .. k-1, then pseudo-operand k: the constant that brings the tile to dom_out
#pragma unroll 1
        for (int j = 0; j <= A.k; ++j) {
            uint32_t y[G::NLL];
            int rj = 0, dmax = 0;
            if (j < A.k) {
                const int32_t* rz = A.raise[j];
                if (rz != nullptr) {
                    rj = ei < n ? rz[ei] : 0;
                    rj = rj > 0 ? rj : 0;
                    dmax = rj;
                    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(dmax, off, 64); dmax = o > dmax ? o : dmax; }
                }
                __builtin_amdgcn_s_setprio(2);
                load_tile<G>(stage, A.op[j] + (size_t)row0 * w32, rows, w32);
                unpack_row<G>(y, stage);
                __builtin_amdgcn_s_setprio(0);
            } else {
                if (c == A.dom_out) break;               // wave-uniform
                load_const_slice<G>(y, rpow + (size_t)(RPOW_SPAN + 1 + A.dom_out - c) * G::NL);
            }
            // steps of this operand: [conv, dmax squarings] when it is raised, then the product with the accumulator (j > 0)
            const int npre = dmax > 0 ? 1 + dmax : 0;
            const int nsteps = npre + (j > 0 ? 1 : 0);
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {
                const bool conv = dmax > 0 && s == 0;
                const bool mul = s == npre;              // the last step of operands j > 0
                uint32_t lhs[G::NLL], r[G::NLL];
#pragma unroll
                for (int i = 0; i < G::NLL; ++i) lhs[i] = mul ? acc[i] : y[i];
                if (!conv) stage_b<G>(y, lds);           // right operand of squarings and of the product: y itself
                mont_mul<G::NLL, G::U, G::T>(r, lhs, conv ? conv_lds : o_lds, conv ? 1 : G::EPB, nm, n0inv);
                const bool keep_y = conv || (!mul && s - 1 < rj);
#pragma unroll
                for (int i = 0; i < G::NLL; ++i) {
                    acc[i] = mul ? r[i] : acc[i];
                    y[i] = keep_y ? r[i] : y[i];
                }
            }
            const int ty = j == A.k ? 1 + A.dom_out - c : (dmax > 0 ? 1 : (j == 0 ? A.tag0 : A.tag));   // y holds (operand) R^ty
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < G::NLL; ++i) acc[i] = y[i];
                c = ty;
            } else {
                c += ty - 1;
            }
        }
        cond_sub<G::NLL, G::T>(acc, nm);
        __builtin_amdgcn_s_setprio(2);
        pack_row<G>(acc, stage);
        store_tile<G>(stage, out + (size_t)row0 * w32, rows, w32);
        __builtin_amdgcn_s_setprio(0);
    }
}

}  // namespace pai
