// DJN obfuscator / encryption on base-n digit pairs for public moduli beyond the one-element-per-lane digit engine
// (n of 2049 .. 4156 bits: 3072- and 4096-bit keys), on the lane-group engine (mont_dev.hpp: pair_mul).
//
// An element of Z/n^2 is the pair (a, b), a + b n == x R (mod n^2), R = 2^(29 NL), NL = limbs of n spread over the T
// lanes of a group; a multiplication is 5 NL^2 limb products with reductions modulo n instead of the 8 NL^2 of the
// Montgomery product modulo n^2 that k_encrypt (kernels_paillier.hpp) runs on 2 NL limbs.
//   k_pair_fb_chain    first table level  S[i][e] = pair(hs^(e 2^(h i))),   one sequential chain per half-width window
//   k_pair_fb_expand   second level       T[j][hi 2^h + lo] = S[2 j + 1][hi] (x) S[2 j][lo],  one product per entry
//   k_pair_fixed_base  (w, v) = plain pair of  hs^r  or  hs^r (1 + m n)  =  prod_j T[j][r_j] (x) (1, m)
// The plain pair leaves as two packed rows per element; w + v n mod n^2 (one more product, on the 2 NL-limb geometry
// where the packed-word I/O lives) is k_pair_finish (kernels_paillier.hpp).  Same contract as k_encrypt mode 1 / 2:
// ipcl::PublicKey::encrypt / apply_obfuscator reached from bindings/ipcl_bindings_classes.cpp:53-60,71-83.
#pragma once
#include "kernels_common.hpp"

namespace pai {

// waves per SIMD the pair kernels are compiled for, by group size (tools/variant_tu.sh for A/B timing)
#define PAIR_WAVES_PER_SIMD(T) ((T) >= 8 ? 2 : 1)
#define PAIR_CT_WAVES_PER_SIMD(G) (G::NLL <= 9 ? 2 : PAIR_WAVES_PER_SIMD(G::T))

struct PairParams {
    const MontCtx* nctx;         // modulus n on the pair geometry (NL limbs, R = 2^(29 NL))
    const uint32_t* nm1;         // n - 1, NL limbs radix 29
    const uint32_t* fb_table;    // [J][2^fb_wbits][2][NL] lazy digit pairs, Montgomery digit form
    int fb_windows, fb_wbits;
    int pt_words, r_words;
    int out_words;               // packed words per output digit row (>= ceil(29 NL / 32))
    int fb_gform;                // g-factored table: entry = (a, t), x R == a (1 + n)^t (k_pair_g_prefix / k_pair_g_finish)
};

// LDS: [c rows][d rows] ([limb][element] operand buffers, G::LDS_WORDS each), then the NL limbs of n - 1
// Groups of 8 lanes (4096-bit keys) run two waves per SIMD and have no registers to spare for a prefetched table entry
// (the compiler spilled it straight after the load, i.e. waited for HBM once per window: 26 % of the wave cycles in
// profiles/r02/pmc_k4096_r02.json): there the next entry is STREAMED into a second pair of LDS operand buffers, two words per
// row block (RowStream, mont_dev.hpp).
template <class G>
struct PairLds {
    static constexpr bool STREAM = G::T >= 8;
    static constexpr int BYTES = (2 * G::LDS_WORDS + 2 * G::NL) * 4;
    static constexpr int BYTES_FB = BYTES + (STREAM ? 2 * G::LDS_WORDS * 4 : 0);      // k_pair_fixed_base: + the second buffer pair
    static constexpr int BYTES_CT = BYTES + 2 * G::LDS_WORDS * 4;                     // k_pair_ctmul: always two buffer pairs
    // g-factored tables on the non-streaming geometries: the lane slices of the exponent sum live in LDS (64-bit columns)
    static constexpr int BYTES_FBG = BYTES_FB + (STREAM ? 0 : G::NLL * 8 * BLOCK_THREADS);
    PAI_DEV static uint64_t* tsum(uint32_t* lds) { return reinterpret_cast<uint64_t*>(lds + BYTES_FB / 4) + threadIdx.x; }
    PAI_DEV static uint32_t* c2(uint32_t* lds) { return lds + 2 * G::LDS_WORDS + 2 * G::NL; }
    PAI_DEV static uint32_t* d2(uint32_t* lds) { return lds + 3 * G::LDS_WORDS + 2 * G::NL; }
    PAI_DEV static uint32_t* mod(uint32_t* lds) { return lds + 2 * G::LDS_WORDS + G::NL; }   // modulus copy (NMLDS geometries)
    PAI_DEV static uint32_t* c(uint32_t* lds) { return lds; }
    PAI_DEV static uint32_t* d(uint32_t* lds) { return lds + G::LDS_WORDS; }
    PAI_DEV static uint32_t* nm1(uint32_t* lds) { return lds + 2 * G::LDS_WORDS; }
};

template <class G>
PAI_DEV void pair_setup(uint32_t* lds, const uint32_t* __restrict__ nm1) {
    uint32_t* dst = PairLds<G>::nm1(lds);
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) dst[i] = nm1[i];
    __syncthreads();
}
// this lane's view of the modulus: registers, or (NMLDS) a copy behind the operand buffers
template <class G>
PAI_DEV void pair_load_modulus(typename G::NM& nm, const MontCtx* __restrict__ ctx, uint32_t* lds) {
    if constexpr (G::NMLDS) {
        uint32_t* dst = PairLds<G>::mod(lds);
        for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) dst[i] = ctx->n[i];
        __syncthreads();
        nm.p = dst + G::NLL * G::gl();
    } else {
        load_modulus<G>(nm, ctx, lds);
    }
}

// this lane's slices of a raw digit pair [2][NL] (16-byte vectors when the slice length allows, else 8-byte)
template <class G>
PAI_DEV void pair_load(uint32_t (&a)[G::NLL], uint32_t (&b)[G::NLL], const uint32_t* __restrict__ ent) {
    const uint32_t* pa = ent + G::NLL * G::gl();
    const uint32_t* pb = ent + G::NL + G::NLL * G::gl();
    if constexpr (G::NLL % 2 != 0) {                      // (odd slices: word by word)
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) { a[j] = pa[j]; b[j] = pb[j]; }
    } else if constexpr (G::NLL % 4 == 0) {
        const uint4* a4 = reinterpret_cast<const uint4*>(pa);
        const uint4* b4 = reinterpret_cast<const uint4*>(pb);
#pragma unroll
        for (int c = 0; c < G::NLL / 4; ++c) {
            const uint4 va = a4[c], vb = b4[c];
            a[4 * c] = va.x; a[4 * c + 1] = va.y; a[4 * c + 2] = va.z; a[4 * c + 3] = va.w;
            b[4 * c] = vb.x; b[4 * c + 1] = vb.y; b[4 * c + 2] = vb.z; b[4 * c + 3] = vb.w;
        }
    } else {
        const uint2* a2 = reinterpret_cast<const uint2*>(pa);
        const uint2* b2 = reinterpret_cast<const uint2*>(pb);
#pragma unroll
        for (int c = 0; c < G::NLL / 2; ++c) {
            const uint2 va = a2[c], vb = b2[c];
            a[2 * c] = va.x; a[2 * c + 1] = va.y;
            b[2 * c] = vb.x; b[2 * c + 1] = vb.y;
        }
    }
}
template <class G>
PAI_DEV void pair_store(const uint32_t (&a)[G::NLL], const uint32_t (&b)[G::NLL], uint32_t* __restrict__ ent) {
    uint32_t* pa = ent + G::NLL * G::gl();
    uint32_t* pb = ent + G::NL + G::NLL * G::gl();
    if constexpr (G::NLL % 2 != 0) {
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) { pa[j] = a[j]; pb[j] = b[j]; }
    } else if constexpr (G::NLL % 4 == 0) {
        uint4* a4 = reinterpret_cast<uint4*>(pa);
        uint4* b4 = reinterpret_cast<uint4*>(pb);
#pragma unroll
        for (int c = 0; c < G::NLL / 4; ++c) {
            a4[c] = make_uint4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
            b4[c] = make_uint4(b[4 * c], b[4 * c + 1], b[4 * c + 2], b[4 * c + 3]);
        }
    } else {
        uint2* a2 = reinterpret_cast<uint2*>(pa);
        uint2* b2 = reinterpret_cast<uint2*>(pb);
#pragma unroll
        for (int c = 0; c < G::NLL / 2; ++c) {
            a2[c] = make_uint2(a[2 * c], a[2 * c + 1]);
            b2[c] = make_uint2(b[2 * c], b[2 * c + 1]);
        }
    }
}
// this lane's slice of ONE raw digit [NL]
template <class G>
PAI_DEV void digit_load(uint32_t (&a)[G::NLL], const uint32_t* __restrict__ dig) {
    const uint32_t* pa = dig + G::NLL * G::gl();
    if constexpr (G::NLL % 2 != 0) {
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) a[j] = pa[j];
    } else if constexpr (G::NLL % 4 == 0) {
        const uint4* a4 = reinterpret_cast<const uint4*>(pa);
#pragma unroll
        for (int c = 0; c < G::NLL / 4; ++c) { const uint4 va = a4[c]; a[4 * c] = va.x; a[4 * c + 1] = va.y; a[4 * c + 2] = va.z; a[4 * c + 3] = va.w; }
    } else {
        const uint2* a2 = reinterpret_cast<const uint2*>(pa);
#pragma unroll
        for (int c = 0; c < G::NLL / 2; ++c) { const uint2 va = a2[c]; a[2 * c] = va.x; a[2 * c + 1] = va.y; }
    }
}
template <class G>
PAI_DEV void digit_store(const uint32_t (&a)[G::NLL], uint32_t* __restrict__ dig) {
    uint32_t* pa = dig + G::NLL * G::gl();
    if constexpr (G::NLL % 4 == 0) {
        uint4* a4 = reinterpret_cast<uint4*>(pa);
#pragma unroll
        for (int c = 0; c < G::NLL / 4; ++c) a4[c] = make_uint4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
    } else {
        uint2* a2 = reinterpret_cast<uint2*>(pa);
#pragma unroll
        for (int c = 0; c < G::NLL / 2; ++c) a2[c] = make_uint2(a[2 * c], a[2 * c + 1]);
    }
}
// (a, b) <- (a, b) (x) (c, d) with (c, d) in registers: staged as LDS rows, then the fused product
template <class G>
PAI_DEV void pair_times(uint32_t (&a)[G::NLL], uint32_t (&b)[G::NLL], const uint32_t (&c)[G::NLL],
                        const uint32_t (&d)[G::NLL], uint32_t* lds, const typename G::NM& nm, uint32_t n0inv) {
    stage_b<G>(c, PairLds<G>::c(lds));
    stage_b<G>(d, PairLds<G>::d(lds));
    pair_mul<G::NLL, G::U, G::T, PAIR_FULL, false>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB, nm,
                                                   n0inv);                   // one-off products: the compact form
}

// ---- first table level: element i walks S[i][e] = S[i][e - 1] (x) B_i, S[i][0] = pair(1) -------------------------------
// B_i = B_0^(2^(h i)) is reached by h i squarings of B_0 = pair(hs R) on the device (the host would need a long division
// per window base: 2.5 s of set-up at 4096-bit keys against ~0.1 s here); the elements of a workgroup run the longest
// chain among them and keep their own prefix.
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_fb_chain(const MontCtx* __restrict__ nctx, const uint32_t* __restrict__ nm1, const uint32_t* __restrict__ base0,
                const uint32_t* __restrict__ one_pair, uint32_t* __restrict__ S, int nwin, int h,
                const uint32_t* __restrict__ bases_plain, int base_words, const uint32_t* __restrict__ kdig, int nd) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pair_setup<G>(lds, nm1);
    typename G::NM nm;
    pair_load_modulus<G>(nm, nctx, lds);
    const uint32_t n0inv = nctx->n0inv;
    const int E1 = 1 << h;
    const int tiles = (nwin + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = tile * G::EPB + G::elem();
        const bool live = i < nwin;
        const int is = live ? i : nwin - 1;
        uint32_t a[G::NLL], b[G::NLL], c[G::NLL], d[G::NLL];
        pair_load<G>(c, d, base0);
        int smax = h * min(nwin - 1, tile * G::EPB + G::EPB - 1);
        if (bases_plain != nullptr) {
            // the window base B_i = hs^(2^(h i)) arrives as a plain residue modulo n^2 (k_sq_chain) and enters digit form
            // through its base-R digits, as a ciphertext does in k_pair_ctmul
            smax = 0;
            const uint32_t* row = bases_plain + (size_t)is * base_words;
#pragma unroll 1
            for (int i = 0; i < nd; ++i) {
                uint32_t kc[G::NLL], kd[G::NLL];
                load_elem_off<G>(a, row, base_words, G::NL * i);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) b[j] = 0;
                pair_load<G>(kc, kd, kdig + (size_t)i * 2 * G::NL);
                pair_times<G>(a, b, kc, kd, lds, nm, n0inv);
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) { c[j] = a[j]; d[j] = b[j]; }
                } else {
                    add_limbs<G>(c, a);
                    add_limbs<G>(d, b);
                }
            }
        }
#pragma unroll 1
        for (int sq = 0; sq < smax; ++sq) {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { a[j] = c[j]; b[j] = d[j]; }
            stage_b<G>(c, PairLds<G>::c(lds));
            stage_b<G>(d, PairLds<G>::d(lds));
            pair_mul<G::NLL, G::U, G::T, PAIR_SQR, false>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                                          nm, n0inv);       // a squaring: 4 NL^2
            const bool keep = sq < h * is;
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { c[j] = keep ? a[j] : c[j]; d[j] = keep ? b[j] : d[j]; }
        }
        stage_b<G>(c, PairLds<G>::c(lds));
        stage_b<G>(d, PairLds<G>::d(lds));
        pair_load<G>(a, b, one_pair);
        uint32_t* out = S + ((size_t)is << h) * 2 * G::NL;
        if (live) pair_store<G>(a, b, out);
#pragma unroll 1
        for (int e = 1; e < E1; ++e) {
            pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                         nm, n0inv);
            if (live) pair_store<G>(a, b, out + (size_t)e * 2 * G::NL);
        }
    }
}

// ---- second level: one independent product per entry ------------------------------------------------------------
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_fb_expand(const MontCtx* __restrict__ nctx, const uint32_t* __restrict__ nm1, const uint32_t* __restrict__ S,
                 uint32_t* __restrict__ T, int J, int h) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pair_setup<G>(lds, nm1);
    typename G::NM nm;
    pair_load_modulus<G>(nm, nctx, lds);
    const uint32_t n0inv = nctx->n0inv;
    const size_t half = (size_t)1 << h, per_window = half * half, total = (size_t)J * per_window;
    const size_t tiles = (total + G::EPB - 1) / G::EPB;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ei = tile * G::EPB + G::elem();
        const bool live = ei < total;
        const size_t es = live ? ei : total - 1;
        const size_t j = es / per_window, dd = es - j * per_window, hi = dd >> h, lo = dd & (half - 1);
        uint32_t a[G::NLL], b[G::NLL], c[G::NLL], d[G::NLL];
        pair_load<G>(a, b, S + (((2 * j + 1) << h) + hi) * 2 * G::NL);
        pair_load<G>(c, d, S + (((2 * j) << h) + lo) * 2 * G::NL);
        pair_times<G>(a, b, c, d, lds, nm, n0inv);
        if (live) pair_store<G>(a, b, T + ei * 2 * G::NL);
    }
}

// ---- g-factoring of a finished pair table (round 4; the lane-group counterpart of k_fb_g_prefix / k_fb_g_finish) -----------
// entry (a, d) -> (a, t = d a^-1 mod n) by Montgomery's simultaneous inversion over chunks of K consecutive entries, one chunk
// per lane GROUP; products are single Montgomery products modulo n (mont_mul, R = 2^(29 NL)) with the right operand staged
// as an LDS column.  pass 1: prefix products P_i = a_0 ... a_i R into `pref` ([entry][NL] raw limbs), the chunk total as a
// packed canonical residue into `tot`; the totals are inverted by launch_inv_eea; pass 2: the back sweep.
template <class G>
PAI_DEV void pair_g_setup(uint32_t* lds, const MontCtx* __restrict__ ctx, uint32_t*& r2_lds) {
    r2_lds = lds + G::LDS_WORDS + G::NL;                            // behind the operand buffer and a modulus copy
    for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) r2_lds[i] = ctx->r2[i];
    __syncthreads();
}
template <class G>
struct PairGLds { static constexpr int BYTES = (G::LDS_WORDS + 2 * G::NL) * 4; };

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_g_prefix(const MontCtx* __restrict__ nctx, const uint32_t* __restrict__ table, size_t count, int K,
                uint32_t* __restrict__ pref, uint32_t* __restrict__ tot, int tw) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, nctx, lds);
    uint32_t* r2_lds;
    pair_g_setup<G>(lds, nctx, r2_lds);
    const uint32_t n0inv = nctx->n0inv;
    const uint32_t* col = lds + G::elem();
    const size_t nchunks = count / (size_t)K;
    const size_t tiles = (nchunks + G::EPB - 1) / G::EPB;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ch = tile * G::EPB + G::elem();
        const bool live = ch < nchunks;
        const size_t cs = live ? ch : nchunks - 1;
        uint32_t P[G::NLL], x[G::NLL], t[G::NLL];
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const size_t g = (size_t)i * nchunks + cs;      // interleaved chunks: at step i the lanes of a wave touch consecutive entries
            digit_load<G>(x, table + g * 2 * G::NL);                                  // a_i
            if (i > 0) {
                stage_b<G>(P, lds);
                mont_mul<G::NLL, G::U, G::T>(t, x, col, G::EPB, nm, n0inv);           // P_i = P_(i-1) a_i R^-1 = a_0 ... a_i R^-i
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) P[j] = t[j];
            } else {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) P[j] = x[j];
            }
            if (live) digit_store<G>(P, pref + g * G::NL);
        }
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) t[j] = P[j];                                 // the chunk total a_0 ... a_(K-1) R^-(K-1), canonical
        cond_sub<G::NLL, G::T>(t, nm);
        cond_sub<G::NLL, G::T>(t, nm);
        // every group runs store_elem (it stages through LDS with wave-level fences); dead groups write their clamped chunk's
        // own total again: the same value
        store_elem<G>(t, tot + cs * (size_t)tw, tw, lds);
    }
}

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_g_finish(const MontCtx* __restrict__ nctx, uint32_t* __restrict__ table, size_t count, int K,
                const uint32_t* __restrict__ pref, const uint32_t* __restrict__ inv, int tw) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    typename G::NM nm;
    load_modulus<G>(nm, nctx, lds);
    uint32_t* r2_lds;
    pair_g_setup<G>(lds, nctx, r2_lds);
    const uint32_t n0inv = nctx->n0inv;
    const uint32_t* col = lds + G::elem();
    const size_t nchunks = count / (size_t)K;
    const size_t tiles = (nchunks + G::EPB - 1) / G::EPB;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t ch = tile * G::EPB + G::elem();
        const bool live = ch < nchunks;
        const size_t cs = live ? ch : nchunks - 1;
        uint32_t I[G::NLL], x[G::NLL], u[G::NLL], t[G::NLL];
        load_elem<G>(x, inv + cs * (size_t)tw, tw);
        mont_mul<G::NLL, G::U, G::T>(I, x, r2_lds, 1, nm, n0inv);                     // (a_0 ... a_(K-1))^-1 R^(K-1) * R
#pragma unroll 1
        for (int i = K - 1; i >= 0; --i) {
            const size_t g = (size_t)i * nchunks + cs;      // interleaved chunks: at step i the lanes of a wave touch consecutive entries
            uint32_t* ent = table + g * 2 * G::NL;
            if (i > 0) {
                digit_load<G>(x, pref + (g - nchunks) * G::NL);
                stage_b<G>(x, lds);
                mont_mul<G::NLL, G::U, G::T>(u, I, col, G::EPB, nm, n0inv);           // a_i^-1 R
            } else {
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) u[j] = I[j];
            }
            digit_load<G>(x, ent + G::NL);                                            // d_i
            stage_b<G>(x, lds);
            mont_mul<G::NLL, G::U, G::T>(t, u, col, G::EPB, nm, n0inv);               // t_i = d_i a_i^-1 (plain)
            cond_sub<G::NLL, G::T>(t, nm);
            cond_sub<G::NLL, G::T>(t, nm);
            if (live) digit_store<G>(t, ent + G::NL);
            if (i > 0) {
                digit_load<G>(x, ent);                                                // a_i
                stage_b<G>(x, lds);
                mont_mul<G::NLL, G::U, G::T>(t, I, col, G::EPB, nm, n0inv);           // running inverse: one power of R less per step
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) I[j] = t[j];
            }
        }
    }
}

// ---- hs^r (and the plaintext factor) ------------------------------------------------------------------------------
// with_m != 0: (w, v) = plain pair of hs^r (1 + m n): the last factor is the PLAIN pair (1, m), which also takes the
// product out of Montgomery form; with_m == 0: plain pair of hs^r (last factor (1, 0)).
// wv_out: [n][2][out_words] packed rows, w first.
// GFORM: the table holds g-factored entries (a, t) — x R == a (1 + n)^t, built by k_pair_g_prefix / k_pair_g_finish —: the
// table products take the rule for a right operand without a second digit (pair_mul c0: 4 NL^2), the exponents t are summed
// in lane-sliced 64-bit columns, and g^(m + sum t) joins through the final plain pair (1, s).
template <class G, bool GFORM>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_fixed_base(PairParams P, const uint32_t* __restrict__ m, const uint32_t* __restrict__ r,
                  uint32_t* __restrict__ wv_out, int n, int with_m) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pair_setup<G>(lds, P.nm1);
    typename G::NM nm;
    pair_load_modulus<G>(nm, P.nctx, lds);
    const uint32_t n0inv = P.nctx->n0inv;
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* rrow = r + (size_t)es * P.r_words;
        auto entry = [&](int jw) -> const uint32_t* {
            const int bit = jw * P.fb_wbits, k = bit >> 5;
            uint64_t bits2 = rrow[k];
            if (k + 1 < P.r_words) bits2 |= (uint64_t)rrow[k + 1] << 32;
            const uint32_t d = (uint32_t)(bits2 >> (bit & 31)) & ((1u << P.fb_wbits) - 1u);
            return P.fb_table + (((size_t)jw << P.fb_wbits) + d) * 2 * G::NL;
        };
        uint32_t a[G::NLL], b[G::NLL], c[G::NLL], d[G::NLL];
        // lane slice of sum t (128 windows x 2^29 stay below 2^37): registers on the streaming geometries (8 lanes x 18 limbs),
        // LDS columns on the others (4 lanes x 28 limbs: 56 more registers made the row loop shuffle through AGPRs)
        constexpr bool TREG = GFORM && PairLds<G>::STREAM;
        uint64_t tacc[TREG ? G::NLL : 1];
        uint64_t* tl = PairLds<G>::tsum(lds);
        pair_load<G>(a, b, entry(0));
        if constexpr (GFORM) {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) {                     // entry 0: (a, 0) and its exponent
                if constexpr (TREG) tacc[j] = b[j];
                else tl[j * BLOCK_THREADS] = b[j];
                b[j] = 0;
            }
        }
        if constexpr (PairLds<G>::STREAM) {
            constexpr int CH = (G::NLL % 4 == 0) ? 4 : 2, NCH = G::NLL / CH;
            // buffer pair k lives at lds + k * PAIR_OFF (c rows, then d rows): plain offsets from the shared base — a runtime
            // choice between two POINTERS makes the compiler lose the LDS address space (flat_load in the row loop)
            constexpr int PAIR_OFF = 2 * G::LDS_WORDS + 2 * G::NL;
            if (P.fb_windows > 1) {
                pair_load<G>(c, d, entry(1));
                stage_b<G>(c, lds + PAIR_OFF);
                if constexpr (GFORM) {
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) tacc[j] += d[j];
                } else {
                    stage_b<G>(d, lds + PAIR_OFF + G::LDS_WORDS);
                }
            }
            const int col = (G::NLL * G::gl()) * G::EPB + G::elem();
#pragma unroll 1
            for (int jw = 1; jw < P.fb_windows; ++jw) {
                const int cur = (jw & 1) * PAIR_OFF, nxt = PAIR_OFF - cur;
                // the next window's entry streams into the other buffer pair while this product runs (the last window
                // re-reads its own entry: harmless, keeps the loop body uniform)
                const uint32_t* ent = entry(jw + 1 < P.fb_windows ? jw + 1 : jw);
                if constexpr (GFORM) {
                    RowStream<CH, NCH, 0> pf;                      // only the first digit goes to LDS
                    pf.src0 = ent + G::NLL * G::gl();
                    pf.src1 = pf.src0;
                    pf.dst0 = lds + nxt + col;
                    pf.dst1 = pf.dst0;
                    pf.stride = G::EPB;
                    const bool more = jw + 1 < P.fb_windows;
                    uint32_t tn[G::NLL];                           // the next entry's exponent slice travels during the product
                    digit_load<G>(tn, ent + G::NL);                // (measured: 37.4 ms against 38.5 ms with the load after it)
                    pair_mul<G::NLL, G::U, G::T, PAIR_C0>(a, b, lds + cur + G::elem(), lds + cur + G::elem(), G::EPB, nm, n0inv, &pf);
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) tacc[j] += more ? tn[j] : 0u;
                } else {
                    RowStream<CH, NCH, NCH> pf;
                    pf.src0 = ent + G::NLL * G::gl();
                    pf.src1 = ent + G::NL + G::NLL * G::gl();
                    pf.dst0 = lds + nxt + col;
                    pf.dst1 = lds + nxt + G::LDS_WORDS + col;
                    pf.stride = G::EPB;
                    pair_mul<G::NLL, G::U, G::T>(a, b, lds + cur + G::elem(), lds + cur + G::LDS_WORDS + G::elem(), G::EPB,
                                                 nm, n0inv, &pf);
                }
                wave_lds_fence();
            }
        } else if constexpr (GFORM) {
            // g-factored entries: only the first digit is prefetched across the product; the exponent slice of the window just
            // multiplied is fetched after it (half the prefetch registers)
            if (P.fb_windows > 1) digit_load<G>(c, entry(1));
#pragma unroll 1
            for (int jw = 1; jw < P.fb_windows; ++jw) {
                stage_b<G>(c, PairLds<G>::c(lds));
                if (jw + 1 < P.fb_windows) digit_load<G>(c, entry(jw + 1));
                pair_mul<G::NLL, G::U, G::T, PAIR_C0>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::c(lds) + G::elem(), G::EPB,
                                                      nm, n0inv);
                digit_load<G>(d, entry(jw) + G::NL);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) tl[j * BLOCK_THREADS] += d[j];
            }
        } else {
        if (P.fb_windows > 1) pair_load<G>(c, d, entry(1));
#pragma unroll 1
        for (int jw = 1; jw < P.fb_windows; ++jw) {
            stage_b<G>(c, PairLds<G>::c(lds));
            stage_b<G>(d, PairLds<G>::d(lds));
            // the next window's entry travels from HBM while this product runs
            if (jw + 1 < P.fb_windows) pair_load<G>(c, d, entry(jw + 1));
            pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                         nm, n0inv);
        }
        }
        set_plain_one<G>(c);
        if (with_m) load_elem<G>(d, m + (size_t)es * P.pt_words, P.pt_words);
        else {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) d[j] = 0;
        }
        if constexpr (GFORM) {
            // s = m + sum t (< 2^8 n, inside the digit): carry-propagated across the group's lanes
            using RW = Rows<G::NLL, G::U, G::T>;
            uint64_t sw[RW::NW];
#pragma unroll
            for (int j = 0; j < RW::NW; ++j) {
                if constexpr (TREG) sw[j] = j < G::NLL ? tacc[j] + d[j] : 0ull;
                else sw[j] = j < G::NLL ? tl[j * BLOCK_THREADS] + d[j] : 0ull;
            }
            RW::finish(sw, d);
        }
        pair_times<G>(a, b, c, d, lds, nm, n0inv);
        if (live) {
            uint32_t* row = wv_out + (size_t)ei * 2 * P.out_words;
            store_elem<G>(a, row, P.out_words, PairLds<G>::c(lds));
            store_elem<G>(b, row + P.out_words, P.out_words, PairLds<G>::c(lds));
        }
    }
}

// ---- ct^e for per-element exponents (ciphertext * plaintext, CipherText::operator*, classes.cpp:324-325) ------------------
// The lane-group counterpart of k_ctmul_padic for n of 2049 .. 4156 bits (BASELINE configs 4 and 5).  Round 2 ran this
// operation as Montgomery products modulo n^2 (k_modexp_var_win: 8 NL^2 limb products each, NL = limbs of n); a 53-bit
// exponent is 52 squarings and ~22 multiplications, and on digit pairs a SQUARING is 4 NL^2 (a a, its reduction, 2 a b, its
// reduction) and a multiplication 5 NL^2.  Flow: the ciphertext enters digit form through its base-R digits,
// sum_i (D_i, 0) (x) pair(R^(i+2) mod n^2) (as stage A of decryption does); per-slot table of the powers 0 .. 2^w - 1,
// row-major [slot][entry][2][NL]; per window w squarings and one multiplication whose table entry STREAMS into the second
// LDS buffer pair during the window's last squaring (RowStream); the plain pair (1, 0) takes the power out of Montgomery
// form and (w, v) leaves as two packed rows for k_pair_finish (w + v n on the n^2 geometry).
struct PairCtMulParams {
    const MontCtx* nctx;         // modulus n on the pair geometry
    const uint32_t* nm1;         // n - 1
    const uint32_t* kdig;        // [nd][2][NL] pairs of R^(i+2) mod n^2
    const uint32_t* one_pair;    // pair(R mod n^2): the Montgomery digit form of 1
    uint32_t* table;             // [grid * EPB][2^wbits][2][NL]
    int nd, wbits, ct_words, e_words, ebits_max, e_bcast, out_words;
    // a second modulus for the workgroups with blockIdx.y == 1 (decryption stage A of mid-size batches: s = p and s = q in one
    // launch, the same ciphertexts, exponents s - 1); unused with gridDim.y == 1
    const MontCtx* nctx1 = nullptr;
    const uint32_t* nm11 = nullptr;
    const uint32_t* kdig1 = nullptr;
    const uint32_t* one_pair1 = nullptr;
    uint32_t* table1 = nullptr;
    const uint32_t* e1 = nullptr;
    uint32_t* wv1 = nullptr;
    int e_words1 = 0, ebits_max1 = 0;
};

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_CT_WAVES_PER_SIMD(G))
k_pair_ctmul(PairCtMulParams P, const uint32_t* __restrict__ ct, const uint32_t* __restrict__ e_, uint32_t* __restrict__ wv_out_, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t* __restrict__ e = e_;
    uint32_t* __restrict__ wv_out = wv_out_;
    if (blockIdx.y == 1) {
        P.nctx = P.nctx1; P.nm1 = P.nm11; P.kdig = P.kdig1; P.one_pair = P.one_pair1; P.table = P.table1;
        P.e_words = P.e_words1; P.ebits_max = P.ebits_max1;
        e = P.e1; wv_out = P.wv1;
    }
    pair_setup<G>(lds, P.nm1);
    typename G::NM nm;
    pair_load_modulus<G>(nm, P.nctx, lds);
    const uint32_t n0inv = P.nctx->n0inv;
    constexpr int PAIR_OFF = 2 * G::LDS_WORDS + 2 * G::NL;          // second buffer pair (plain offsets: see k_pair_fixed_base)
    constexpr int CH = (G::NLL % 4 == 0) ? 4 : (G::NLL % 2 == 0 ? 2 : 1), NCH = G::NLL / CH;
    const int t = G::gl();
    const int W = P.wbits, NT = 1 << W;
    const int nwin = (P.ebits_max + W - 1) / W;
    const size_t slot = (size_t)blockIdx.x * G::EPB + G::elem();
    uint32_t* trow = P.table + slot * (size_t)NT * 2 * G::NL;
    const int col = (G::NLL * t) * G::EPB + G::elem();
    const int tiles = (n + G::EPB - 1) / G::EPB;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile * G::EPB + G::elem();
        const bool live = ei < n;
        const int es = live ? ei : n - 1;
        const uint32_t* erow = e + (size_t)(P.e_bcast ? 0 : es) * P.e_words;
        auto window = [&](int wi) -> int {
            const int bit = wi * W, k = bit >> 5;
            uint64_t bits2 = k < P.e_words ? erow[k] : 0u;
            if (k + 1 < P.e_words) bits2 |= (uint64_t)erow[k + 1] << 32;
            return (int)((uint32_t)(bits2 >> (bit & 31)) & (uint32_t)(NT - 1));
        };
        uint32_t a[G::NLL], b[G::NLL];
        {   // digit form of the ciphertext (Montgomery digit form: the pair of ct R)
            const uint32_t* row = ct + (size_t)es * P.ct_words;
            uint32_t sa[G::NLL], sb[G::NLL], c[G::NLL], d[G::NLL];
#pragma unroll 1
            for (int i = 0; i < P.nd; ++i) {
                load_elem_off<G>(a, row, P.ct_words, G::NL * i);
#pragma unroll
                for (int j = 0; j < G::NLL; ++j) b[j] = 0;
                pair_load<G>(c, d, P.kdig + (size_t)i * 2 * G::NL);
                pair_times<G>(a, b, c, d, lds, nm, n0inv);
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < G::NLL; ++j) { sa[j] = a[j]; sb[j] = b[j]; }
                } else {
                    add_limbs<G>(sa, a);
                    add_limbs<G>(sb, b);
                }
            }
            // table: T[0] = 1, T[1] = x, T[k] = T[k - 1] x  (x staged once as the right operand)
            pair_load<G>(c, d, P.one_pair);
            pair_store<G>(c, d, trow);
            pair_store<G>(sa, sb, trow + 2 * G::NL);
            stage_b<G>(sa, PairLds<G>::c(lds));
            stage_b<G>(sb, PairLds<G>::d(lds));
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) { a[j] = sa[j]; b[j] = sb[j]; }
#pragma unroll 1
            for (int k = 2; k < NT; ++k) {
                pair_mul<G::NLL, G::U, G::T, PAIR_FULL, false>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(),
                                                               G::EPB, nm, n0inv);
                pair_store<G>(a, b, trow + (size_t)k * 2 * G::NL);
            }
        }
        pair_load<G>(a, b, trow + (size_t)window(nwin - 1) * 2 * G::NL);
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
            const int dg = window(wi);
            const bool any = __any(dg != 0);
            RowStream<CH, NCH, NCH> pf;
            pf.src0 = trow + (size_t)dg * 2 * G::NL + G::NLL * t;
            pf.src1 = pf.src0 + G::NL;
            pf.dst0 = lds + PAIR_OFF + col;
            pf.dst1 = lds + PAIR_OFF + G::LDS_WORDS + col;
            pf.stride = G::EPB;
            const int nsteps = W + (any ? 1 : 0);
#pragma unroll 1
            for (int s = 0; s < nsteps; ++s) {                 // one rolled body: W squarings, then the multiplication
                const bool is_mul = s == W;
                if (!is_mul) {
                    stage_b<G>(a, PairLds<G>::c(lds));
                    stage_b<G>(b, PairLds<G>::d(lds));
                } else {
                    wave_lds_fence();
                }
                pf.on = any && s == W - 1;
                // two straight-line forms (no wave-uniform branches inside the rows): the squaring streams the window's table
                // entry into the second buffer pair, the multiplication reads it from there
                if (!is_mul) pair_mul<G::NLL, G::U, G::T, PAIR_SQR>(a, b, lds + G::elem(), lds + G::LDS_WORDS + G::elem(), G::EPB, nm, n0inv, &pf);
                else pair_mul<G::NLL, G::U, G::T, PAIR_FULL>(a, b, lds + PAIR_OFF + G::elem(), lds + PAIR_OFF + G::LDS_WORDS + G::elem(), G::EPB,
                                                             nm, n0inv);
            }
        }
        {   // leave Montgomery form: times the plain pair (1, 0)
            uint32_t c[G::NLL], d[G::NLL];
            set_plain_one<G>(c);
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) d[j] = 0;
            pair_times<G>(a, b, c, d, lds, nm, n0inv);
        }
        if (live) {
            uint32_t* row = wv_out + (size_t)ei * 2 * P.out_words;
            store_elem<G>(a, row, P.out_words, PairLds<G>::c(lds));
            store_elem<G>(b, row + P.out_words, P.out_words, PairLds<G>::c(lds));
        }
    }
}

}  // namespace pai
