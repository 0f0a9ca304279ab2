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
#ifndef PAIR_WAVES_T8
#define PAIR_WAVES_T8 2
#endif
#define PAIR_WAVES_PER_SIMD(T) ((T) >= 8 ? PAIR_WAVES_T8 : 1)
#ifndef PAIR_PREFETCH
#define PAIR_PREFETCH 1             // the next window's table entry is loaded while the current product runs
#endif

struct PairParams {
    const MontCtx* nctx;         // modulus n on the pair geometry (NL limbs, R = 2^(29 NL))
    const uint32_t* nm1;         // n - 1, NL limbs radix 29
    const uint32_t* fb_table;    // [J][2^fb_wbits][2][NL] lazy digit pairs, Montgomery digit form
    int fb_windows, fb_wbits;
    int pt_words, r_words;
    int out_words;               // packed words per output digit row (>= ceil(29 NL / 32))
};

// LDS: [c rows][d rows] ([limb][element] operand buffers, G::LDS_WORDS each), then the NL limbs of n - 1
// Groups of 8 lanes (4096-bit keys) run two waves per SIMD and have no registers to spare for a prefetched table entry
// (the compiler spilled it straight after the load, i.e. waited for HBM once per window: 26 % of the wave cycles in
// profiles/r02/pmc_k4096_r02.json): there the next entry is STREAMED into a second pair of LDS operand buffers, two words per
// row block (RowStream, mont_dev.hpp).
#ifndef PAIR_STREAM_T
#define PAIR_STREAM_T 8
#endif
template <class G>
struct PairLds {
    static constexpr bool STREAM = G::T >= PAIR_STREAM_T;
    static constexpr int BYTES = (2 * G::LDS_WORDS + 2 * G::NL) * 4;
    static constexpr int BYTES_FB = BYTES + (STREAM ? 2 * G::LDS_WORDS * 4 : 0);      // k_pair_fixed_base: + the second buffer pair
    static constexpr int BYTES_CT = BYTES + 2 * G::LDS_WORDS * 4;                     // k_pair_ctmul: always two buffer pairs
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
    static_assert(G::NLL % 2 == 0, "limb slices move as vectors");
    const uint32_t* pa = ent + G::NLL * G::gl();
    const uint32_t* pb = ent + G::NL + G::NLL * G::gl();
    if constexpr (G::NLL % 4 == 0) {
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
    if constexpr (G::NLL % 4 == 0) {
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
// (a, b) <- (a, b) (x) (c, d) with (c, d) in registers: staged as LDS rows, then the fused product
template <class G>
PAI_DEV void pair_times(uint32_t (&a)[G::NLL], uint32_t (&b)[G::NLL], const uint32_t (&c)[G::NLL],
                        const uint32_t (&d)[G::NLL], uint32_t* lds, const typename G::NM& nm, uint32_t n0inv) {
    stage_b<G>(c, PairLds<G>::c(lds));
    stage_b<G>(d, PairLds<G>::d(lds));
    pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                 PairLds<G>::nm1(lds), nm, n0inv);
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
            pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                         PairLds<G>::nm1(lds), nm, n0inv, (NoStream*)nullptr, true);       // a squaring: 4 NL^2
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
                                         PairLds<G>::nm1(lds), nm, n0inv);
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

// ---- hs^r (and the plaintext factor) ------------------------------------------------------------------------------
// with_m != 0: (w, v) = plain pair of hs^r (1 + m n): the last factor is the PLAIN pair (1, m), which also takes the
// product out of Montgomery form; with_m == 0: plain pair of hs^r (last factor (1, 0)).
// wv_out: [n][2][out_words] packed rows, w first.
template <class G>
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
        pair_load<G>(a, b, entry(0));
        if constexpr (PairLds<G>::STREAM) {
            constexpr int CH = (G::NLL % 4 == 0) ? 4 : 2, NCH = G::NLL / CH;
            // buffer pair k lives at lds + k * PAIR_OFF (c rows, then d rows): plain offsets from the shared base — a runtime
            // choice between two POINTERS makes the compiler lose the LDS address space (flat_load in the row loop)
            constexpr int PAIR_OFF = 2 * G::LDS_WORDS + 2 * G::NL;
            if (P.fb_windows > 1) {
                pair_load<G>(c, d, entry(1));
                stage_b<G>(c, lds + PAIR_OFF);
                stage_b<G>(d, lds + PAIR_OFF + G::LDS_WORDS);
            }
            const int col = (G::NLL * G::gl()) * G::EPB + G::elem();
#pragma unroll 1
            for (int jw = 1; jw < P.fb_windows; ++jw) {
                const int cur = (jw & 1) * PAIR_OFF, nxt = PAIR_OFF - cur;
                // the next window's entry streams into the other buffer pair while this product runs (the last window
                // re-reads its own entry: harmless, keeps the loop body uniform)
                const uint32_t* ent = entry(jw + 1 < P.fb_windows ? jw + 1 : jw);
                RowStream<CH, NCH, NCH> pf;
                pf.src0 = ent + G::NLL * G::gl();
                pf.src1 = ent + G::NL + G::NLL * G::gl();
                pf.dst0 = lds + nxt + col;
                pf.dst1 = lds + nxt + G::LDS_WORDS + col;
                pf.stride = G::EPB;
                pair_mul<G::NLL, G::U, G::T>(a, b, lds + cur + G::elem(), lds + cur + G::LDS_WORDS + G::elem(), G::EPB,
                                             PairLds<G>::nm1(lds), nm, n0inv, &pf);
                wave_lds_fence();
            }
        } else {
        if (P.fb_windows > 1) pair_load<G>(c, d, entry(1));
#pragma unroll 1
        for (int jw = 1; jw < P.fb_windows; ++jw) {
#if PAIR_PREFETCH
            stage_b<G>(c, PairLds<G>::c(lds));
            stage_b<G>(d, PairLds<G>::d(lds));
            // the next window's entry travels from HBM while this product runs
            if (jw + 1 < P.fb_windows) pair_load<G>(c, d, entry(jw + 1));
#else
            if (jw > 1) pair_load<G>(c, d, entry(jw));
            stage_b<G>(c, PairLds<G>::c(lds));
            stage_b<G>(d, PairLds<G>::d(lds));
#endif
            pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB,
                                         PairLds<G>::nm1(lds), nm, n0inv);
        }
        }
        set_plain_one<G>(c);
        if (with_m) load_elem<G>(d, m + (size_t)es * P.pt_words, P.pt_words);
        else {
#pragma unroll
            for (int j = 0; j < G::NLL; ++j) d[j] = 0;
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
};

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, PAIR_WAVES_PER_SIMD(G::T))
k_pair_ctmul(PairCtMulParams P, const uint32_t* __restrict__ ct, const uint32_t* __restrict__ e, uint32_t* __restrict__ wv_out, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pair_setup<G>(lds, P.nm1);
    typename G::NM nm;
    pair_load_modulus<G>(nm, P.nctx, lds);
    const uint32_t n0inv = P.nctx->n0inv;
    constexpr int PAIR_OFF = 2 * G::LDS_WORDS + 2 * G::NL;          // second buffer pair (plain offsets: see k_pair_fixed_base)
    constexpr int CH = (G::NLL % 4 == 0) ? 4 : 2, NCH = G::NLL / CH;
    const int t = G::gl();
    const int W = P.wbits, NT = 1 << W;
    const int nwin = (P.ebits_max + W - 1) / W;
    const size_t slot = (size_t)blockIdx.x * G::EPB + G::elem();
    uint32_t* trow = P.table + slot * (size_t)NT * 2 * G::NL;
    const int col = (G::NLL * t) * G::EPB + G::elem();
    const uint32_t* mm1 = PairLds<G>::nm1(lds);
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
                pair_mul<G::NLL, G::U, G::T>(a, b, PairLds<G>::c(lds) + G::elem(), PairLds<G>::d(lds) + G::elem(), G::EPB, mm1, nm, n0inv);
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
                const int off = is_mul ? PAIR_OFF : 0;
                pair_mul<G::NLL, G::U, G::T>(a, b, lds + off + G::elem(), lds + off + G::LDS_WORDS + G::elem(), G::EPB, mm1, nm, n0inv,
                                             &pf, !is_mul);
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
