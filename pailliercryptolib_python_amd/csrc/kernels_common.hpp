// Shared device helpers for the Paillier kernels: element I/O (packed u32 words <-> radix-2^29
// limb slices spread over a lane group), LDS staging of the multiplier operand, uniform exponent
// window extraction.  See mont_dev.hpp for the arithmetic itself.
#pragma once
#include <type_traits>
#include "mont_dev.hpp"

namespace pai {

constexpr int BLOCK_THREADS = 256;
constexpr int MODMUL_FULL = 0, MODMUL_MONT = 1;      // k_modmul modes (kernels_modexp.hpp)

// Waves per SIMD the exponentiation-type lane-group kernels are compiled for: two, except where the 8-lane geometries
// (28x8, 36x8: keys above 2048 bits) spill INSIDE their row blocks at 256 registers — measured per kernel at one wave (512
// registers): k_mexp<36x8> 101 -> 54 ms, k_modexp_var_win<28x8> 22.1 -> 20.6 ms per 65 536 (one wave each), but
// k_modexp_var_win<36x8> 32.2 -> 34.4 and k_mexp<28x8> 31.7 -> 33.8 (they keep two).
#define PAI_LG_WAVES(G) 2
#define PAI_MEXP_WAVES(G) (((G::T) >= 8 && (G::NLL) >= 36) ? 1 : 2)
#define PAI_VARWIN_WAVES(G) (((G::T) >= 8 && (G::NLL) < 36) ? 1 : 2)

// shape of a multi-exponentiation on the lane-group engine (kernels_modexp.hpp: k_mexp)
struct MexpParams {
    int R, K, M, chunk, nsigns, e_words, ebits_max, wbits, w32;
};

// Geometry of one kernel instance: NLL limbs per lane, T lanes per element, U rows per block,
// NMLDS = the modulus slice is re-read from LDS during the q*n step instead of living in VGPRs.
// M1 = "minus-one" Montgomery contexts (mont_dev.hpp: Rows::block_m1) — the wide-group latency kernels.
template <int NLL_, int T_, int U_, bool NMLDS_ = false, bool M1_ = false>
struct Geo {
    static constexpr int NLL = NLL_, T = T_, U = U_;
    static constexpr bool NMLDS = NMLDS_;
    static constexpr bool M1 = M1_;
    static constexpr int NL = NLL * T;                 // limbs per element
    static constexpr int EPB = BLOCK_THREADS / T;      // elements per workgroup
    static constexpr int LDS_WORDS = NL * EPB;         // one [limb][element] operand buffer
    static constexpr int LDS_BYTES = (LDS_WORDS + NL) * 4;   // + the modulus copy behind it
    using NM = typename std::conditional<NMLDS_, NmLds<NLL_>, NmRegs<NLL_>>::type;
    // Raw-row staging area (load_tile / unpack_row): EPB rows of SW packed words.  SW covers the geometry's whole
    // capacity plus the word a lane's funnel shift peeks at, rounded to 16 bytes; words beyond a row's W32 are zero.
    static constexpr int SW = (((NL * RB + 31) / 32 + 1) + 3) / 4 * 4;
    static constexpr int NWIN = (RB * NLL + 62) / 32;  // packed words overlapped by one lane's slice (any alignment)
    static constexpr int STAGE_WORDS = EPB * SW;
    static constexpr int STAGE_BYTES = STAGE_WORDS * 4;
    static_assert(((RB * NLL * (T - 1)) >> 5) + NWIN <= SW, "a lane's word window must stay inside the staged row");
    PAI_DEV static int elem() { return (int)threadIdx.x / T; }
    PAI_DEV static int gl() { return (int)threadIdx.x & (T - 1); }
};

// Set up this lane's view of the modulus.  Must be called by every thread of the block once, before
// any arithmetic (contains a __syncthreads when the LDS copy is used).
template <class G>
PAI_DEV void load_modulus(typename G::NM& nm, const MontCtx* __restrict__ ctx, uint32_t* lds) {
    if constexpr (G::NMLDS) {
        uint32_t* dst = lds + G::LDS_WORDS;
        for (int i = threadIdx.x; i < G::NL; i += BLOCK_THREADS) dst[i] = ctx->n[i];
        __syncthreads();
        nm.p = dst + G::NLL * G::gl();
    } else {
        const int t = G::gl();
        const uint32_t* src = G::M1 ? ctx->npp : ctx->n;     // minus-one contexts multiply the quotient digits by (M + 1) / 2^(29 U)
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) nm.v[j] = src[G::NLL * t + j];
    }
}

// this lane's slice of a NLMAX-padded constant (modulus, R^2, ...)
template <class G>
PAI_DEV void load_const_slice(uint32_t (&x)[G::NLL], const uint32_t* __restrict__ c) {
    const int t = G::gl();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) x[j] = c[G::NLL * t + j];
}

// packed little-endian u32 words of one element (row pointer, W32 words) -> this lane's limb slice.
// Words at index >= W32 read as zero.  The value must fit in 29*NL bits.
template <class G>
PAI_DEV void load_elem(uint32_t (&x)[G::NLL], const uint32_t* __restrict__ row, int W32) {
    const int t = G::gl();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) {
        const int bit = RB * (G::NLL * t + j);
        const int k = bit >> 5, s = bit & 31;
        // branch-free: clamp the indices, zero what lies beyond the row
        const int k0 = k < W32 ? k : W32 - 1, k1 = k + 1 < W32 ? k + 1 : W32 - 1;
        uint32_t lo = row[k0], hi = row[k1];
        lo = k < W32 ? lo : 0u;
        hi = k + 1 < W32 ? hi : 0u;
        const uint64_t v = ((uint64_t)hi << 32) | lo;
        x[j] = (uint32_t)(v >> s) & RMASK;
        // keep the compiler from hoisting all 2*NLL loads ahead of their uses (register pressure)
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- coalesced tile I/O ------------------------------------------------------------------------------------
// load_elem above issues 2*NLL dependent-address dword loads per lane (the compiler hoists the 2*NLL per-limb
// offsets out of the tile loop, spills them, and serialises load -> wait -> extract: ~100 us per tile on MI355X,
// which is what bounded k_modmul in round 1).  Kernels whose arithmetic per tile is short use the helpers below
// instead.  The unit is a WAVE tile: the 64 / T elements of one wavefront, whose packed rows are consecutive in
// memory.  The wave copies them HBM -> LDS with full-width coalesced loads into its own slice of the staging area
// (rows elem() of stride G::SW), then every lane funnel-shifts its bit window out of LDS (one runtime shift per
// lane, compile-time limb positions after it).  Nothing here needs a workgroup barrier — lane groups never span
// waves — so the waves of a workgroup run their load / multiply / store phases independently and the two waves
// that share a SIMD overlap one's memory phase with the other's multiply phase.
template <class G>
struct WaveTile {
    static constexpr int EPW = 64 / G::T;                    // elements (rows) per wave
    static constexpr int SV = G::SW / 4;
    static constexpr int IT4 = (EPW * SV + 63) / 64;         // 16-byte copy iterations per lane
    PAI_DEV static int lane() { return (int)threadIdx.x & 63; }
    PAI_DEV static int wave() { return (int)threadIdx.x >> 6; }
    PAI_DEV static uint32_t* slice(uint32_t* stage) { return stage + wave() * EPW * G::SW; }
};

// Zeroes this wave's staging slice once per kernel: the pad words behind every row (the funnel shift of unpack_row
// peeks at them) and the rows of a ragged last tile then read as zero / stale-but-unused without per-tile filling.
template <class G>
PAI_DEV void clear_stage(uint32_t* stage) {
    using WT = WaveTile<G>;
    uint4* d4 = reinterpret_cast<uint4*>(WT::slice(stage));
    for (int i = WT::lane(); i < WT::EPW * WT::SV; i += 64) d4[i] = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_fence();
}

// Copies `rows` (<= EPW) consecutive packed rows of W32 words starting at `src` into this wave's staging slice (pad
// words untouched: clear_stage).  The rows are contiguous in memory, so lane l of iteration `it` simply loads 16-byte
// chunk l + 64 it of the tile — straight-line code: every load is issued (index clamped, never branched around)
// before the first LDS write, one HBM round trip per tile.  bcast: every staged row is a copy of row 0 of src.
template <class G>
PAI_DEV void load_tile_at(uint32_t* dst, const uint32_t* __restrict__ src, int rows, int W32, bool bcast = false) {
    using WT = WaveTile<G>;
    const int lane = WT::lane();
    wave_lds_fence();
    if (!bcast && (W32 & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const int wv = W32 >> 2;                             // 16-byte chunks per row (wave-uniform)
        const int total = rows * wv;
        const uint32_t inv = (65536u + (uint32_t)wv - 1u) / (uint32_t)wv;     // c / wv == (c * inv) >> 16 for c < 1024
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        uint4 v[WT::IT4];
#pragma unroll
        for (int it = 0; it < WT::IT4; ++it) {
            const int c = lane + it * 64;
            v[it] = s4[c < total ? c : total - 1];
        }
#pragma unroll
        for (int it = 0; it < WT::IT4; ++it) {
            const int c = lane + it * 64;
            const int e = (int)(((uint32_t)c * inv) >> 16), k = c - e * wv;
            if (c < total) d4[e * WT::SV + k] = v[it];
        }
    } else {                                                 // broadcast row / odd word counts / unaligned rows
        for (int i = lane; i < WT::EPW * G::SW; i += 64) {
            const int e = i / G::SW, k = i - e * G::SW;
            if (k < W32 && (bcast || e < rows)) dst[i] = src[(size_t)(bcast ? 0 : e) * W32 + k];
        }
    }
    wave_lds_fence();
}

template <class G>
PAI_DEV void load_tile(uint32_t* stage, const uint32_t* __restrict__ src, int rows, int W32, bool bcast = false) {
    load_tile_at<G>(WaveTile<G>::slice(stage), src, rows, W32, bcast);
}
// this lane's limb slice of its element's staged row
template <class G>
PAI_DEV void unpack_row_at(uint32_t (&x)[G::NLL], const uint32_t* row) {
    const int start = RB * G::NLL * G::gl();
    const uint32_t* p = row + (start >> 5);
    const uint32_t s = (uint32_t)start & 31u;
    uint32_t w[G::NWIN + 1];
#pragma unroll
    for (int i = 0; i < G::NWIN; ++i) w[i] = p[i];
    w[G::NWIN] = 0u;
#pragma unroll
    for (int i = 0; i < G::NWIN; ++i) w[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], s);   // bit window now starts at w[0] bit 0
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) {
        const int bit = RB * j, k = bit >> 5, sh = bit & 31;
        x[j] = (sh + RB <= 32 ? (w[k] >> sh) : __builtin_amdgcn_alignbit(w[k + 1], w[k], (uint32_t)sh)) & RMASK;
    }
}

template <class G>
PAI_DEV void unpack_row(uint32_t (&x)[G::NLL], const uint32_t* stage) {
    unpack_row_at<G>(x, stage + G::elem() * G::SW);
}

// limb slices (canonical 29-bit limbs) -> packed u32 words of one element, staged through LDS
// (buffer `lds` of G::LDS_WORDS words, layout [limb][element]).  Bits above 32*W32 must be zero.
template <class G>
PAI_DEV void store_elem(const uint32_t (&x)[G::NLL], uint32_t* __restrict__ row, int W32, uint32_t* lds) {
    const int t = G::gl(), e = G::elem();
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) lds[(G::NLL * t + j) * G::EPB + e] = x[j];
    wave_lds_fence();
    for (int k = t; k < W32; k += G::T) {
        const int j0 = (32 * k) / RB;
        const int s0 = 32 * k - RB * j0;
        uint64_t v = (uint64_t)lds[j0 * G::EPB + e] >> s0;
        if (j0 + 1 < G::NL) v |= (uint64_t)lds[(j0 + 1) * G::EPB + e] << (RB - s0);
        if (j0 + 2 < G::NL) v |= (uint64_t)lds[(j0 + 2) * G::EPB + e] << (2 * RB - s0);
        row[k] = (uint32_t)v;
    }
    wave_lds_fence();
}

// limb slices (canonical 29-bit limbs) -> packed words of the element's staged row: the inverse of unpack_row.  Every
// lane assembles the words its bit window overlaps in registers (compile-time limb positions, then one runtime
// funnel shift by the window's bit offset); the word that straddles two lanes of a group is completed with the piece
// handed over by the previous lane (DPP / bpermute), and the lane writes the words that start inside its window.
// Bits above 32*W32 must be zero.  The caller synchronises, then store_tile writes the rows out coalesced.
template <class G>
PAI_DEV void pack_row_at(const uint32_t (&x)[G::NLL], uint32_t* rowbase) {
    constexpr int NW = G::NWIN;                              // words a window can overlap, both partial ends included
    constexpr int WBITS = RB * G::NLL;
    static_assert(WBITS % 32 >= 2 && NW - 2 == WBITS / 32, "window geometry");
    const int t = G::gl();
    const int start = WBITS * t;
    const int k0 = start >> 5;
    const uint32_t s = (uint32_t)start & 31u;
    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = 0u;
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) {                       // the window with its first bit at w[0] bit 0
        const int bit = RB * j, k = bit >> 5, sh = bit & 31;
        w[k] |= x[j] << sh;
        if (sh + RB > 32) w[k + 1] |= x[j] >> (32 - sh);
    }
    // shift left by s bits across the words (64-bit shifts take the count 32 when s == 0)
#pragma unroll
    for (int i = NW - 1; i >= 1; --i) w[i] = (uint32_t)((((uint64_t)w[i] << 32) | w[i - 1]) >> (32u - s));
    w[0] <<= s;
    uint32_t* row = rowbase + k0;
    wave_lds_fence();
    if constexpr (G::T > 1) {
        // This lane owns the words that START inside its window: own = k0(t + 1) - k0(t) of them (NW - 2 or NW - 1).
        // The word after them straddles into the next lane's window: hand this lane's piece of it over.
        const int own = ((WBITS * (t + 1)) >> 5) - k0;
        const uint32_t top = own == NW - 1 ? w[NW - 1] : w[NW - 2];
        w[0] |= from_prev<G::T>(top);
#pragma unroll
        for (int i = 0; i < NW - 2; ++i) row[i] = w[i];
        if (own == NW - 1) row[NW - 2] = w[NW - 2];
        if (t == G::T - 1) row[own] = top;                  // nobody follows the group's last lane: it keeps its top piece
    } else {
#pragma unroll
        for (int i = 0; i < NW - 1; ++i) row[i] = w[i];      // T == 1: k0 == 0, everything is this lane's
    }
}

template <class G>
PAI_DEV void pack_row(const uint32_t (&x)[G::NLL], uint32_t* stage) {
    pack_row_at<G>(x, stage + G::elem() * G::SW);
}

// this wave's staged rows -> `rows` consecutive packed rows at dst, coalesced full-width stores
template <class G>
PAI_DEV void store_tile_at(const uint32_t* src, uint32_t* __restrict__ dst, int rows, int W32) {
    using WT = WaveTile<G>;
    const int lane = WT::lane();
    wave_lds_fence();
    if ((W32 & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        const int wv = W32 >> 2;
        const int total = rows * wv;
        const uint32_t inv = (65536u + (uint32_t)wv - 1u) / (uint32_t)wv;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
        uint4 v[WT::IT4];
#pragma unroll
        for (int it = 0; it < WT::IT4; ++it) {               // every LDS read first, then the stores
            const int c = lane + it * 64;
            const int cc = c < total ? c : total - 1;
            const int e = (int)(((uint32_t)cc * inv) >> 16), k = cc - e * wv;
            v[it] = s4[e * WT::SV + k];
        }
#pragma unroll
        for (int it = 0; it < WT::IT4; ++it) {
            const int c = lane + it * 64;
            if (c < total) d4[c] = v[it];
        }
    } else {
        for (int i = lane; i < rows * W32; i += 64) {
            const int e = i / W32, k = i - e * W32;
            dst[i] = src[e * G::SW + k];
        }
    }
    wave_lds_fence();
}

template <class G>
PAI_DEV void store_tile(const uint32_t* stage, uint32_t* __restrict__ dst, int rows, int W32) {
    store_tile_at<G>(WaveTile<G>::slice(const_cast<uint32_t*>(stage)), dst, rows, W32);
}

// publish this lane's slice as the element's multiplier operand b in LDS
template <class G>
PAI_DEV void stage_b(const uint32_t (&x)[G::NLL], uint32_t* lds) {
    const int t = G::gl(), e = G::elem();
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) lds[(G::NLL * t + j) * G::EPB + e] = x[j];
    wave_lds_fence();
}

// r = a * b mod-Montgomery with b taken from the element's LDS column
// (minus-one geometries: nm is the (M + 1) / 2^(29 U) slice and the scalar next to it the number of row blocks)
template <class G>
PAI_DEV void mm_lds(uint32_t (&r)[G::NLL], const uint32_t (&a)[G::NLL], const uint32_t* lds,
                    const typename G::NM& nm, uint32_t n0inv) {
    if constexpr (G::M1) mont_mul_m1<G::NLL, G::U, G::T>(r, a, lds + G::elem(), G::EPB, nm, (int)n0inv);
    else mont_mul<G::NLL, G::U, G::T>(r, a, lds + G::elem(), G::EPB, nm, n0inv);
}

// x = x * x (Montgomery) : stage x as b, multiply by itself
template <class G>
PAI_DEV void mm_square(uint32_t (&x)[G::NLL], uint32_t* lds, const typename G::NM& nm, uint32_t n0inv) {
    stage_b<G>(x, lds);
    uint32_t r[G::NLL];
    mm_lds<G>(r, x, lds, nm, n0inv);
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
}

// x = x * y (Montgomery), y in registers: x is staged as b, y is the register operand
template <class G>
PAI_DEV void mm_times(uint32_t (&x)[G::NLL], const uint32_t (&y)[G::NLL], uint32_t* lds,
                      const typename G::NM& nm, uint32_t n0inv) {
    stage_b<G>(x, lds);
    uint32_t r[G::NLL];
    mm_lds<G>(r, y, lds, nm, n0inv);
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) x[j] = r[j];
}

// Minus-one geometries: x (a residue modulo M k, x <= M k < 2^(29 (m + U)), m the limbs of M) -> the canonical residue
// modulo M itself.  The quotient x / M is SHORT (k < 2^(29 U)): Barrett on the top U + 3 limbs of x with
// mu = floor(2^(29 (m + U + 1)) / M) gives q^ in {q - 2 .. q} as U + 1 limbs, x - q^ M costs U + 1 products per limb, and two
// conditional subtractions finish.  (Rounds 3-5 did this by two conventional Montgomery products with M's own context —
// 2 NL rows with a dependent quotient digit each: the larger part of the exit of every small-batch kernel.)
template <class G>
PAI_DEV void m1_reduce_to_true_modulus(uint32_t (&x)[G::NLL], uint32_t* lds, const MontCtx* __restrict__ f) {
    constexpr int NLL = G::NLL, U = G::U, T = G::T, S = U + 3;
    static_assert(U <= NLL && S <= 12, "window of the modulus from one neighbour lane; mu[] of the context");
    NmRegs<NLL> nf;
    load_const_slice<G>(nf.v, f->n);
    const int m = (int)f->mlimbs;
    stage_b<G>(x, lds);
    uint32_t xt[S];
#pragma unroll
    for (int a = 0; a < S; ++a) {
        const int idx = m - 2 + a;
        xt[a] = idx < G::NL ? lds[idx * G::EPB + G::elem()] : 0u;
    }
    wave_lds_fence();
    uint64_t col[2 * S];
#pragma unroll
    for (int k = 0; k < 2 * S; ++k) col[k] = 0;
#pragma unroll
    for (int b = 0; b < S; ++b) {
        const uint32_t mb = f->mu[b];
#pragma unroll
        for (int a = 0; a < S; ++a) col[a + b] += (uint64_t)xt[a] * mb;
    }
    uint64_t c = 0;
    uint32_t q[U + 1];
#pragma unroll
    for (int k = 0; k < S + U + 1; ++k) {
        const uint64_t t = col[k] + c;
        if (k >= S) q[k - S] = (uint32_t)t & RMASK;
        c = t >> RB;
    }
    // this lane's limbs of q^ M need the U limbs of M below its own: the neighbour's top ones
    uint32_t w[NLL + U];
#pragma unroll
    for (int j = 0; j < NLL; ++j) w[U + j] = nf.v[j];
#pragma unroll
    for (int t = 0; t < U; ++t) w[U - 1 - t] = T > 1 ? from_prev<T>(nf.v[NLL - 1 - t]) : 0u;
    int64_t d[NLL];
#pragma unroll
    for (int j = 0; j < NLL; ++j) {
        uint64_t p = 0;
#pragma unroll
        for (int i = 0; i <= U; ++i) p += (uint64_t)q[i] * w[U + j - i];
        d[j] = (int64_t)x[j] - (int64_t)p;
    }
    Rows<NLL, U, T>::finish_signed(d, x);
    cond_sub<NLL, T>(x, nf);
    cond_sub<NLL, T>(x, nf);
}

// plain integer 1 spread over the group (lane 0, limb 0)
template <class G>
PAI_DEV void set_plain_one(uint32_t (&x)[G::NLL]) {
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) x[j] = 0;
    if (G::gl() == 0) x[0] = 1;
}

// w bits of a little-endian u32 exponent starting at bit `pos` (uniform operands => scalar code)
PAI_DEV uint32_t exp_bits(const uint32_t* __restrict__ e, int ewords, int pos, int w) {
    const int k = pos >> 5, s = pos & 31;
    uint64_t v = (k < ewords) ? e[k] : 0u;
    if (k + 1 < ewords) v |= (uint64_t)e[k + 1] << 32;
    return (uint32_t)(v >> s) & ((1u << w) - 1u);
}

}  // namespace pai
