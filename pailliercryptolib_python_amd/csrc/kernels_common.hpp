// Shared device helpers for the Paillier kernels: element I/O (packed u32 words <-> radix-2^29
// limb slices spread over a lane group), LDS staging of the multiplier operand, uniform exponent
// window extraction.  See mont_dev.hpp for the arithmetic itself.
#pragma once
#include <type_traits>
#include "mont_dev.hpp"

namespace pai {

constexpr int BLOCK_THREADS = 256;
constexpr int MODMUL_FULL = 0, MODMUL_MONT = 1;      // k_modmul modes (kernels_modexp.hpp)

// Geometry of one kernel instance: NLL limbs per lane, T lanes per element, U rows per block,
// NMLDS = the modulus slice is re-read from LDS during the q*n step instead of living in VGPRs.
template <int NLL_, int T_, int U_, bool NMLDS_ = false>
struct Geo {
    static constexpr int NLL = NLL_, T = T_, U = U_;
    static constexpr bool NMLDS = NMLDS_;
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
#pragma unroll
        for (int j = 0; j < G::NLL; ++j) nm.v[j] = ctx->n[G::NLL * t + j];
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
// which is what bounded k_modmul in round 1).  Kernels whose arithmetic per tile is short use this pair instead:
// the workgroup copies the tile's rows HBM -> LDS with full-width coalesced loads, then every lane funnel-shifts
// its own bit window out of LDS (one runtime shift per lane, compile-time limb positions after it).

// Cooperative copy of `rows` consecutive packed rows (W32 words each, starting at `src`) into the staging area,
// row stride G::SW, zero beyond W32 / beyond `rows`.  All threads of the block must call it; the caller
// synchronises before unpacking.  bcast: every staged row is a copy of row 0 of src.
template <class G>
PAI_DEV void load_tile(uint32_t* stage, const uint32_t* __restrict__ src, int rows, int W32, bool bcast = false) {
    if ((W32 & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        constexpr int SV = G::SW / 4;
        const int wv = W32 >> 2;
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(stage);
        for (int i = threadIdx.x; i < G::EPB * SV; i += BLOCK_THREADS) {
            const int e = i / SV, k = i - e * SV;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (k < wv && (bcast || e < rows)) v = s4[(size_t)(bcast ? 0 : e) * wv + k];
            d4[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < G::EPB * G::SW; i += BLOCK_THREADS) {
            const int e = i / G::SW, k = i - e * G::SW;
            uint32_t v = 0u;
            if (k < W32 && (bcast || e < rows)) v = src[(size_t)(bcast ? 0 : e) * W32 + k];
            stage[i] = v;
        }
    }
}

// this lane's limb slice of its element's staged row
template <class G>
PAI_DEV void unpack_row(uint32_t (&x)[G::NLL], const uint32_t* stage) {
    const int start = RB * G::NLL * G::gl();
    const uint32_t* p = stage + G::elem() * G::SW + (start >> 5);
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

// limb slices -> packed words of the element's staged row (through the [limb][element] buffer `lds`, as store_elem);
// the caller synchronises, then store_tile writes the rows out with coalesced full-width stores
template <class G>
PAI_DEV void pack_row(const uint32_t (&x)[G::NLL], uint32_t* stage, int W32, uint32_t* lds) {
    const int t = G::gl(), e = G::elem();
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) lds[(G::NLL * t + j) * G::EPB + e] = x[j];
    wave_lds_fence();
    uint32_t* row = stage + e * G::SW;
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

template <class G>
PAI_DEV void store_tile(const uint32_t* stage, uint32_t* __restrict__ dst, int rows, int W32) {
    if ((W32 & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        constexpr int SV = G::SW / 4;
        const int wv = W32 >> 2;
        const uint4* s4 = reinterpret_cast<const uint4*>(stage);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
        for (int i = threadIdx.x; i < rows * wv; i += BLOCK_THREADS) {
            const int e = i / wv, k = i - e * wv;
            d4[i] = s4[e * SV + k];
        }
    } else {
        for (int i = threadIdx.x; i < rows * W32; i += BLOCK_THREADS) {
            const int e = i / W32, k = i - e * W32;
            dst[i] = stage[e * G::SW + k];
        }
    }
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
template <class G>
PAI_DEV void mm_lds(uint32_t (&r)[G::NLL], const uint32_t (&a)[G::NLL], const uint32_t* lds,
                    const typename G::NM& nm, uint32_t n0inv) {
    mont_mul<G::NLL, G::U, G::T>(r, a, lds + G::elem(), G::EPB, nm, n0inv);
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
