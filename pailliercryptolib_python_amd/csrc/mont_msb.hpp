// Most-significant-limb-first interleaved modular product on the lane-group engine: r = a * b mod M for operands in
// WIRE form (plain residues) in ONE pass of 2 NL^2 limb products — the cost of one Montgomery product, where the
// Montgomery route needs two (a b R^-1, then * R^2 R^-1) because neither operand carries the factor R.
// Serves CipherText::operator+ (bindings/ipcl_bindings_classes.cpp:318-321) on ciphertexts that arrive as wire words.
// The cell-exact model with the bounds below asserted: tools/msb_model.py.
//
// The mirror image of Rows::block (mont_dev.hpp): the multiplier's limbs are taken from the top, the window slides UP
// by one column per row (acc <- acc B + a b_i), and the quotient digit comes from the TOP of the accumulator instead of
// the bottom:  q^ = floor(V mu / 2^63),  V = the accumulator's highest ~62 bits,  mu = floor(2^(P+31) / Mt).
//   * Mt = M B^off puts the modulus' top limb at limb NL - 1 for every key size (compile-time cell positions); the
//     multiplier enters as b B^off (rows below `off` read zero) and the result leaves as r B^off — a limb shift on the way out.
//   * the reduction ADDS q^ W, W = B^NL - Mt, so every cell stays an unsigned lazy 64-bit column — and the q^ B^NL this leaves
//     on top of acc - q^ Mt needs NO correction: it sits at aligned column NL, the next row moves it to NL + 1, and with tb =
//     the bits of Mt in limb NL - 1 (3 <= tb <= 26, MsbCtx::tb) every column from NL + 1 up weighs a multiple of 2^(P+32) while
//     the accumulator proper is < 2 Mt < 2^(P+1): the digit estimate (computed modulo 2^64 from cells whose weights it knows)
//     never sees it, the slide drops it, and the final carry sweep masks it off the top limb.
//   * q^ is never above the true digit (all roundings go down; a window that reads negative because the lazy columns below
//     it still hold the carries gives 0) and at most 1 below it: by induction 0 <= acc < 2 Mt (then acc B + a b_i <
//     (2 B + 4) Mt for any a of the row's word size, off >= 1; V < 2^62 (1 + 2^-28), and the roundings — mu's, < V / 2^63, the
//     columns left out, < 2^-24 — stay below 0.51), so q^ <= 2 B + 4 and a column gains < 3 x 2^58 per row: up to 20 rows
//     between carry normalisations (Montgomery rows: 24).  A normalisation runs BEFORE the hand-over to the next lane and
//     splits the top cell too, so that the cell the next lane receives is as small as its own.
//   * V needs the top FOUR cells (the lazy carries of the fourth still weigh 2^(35-29-tb+32) quotient units / 2^32).
#pragma once
#include "mont_dev.hpp"

namespace pai {

constexpr int MSB_OFF_MAX = 16;             // limbs the modulus may be shifted up by (zero limbs behind k_modmul_msb's operand buffer)
constexpr int MSB_NORM_ROWS = 20;           // rows between carry normalisations: the largest multiple of U below (3 x 2^58 per row and column)

struct MsbCtx {
    uint32_t w[NLMAX];        // B^NL - Mt
    uint32_t mt[NLMAX];       // Mt = M B^off
    uint32_t mu;              // floor(2^(P + 31) / Mt), P = bit length of Mt
    uint32_t tb;              // P - 29 (NL - 1)
    uint32_t off;             // limbs M was shifted up by (>= 1)
    uint32_t nl;
    uint32_t m2, eight;       // 2^(32 - tb), 8: multipliers of v_mad_u64_u32 the compiler must not see as literals (MsbK)
};

// uniform constants of the digit estimate (eight / m2 are run-time values on purpose: as literals the compiler turns the
// multiply-adds that carry a free 64-bit addition into shift / mask / add sequences three times as long)
struct MsbK {
    uint32_t mu, sh1 /* tb - 3 */, sh2 /* 32 - tb */, sh3 /* 29 - tb */, m2 /* 2^(32 - tb) */, eight;
};

// value held by the LAST lane of the caller's group
template <int T> PAI_DEV uint32_t bcast_last(uint32_t v) {
    if constexpr (T == 1) return v;
    else if constexpr (T == 2) return dpp_mov<0xF5>(v);          // quad_perm [1,1,3,3]
    else if constexpr (T == 4) return dpp_mov<0xFF>(v);          // quad_perm [3,3,3,3]
    else {
        static_assert(T == 8, "lane groups of up to 8");
        const uint32_t q = dpp_mov<0xFF>(v);                     // lanes 4-7 of each group of 8 now hold lane 7's value ...
        return (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x104, 0xF, 0x5, false);   // ... lanes 0-3 fetch it: row_shl:4, banks 0 and 2
    }
}
// from_prev without the zero for the group's first lane (the caller masks what it adds)
template <int T> PAI_DEV uint32_t from_prev_raw(uint32_t v) {
    if constexpr (T == 2) return dpp_mov<0xA0>(v);
    else return dpp_mov<0x111>(v);                               // row_shr:1
}

PAI_DEV uint32_t msb_digit(uint64_t c3, uint64_t c2, uint64_t c1, uint64_t c0, const MsbK& k) {
    const uint64_t x1 = (uint64_t)(uint32_t)(c0 >> 32) * k.eight + c1;              // c1 + c0 / B (c0's low 3 carry bits dropped)
    const uint64_t x2 = x1 >> k.sh1;
    const uint64_t v = (uint64_t)(uint32_t)c2 * k.m2 + x2;
    const uint32_t vh = (uint32_t)(v >> 32) + ((uint32_t)(c2 >> 32) << k.sh2) + ((uint32_t)c3 << k.sh3);
    const uint64_t t = (uint64_t)vh * k.mu + __umulhi((uint32_t)v, k.mu);
    const uint32_t q = __builtin_amdgcn_alignbit((uint32_t)(t >> 32), (uint32_t)t, 31);
    return (int32_t)vh < 0 ? 0u : q;
}

template <int NLL, int U, int T>
struct RowsMsb {
    static constexpr int NW = NLL + U;
    static constexpr int NL = NLL * T;
    static constexpr int NORM_BLOCKS = MSB_NORM_ROWS / U;
    static_assert(NLL >= 4 && NL % U == 0 && T <= 8 && NORM_BLOCKS >= 1, "geometry");

    // U rows (multiplier limbs bv[0] = the highest).  Row u works at cell offset U - 1 - u: aligned column c of lane t is cell
    // c - t NLL + U - 1 - u.
    template <class NM>
    PAI_DEV static void rows(uint64_t (&acc)[NW], const uint32_t (&a)[NLL], const uint32_t (&bv)[U], const NM& wm, const MsbK& k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = U - 1 - u;
#pragma unroll
            for (int j = NLL - 1; j >= 0; --j) acc[j + o] += (uint64_t)a[j] * bv[u];
            const uint32_t q = bcast_last<T>(msb_digit(acc[o + NLL], acc[o + NLL - 1], acc[o + NLL - 2], acc[o + NLL - 3], k));
            wm.template mac<NLL>(acc, o, q);
        }
    }
    // carry-save normalisation of the WHOLE window, the top cell included: its carry (returned) belongs to the next lane's cell U.
    // Done before the hand-over, so that a cell that changes lanes is as small as one that stays.
    PAI_DEV static uint64_t normalize(uint64_t (&acc)[NW]) {
        const uint64_t top = acc[NW - 1] + (acc[NW - 2] >> RB);
#pragma unroll
        for (int j = NW - 2; j >= 1; --j) acc[j] = (acc[j] & RMASK) + (acc[j - 1] >> RB);
        acc[0] &= RMASK;
        acc[NW - 1] = top & RMASK;
        return top >> RB;
    }
    // the U cells above the lane's own range (and the normalisation's carry, if any) go to the next lane; the window moves up by U
    // acc += nf01 * (hi : lo) (nf01 = 0 in the group's first lane, which has no previous lane, else 1) without building a register
    // pair: v_mad_u64_u32 adds a zero-extended word times nf01 for free, the high word takes one masked addition
    PAI_DEV static void add64(uint64_t& acc, uint32_t lo, uint32_t hi, uint32_t nf01) {
        const uint64_t t = (uint64_t)lo * nf01 + acc;
        acc = ((uint64_t)((uint32_t)(t >> 32) + (hi & (0u - nf01))) << 32) | (uint32_t)t;
    }
    template <bool CARRY>
    PAI_DEV static void handover_slide(uint64_t (&acc)[NW], uint64_t carry, uint32_t nf01) {
        if constexpr (T > 1) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                add64(acc[u], from_prev_raw<T>((uint32_t)acc[NLL + u]), from_prev_raw<T>((uint32_t)(acc[NLL + u] >> 32)), nf01);
            if constexpr (CARRY) add64(acc[U], from_prev_raw<T>((uint32_t)carry), from_prev_raw<T>((uint32_t)(carry >> 32)), nf01);
        }
#pragma unroll
        for (int j = NLL - 1; j >= 0; --j) acc[j + U] = acc[j];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0;
    }
};

// r = a * bs mod Mt with bs = b B^off staged by the caller, 0 <= r < 2 Mt, canonical 29-bit limbs.  bs's limbs are read from
// b_ptr[i * bstride] (LDS, [limb][element]); wm = this lane's slice of MsbCtx::w.
template <int NLL, int U, int T, class NM>
PAI_DEV void msb_mul(uint32_t (&r)[NLL], const uint32_t (&a)[NLL], const uint32_t* b_ptr, int bstride, const NM& wm, const MsbK& k) {
    using RW = RowsMsb<NLL, U, T>;
    uint64_t acc[RW::NW];
#pragma unroll
    for (int j = 0; j < RW::NW; ++j) acc[j] = 0;
    constexpr int NB = RW::NL / U;
    const uint32_t nf01 = group_lane<T>() != 0 ? 1u : 0u;
    int since = 0;
#pragma unroll 1
    for (int blk = 0; blk < NB; ++blk) {
        uint32_t bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = b_ptr[(RW::NL - 1 - (blk * U + u)) * bstride];
        NM wb = wm;
        if constexpr (std::is_same<NM, NmLds<NLL>>::value) {
            // W's slice is re-read from LDS in every block: without this the loads are hoisted out of the loop as invariants and
            // the 36 registers they were to free are back (spilled)
            int z = 0;                                          // (an opaque zero: the pointer itself would lose its LDS address space)
            asm volatile("" : "+v"(z));
            wb.p = wm.p + z;
        }
        RW::rows(acc, a, bv, wb, k);
        if (++since == RW::NORM_BLOCKS && blk != NB - 1) {
            since = 0;
            const uint64_t carry = RW::normalize(acc);
            RW::template handover_slide<true>(acc, carry, nf01);
        } else {
            RW::template handover_slide<false>(acc, 0, nf01);
        }
    }
    // the last block slid too: the own cells are [U, NLL + U)
    uint64_t fin[RW::NW];
#pragma unroll
    for (int j = 0; j < RW::NW; ++j) fin[j] = j < NLL ? acc[j + U] : 0;
    Rows<NLL, U, T>::finish(fin, r);
}

}  // namespace pai
