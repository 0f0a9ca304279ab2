// Device-side multi-precision Montgomery arithmetic for gfx950 (CDNA4).
//
// This is the replacement for the reference's un-vendored IPP-Crypto kernel mbx_exp{1024..4096}_mb8
// (8 moduli in 8 AVX-512 lanes, 52-bit IFMA limbs; named at README.md:32, reached through
// ipcl::modExp from bindings/ipcl_bindings_classes.cpp:57,130,325).  The CDNA4 mapping is different
// on purpose:
//
//  * A big integer of NL = T*NLL radix-2^29 limbs is held by a *group of T adjacent lanes*
//    (T = 1,2,4,8; NLL = 36 or 27 limbs per lane), so a wavefront carries 64/T integers.  The
//    O(NL^2) inner loops are pure v_mad_u64_u32 on compile-time register indices (measured on
//    MI355X: half rate, 4 cycles per wave64 => 36 T MAC32/s per GPU —
//    profiles/r01/ubench_valu_mi355x.jsonl).  gfx950 VALU operands are limited to 256 VGPRs per
//    lane, which is what fixes NLL: a[NLL] + n[NLL] + a 2*(NLL+U)-register accumulator window must
//    fit with room for 2-3 waves per SIMD.
//  * radix 2^29 limbs in 32-bit VGPRs with *lazy* 64-bit column accumulators: a product is < 2^58,
//    so ~30 rows can be accumulated with plain v_mad_u64_u32 (which adds a 64-bit addend for free)
//    before a carry normalisation.  On gfx950 every carry instruction (v_add_co/v_addc_co,
//    v_lshl_add_u64) costs as much as a multiply, so a radix-2^32 CIOS would spend half its issue
//    slots on carries.
//  * the multiplier limbs b[i] stream from LDS, layout [limb][element] (conflict-free, broadcast
//    inside a lane group), one ds_read per row; the quotient digit q_i is computed by the group's
//    lane 0 and broadcast; the limb that leaves the bottom of lane t's window is handed to lane
//    t-1's top once per block of U rows.
//  * rows are processed U at a time with register renaming; the accumulator window is shifted down
//    by U limbs once per block and carry-normalised every NORM_ROWS rows.
//
// Montgomery domain: R = 2^(29*NL) with R > 4M, so operands stay in [0, 2M) and no conditional
// subtraction is needed between multiplications (Walter).  Values are canonicalised once at the
// end of an operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pai {

constexpr int RB = 29;                          // radix bits
constexpr uint32_t RMASK = (1u << RB) - 1u;
constexpr int NORM_ROWS = 24;                   // rows between carry normalisations (must stay < 30)

#define PAI_DEV __device__ __forceinline__

// Per-modulus constants in device memory (all limbs radix 2^29, zero-padded to NLMAX).
constexpr int NLMAX = 288;                      // 8192-bit moduli (+2 bits) => 283 limbs, padded
struct MontCtx {
    uint32_t n[NLMAX];        // modulus M
    uint32_t r2[NLMAX];       // R^2 mod M
    uint32_t one[NLMAX];      // R mod M
    uint32_t n0inv;           // -M^-1 mod 2^29
    uint32_t nl;              // limbs in use (template instance must match)
    uint32_t bits;            // bit length of M
    uint32_t rows;            // "minus-one" contexts (Rows::block_m1): rows per product, R = 2^(29 rows); else 0
    uint32_t npp[NLMAX];      // "minus-one" contexts: (M + 1) / 2^(29 U), the multiplier of the quotient digits
    uint32_t mlimbs;          // conventional contexts: m = radix-2^29 limbs of M ...
    uint32_t mu[12];          // ... and floor(2^(29 (m + U + 1)) / M), U the geometry's rows per block (m1_reduce_to_true_modulus)
};

// ---- lane-group helpers -------------------------------------------------------------------------
template <int T> PAI_DEV int group_lane() { return (int)(threadIdx.x & (T - 1)); }

// Cross-lane moves inside a lane group: DPP (a VALU move, no LDS pipe, short latency) for every group size.
//   quad_perm control = p0 | p1<<2 | p2<<4 | p3<<6 ; row_shl:1 = 0x101 ; row_shr:1 = 0x111
template <int CTRL> PAI_DEV uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);   // bound_ctrl: missing source -> 0
}
// value held by lane 0 of the caller's group
template <int T> PAI_DEV uint32_t bcast0(uint32_t v) {
    if constexpr (T == 1) return v;
    else if constexpr (T == 2) return dpp_mov<0xA0>(v);          // quad_perm [0,0,2,2]
    else if constexpr (T == 4) return dpp_mov<0x00>(v);          // quad_perm [0,0,0,0]
    else if constexpr (T == 8) {
        // two DPP moves instead of ds_bpermute (one ds_swizzle_b32 measured no better: profiles/r04/keysize_bcast8_swizzle.jsonl):
        // lane 0 / 4 of each quad pair, then the upper quad fetches from 4 lanes below
        // (groups of 8 are aligned halves of a 16-lane DPP row); the second move writes banks 1 and 3 only (lanes 4-7, 12-15),
        // the other lanes keep q: no select
        const uint32_t q = dpp_mov<0x00>(v);
        return (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x114, 0xF, 0xA, false);   // row_shr:4, bank_mask 0b1010
    } else if constexpr (T == 64) return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);      // through an SGPR: no LDS-pipe latency
    else if constexpr (T == 32) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
        return (threadIdx.x & 32) ? hi : lo;
    } else if constexpr (T == 16) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
        const uint32_t lo = (threadIdx.x & 16) ? b : a, hi = (threadIdx.x & 16) ? d : c;
        return (threadIdx.x & 32) ? hi : lo;
    } else return (uint32_t)__shfl((int)v, (int)((threadIdx.x & 63) & ~(T - 1)), 64);
}
// value held by the next lane of the group (lane T-1 receives 0)
template <int T> PAI_DEV uint32_t from_next(uint32_t v) {
    if constexpr (T == 1) return 0u;
    else {
        uint32_t r;
        if constexpr (T == 2) r = dpp_mov<0xF5>(v);              // quad_perm [1,1,3,3]
        else if constexpr (T == 4 || T == 8) r = dpp_mov<0x101>(v);   // row_shl:1  (lane i <- lane i+1 inside a 16-lane row)
        else if constexpr (T >= 16) r = dpp_mov<0x130>(v);       // wave_shl:1 (lane i <- lane i+1 across the whole wave)
        else r = (uint32_t)__shfl_down((int)v, 1, 64);
        if constexpr (T == 64) return r;                         // bound_ctrl: the wave's last lane already reads 0
        return (group_lane<T>() == T - 1) ? 0u : r;
    }
}
// from_next for values that are ZERO in every group's lane 0 (the limb a Montgomery row retires: the quotient digit makes it
// so): lane T-1 then reads the next group's zero (or the DPP row's / wave's edge, bound_ctrl: 0) and needs no select
template <int T> PAI_DEV uint32_t from_next_z(uint32_t v) {
    if constexpr (T == 4 || T == 8) return dpp_mov<0x101>(v);    // row_shl:1
    else if constexpr (T >= 16) return dpp_mov<0x130>(v);        // wave_shl:1
    else return from_next<T>(v);
}
// value held by the previous lane of the group (lane 0 receives 0)
template <int T> PAI_DEV uint32_t from_prev(uint32_t v) {
    if constexpr (T == 1) return 0u;
    else {
        uint32_t r;
        if constexpr (T == 2) r = dpp_mov<0xA0>(v);              // quad_perm [0,0,2,2]
        else if constexpr (T == 4 || T == 8) r = dpp_mov<0x111>(v);   // row_shr:1  (lane i <- lane i-1 inside a 16-lane row)
        else if constexpr (T >= 16) r = dpp_mov<0x138>(v);       // wave_shr:1 (lane i <- lane i-1 across the whole wave)
        else r = (uint32_t)__shfl_up((int)v, 1, 64);
        if constexpr (T == 64) return r;                         // bound_ctrl: lane 0 already reads 0
        return (group_lane<T>() == 0) ? 0u : r;
    }
}
template <int T> PAI_DEV bool group_any(bool p) {
    if constexpr (T == 1) return p;
    else {
        unsigned long long m = __ballot(p);
        if constexpr (T == 64) return m != 0ull;
        else {
            const int base = (threadIdx.x & 63) & ~(T - 1);
            return ((m >> base) & ((1ull << T) - 1ull)) != 0ull;
        }
    }
}
// orders this wave's LDS writes before its later LDS reads (lane groups never span waves)
PAI_DEV void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for this wave's global
// stores / loads in flight (vmcnt), so a tile's stores drain behind the next tile's loads
PAI_DEV void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- the row engine ---------------------------------------------------------------------------
// Each lane owns NLL limbs; acc is its window of NLL+U lazy 64-bit columns.  Row i adds
// a[j]*b_i (+ q_i*n[j]) into columns j, retires column 0 and slides the window.
template <int NLL, int U, int T>
struct Rows {
    static constexpr int NW = NLL + U;
    static constexpr int NL = NLL * T;

    PAI_DEV static void zero(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[j] = 0;
    }

    // parallel carry-save normalisation of the window: afterwards every column < 2^29 + 2^35
    PAI_DEV static void normalize(uint64_t (&acc)[NW]) {
#pragma unroll
        for (int j = NW - 1; j >= 1; --j) {
            uint64_t keep = (j == NW - 1) ? acc[j] : (acc[j] & RMASK);
            acc[j] = keep + (acc[j - 1] >> RB);
        }
        acc[0] &= RMASK;
    }

    // U Montgomery rows, then the window moves down by U columns.
    // low[u] receives the 29-bit limb retired by row u (meaningful in group lane 0: it is the next
    // output limb of a plain product; it is zero when QN).
    template <bool AB, bool QN, class NM>
    PAI_DEV static void block(uint64_t (&acc)[NW], const uint32_t (&a)[NLL], const uint32_t (&bv)[U],
                              const NM& nm, uint32_t n0inv, uint32_t (&low)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (AB) {
#pragma unroll
                for (int j = 0; j < NLL; ++j) acc[j + u] += (uint64_t)a[j] * bv[u];
            }
            if constexpr (QN) {
                const uint32_t q = bcast0<T>(((uint32_t)acc[u] * n0inv) & RMASK);
                nm.template mac<NLL>(acc, u, q);
            }
            acc[u + 1] += acc[u] >> RB;
            low[u] = (uint32_t)acc[u] & RMASK;     // zero in group lane 0 when QN
        }
        // hand the retired low parts to the previous lane's top columns, then slide the window
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc[j] = acc[j + U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (T > 1) acc[NLL - U + u] += (uint64_t)(QN ? from_next_z<T>(low[u]) : from_next<T>(low[u]));
            acc[NLL + u] = 0;
        }
    }

    // U rows for a modulus M == -1 (mod 2^(29 U)) ("minus-one" contexts: the caller's modulus times -M^-1 mod 2^(29 U)):
    // T + q M with q = T mod 2^(29 U) is T - q + q (M + 1), so the U quotient digits of a block ARE the U limbs the
    // block retires — no multiplication by -M^-1 and, what matters on a lone wave, no dependency of row u + 1's digit on
    // row u's multiply-accumulates: all a*b of the block, one short carry chain, the broadcasts, all q*(M + 1)/2^(29 U).
    // The wide-group (latency) geometries spend 80 % of their instructions outside the multiplier; this halves them.
    template <class NM>
    PAI_DEV static void block_m1(uint64_t (&acc)[NW], const uint32_t (&a)[NLL], const uint32_t (&bv)[U], const NM& npp) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NLL; ++j) acc[j + u] += (uint64_t)a[j] * bv[u];
        }
        uint32_t low[U], q[U];
        uint64_t c = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t t = acc[u] + c;
            low[u] = (uint32_t)t & RMASK;
            c = t >> RB;
        }
        acc[U] += c;
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = bcast0<T>(low[u]);            // group lane 0 retires the quotient digits
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc[j] = acc[j + U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[NLL + u] = 0;
            if constexpr (T > 1) acc[NLL - U + u] += (uint64_t)from_next<T>(low[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) npp.template mac<NLL>(acc, u, q[u]);
    }

    // full carry propagation into canonical 29-bit limbs (across the T lanes of the group)
    PAI_DEV static void finish(const uint64_t (&acc)[NW], uint32_t (&r)[NLL]) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NLL; ++j) {
            uint64_t t = acc[j] + c;
            r[j] = (uint32_t)t & RMASK;
            c = t >> RB;
        }
        if constexpr (T > 1) {
            // c (< 2^36) belongs to the next lane's limb 0.  Each pass adds the incoming carry and
            // re-propagates locally; after the first pass carries are 0/1 and die out quickly.
            uint32_t cin_lo = from_prev<T>((uint32_t)c), cin_hi = from_prev<T>((uint32_t)(c >> 32));
            uint64_t cin = ((uint64_t)cin_hi << 32) | cin_lo;
            while (__any(cin != 0)) {
                uint64_t cc = cin;
                // the incoming carry (< 2^36) normally dies within the first two limbs: ripple in
                // groups of four limbs and stop as soon as no lane of the wave has a carry left
#pragma unroll
                for (int j0 = 0; j0 < NLL; j0 += 4) {
#pragma unroll
                    for (int j = j0; j < j0 + 4 && j < NLL; ++j) {
                        uint64_t t = (uint64_t)r[j] + cc;
                        r[j] = (uint32_t)t & RMASK;
                        cc = t >> RB;
                    }
                    if (!__any(cc != 0)) break;
                }
                cin = (uint64_t)from_prev<T>((uint32_t)cc);   // 0 or 1 now
            }
        }
    }

    // Same for signed per-limb values (|s[j]| < 2^62) whose total is known to be non-negative and
    // to fit in 29*NL bits: used for limb-wise differences such as mq - mp + q.
    PAI_DEV static void finish_signed(const int64_t (&s)[NLL], uint32_t (&r)[NLL]) {
        int64_t c = 0;
#pragma unroll
        for (int j = 0; j < NLL; ++j) {
            int64_t t = s[j] + c;
            r[j] = (uint32_t)t & RMASK;
            c = t >> RB;                       // arithmetic shift: floor division
        }
        if constexpr (T > 1) {
            uint32_t cin_lo = from_prev<T>((uint32_t)c), cin_hi = from_prev<T>((uint32_t)((uint64_t)c >> 32));
            int64_t cin = (int64_t)(((uint64_t)cin_hi << 32) | cin_lo);
            while (__any(cin != 0)) {
                int64_t cc = cin;
#pragma unroll
                for (int j = 0; j < NLL; ++j) {
                    int64_t t = (int64_t)r[j] + cc;
                    r[j] = (uint32_t)t & RMASK;
                    cc = t >> RB;
                }
                uint32_t lo2 = from_prev<T>((uint32_t)cc), hi2 = from_prev<T>((uint32_t)((uint64_t)cc >> 32));
                cin = (int64_t)(((uint64_t)hi2 << 32) | lo2);
            }
        }
    }
};

// Where the lane's slice of the modulus comes from during the q*n step.
//  NmRegs: NLL VGPRs (or SGPRs when T == 1 and the compiler proves uniformity).
//  NmLds : read back from LDS ([lane-slice][limb], 16-byte aligned) four limbs at a time — frees
//          NLL registers at the price of NLL/4 ds_read_b128 per row.
template <int NLL>
struct NmRegs {
    uint32_t v[NLL];
    template <int N, int NW>
    PAI_DEV void mac(uint64_t (&acc)[NW], int u, uint32_t q) const {
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j + u] += (uint64_t)v[j] * q;
    }
    PAI_DEV uint32_t limb(int j) const { return v[j]; }
};
template <int NLL>
struct NmLds {
    const uint32_t* p;      // this lane's slice in LDS
    template <int N, int NW>
    PAI_DEV void mac(uint64_t (&acc)[NW], int u, uint32_t q) const {
        static_assert(N % 4 == 0, "NLL must be a multiple of 4 for vector LDS reads");
        const uint4* p4 = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int j = 0; j < N; j += 4) {
            const uint4 w = p4[j / 4];
            acc[j + 0 + u] += (uint64_t)w.x * q;
            acc[j + 1 + u] += (uint64_t)w.y * q;
            acc[j + 2 + u] += (uint64_t)w.z * q;
            acc[j + 3 + u] += (uint64_t)w.w * q;
        }
    }
    PAI_DEV uint32_t limb(int j) const { return p[j]; }
};

// Hook called once per row block of a product (pair_mul / mont_mul_pf): lets a kernel stream the NEXT operand from HBM
// into LDS a few words per block, through a register or two, while the multiplier runs (RowStream below).
struct NoStream {
    PAI_DEV void step(int) {}
    PAI_DEV void drain() {}
};

// Streams `NCH` chunks of CH words (this lane's contiguous slice of a row in global memory, CH * 4-byte aligned) into LDS
// column storage dst[w * stride] for word w of the slice: block k of the running product issues the load of chunk k and
// stores chunk k - 1, so a chunk has a whole row block (thousands of cycles) to arrive and occupies CH registers.
// Two consecutive slices (digit pairs: the c part then the d part) are served by the split at chunk NCH0.
template <int CH, int NCH0, int NCH1>
struct RowStream {
    static_assert(CH == 1 || CH == 2 || CH == 4, "chunk width");
    const uint32_t* src0;
    const uint32_t* src1;
    uint32_t* dst0;
    uint32_t* dst1;
    int stride;
    bool on = true;            // wave-uniform: a product that has nothing to stream
    uint32_t r[CH];
    PAI_DEV void load(int k) {
        const uint32_t* p = k < NCH0 ? src0 + k * CH : src1 + (k - NCH0) * CH;
        if constexpr (CH == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
        } else if constexpr (CH == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(p);
            r[0] = v.x; r[1] = v.y;
        } else {
            r[0] = *p;
        }
    }
    PAI_DEV void store(int k) {
        uint32_t* q = k < NCH0 ? dst0 + (k * CH) * stride : dst1 + ((k - NCH0) * CH) * stride;
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i * stride] = r[i];
    }
    PAI_DEV void step(int blk) {
        if (!on) return;
        if (blk >= 1 && blk <= NCH0 + NCH1) store(blk - 1);
        if (blk < NCH0 + NCH1) load(blk);
    }
    // after the product's last block: whatever is still in flight (products with fewer blocks than chunks + 1)
    template <int NB>
    PAI_DEV void drain_from() {
#pragma unroll 1
        for (int k = NB; k <= NCH0 + NCH1; ++k) step(k);
    }
};

// r = a * b * R^-1 mod M (lazy: < 2M when a,b < 2M).  b limbs are read from b_ptr[i * bstride]
// (LDS, [limb][element]); nm = this lane's slice of the modulus.
template <int NLL, int U, int T, class NM, class PF = NoStream>
PAI_DEV void mont_mul(uint32_t (&r)[NLL], const uint32_t (&a)[NLL], const uint32_t* b_ptr, int bstride,
                      const NM& nm, uint32_t n0inv, PF* pf = nullptr) {
    static_assert(NLL % U == 0, "NLL must be a multiple of the row-block size U");
    static_assert(NORM_ROWS % U == 0, "NORM_ROWS must be a multiple of U");
    using RW = Rows<NLL, U, T>;
    uint64_t acc[RW::NW];
    RW::zero(acc);
    constexpr int NB = RW::NL / U;
    constexpr int NORM_BLOCKS = NORM_ROWS / U;
    int since = 0;
#pragma unroll 1
    for (int blk = 0; blk < NB; ++blk) {
        if constexpr (!std::is_same<PF, NoStream>::value) pf->step(blk);
        uint32_t bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = b_ptr[(blk * U + u) * bstride];
        uint32_t low[U];
        RW::template block<true, true>(acc, a, bv, nm, n0inv, low);
        if (++since == NORM_BLOCKS && blk != NB - 1) { RW::normalize(acc); since = 0; }   // finish() takes lazy columns
    }
    if constexpr (!std::is_same<PF, NoStream>::value) pf->template drain_from<NB>();
    RW::finish(acc, r);
}

// r = a * b * 2^(-29 U nblk) mod M for a "minus-one" context (Rows::block_m1): only the first U * nblk limbs of b are
// read (b < 2^(29 U nblk) = R), a may use the whole geometry; npp = this lane's slice of (M + 1) / 2^(29 U).
// Lazy in, lazy out (< 2M) for R > 16 M (the margin lets a sum of two lazy values in as an operand).
template <int NLL, int U, int T, class NM>
PAI_DEV void mont_mul_m1(uint32_t (&r)[NLL], const uint32_t (&a)[NLL], const uint32_t* b_ptr, int bstride, const NM& npp,
                         int nblk) {
    static_assert(NORM_ROWS % U == 0, "NORM_ROWS must be a multiple of U");
    using RW = Rows<NLL, U, T>;
    uint64_t acc[RW::NW];
    RW::zero(acc);
    constexpr int NORM_BLOCKS = NORM_ROWS / U;          // two products per row and column, as in mont_mul
    int since = 0;
    // (reading the digits of block k + 1 while block k multiplies — which pays in kernels_declat.hpp's split-window blocks —
    // measured NEGATIVE here: the two-block loop body loses the compiler's renaming of the window slide: k_dec_a_rl 3.37 ->
    // 3.63 ms, ct x pt 0.372 -> 0.393 ms, profiles/r05/README.md.  So did the ROW form that serves kernels_declat.hpp's chains —
    // one 64-bit column per limb, one quotient digit per row: with three limbs per lane the block's single mask -> broadcast ->
    // multiply chain per three rows is worth more than the row form's four instructions less: small-batch encrypt 0.160 -> 0.204 ms,
    // k_dec_a_rl 3.34 -> 4.1 ms)
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
        uint32_t bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = b_ptr[(blk * U + u) * bstride];
        RW::block_m1(acc, a, bv, npp);
        if (++since == NORM_BLOCKS && blk != nblk - 1) { RW::normalize(acc); since = 0; }
    }
    // block_m1 adds q (M + 1) / 2^(29 U) AFTER the slide, so the window's top U columns still hold partial sums when the
    // last block ends (mont_mul's are zero there): they are the next lane's lowest columns — hand them over before finish()
    if constexpr (T > 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t top = acc[NLL + u];
            const uint32_t lo = from_prev<T>((uint32_t)top), hi = from_prev<T>((uint32_t)(top >> 32));
            acc[u] += ((uint64_t)hi << 32) | lo;
        }
    }
    RW::finish(acc, r);
}

// ---- digit pairs with base M on the lane-group engine ---------------------------------------------------------
// An element x of Z/M^2 is the pair (a, b), a + b M == x R (mod M^2) with R = 2^(29 NL) (mont_padic.hpp has the
// algebra): (a, b) (x) (c, d) = (w, v),  w = (a c + m M) / R,  v = (a d + b c - m + R M + m' M) / R — both
// reductions modulo M (NL limbs) instead of M^2 (2 NL limbs): 5 NL^2 limb products per multiplication instead of the
// 8 NL^2 of a Montgomery product modulo M^2.  Both halves run FUSED, row by row: row i of the first half yields the
// quotient digit m_i, which is exactly what column i of the second half still needs, so it enters the second window
// (group lane 0, as 2^29 - 1 - m_i; the + 1 and the (M - 1) R that complete R M - m enter at column 0 and at the top
// of the group's last lane) and is never stored.
// (a, b) in registers (lane slices), (c, d) as LDS rows c_ptr / d_ptr [limb * stride], M - 1 as LDS limbs (uniform).
// Inputs lazy (< 2M + eps), outputs lazy; R / M >= 2^20 required.
constexpr int PAIR_NORM_MAX = 18;              // three 2^58 products per row and column: 18 rows (54 x 2^58 + the normalised rest < 2^29 + 2^35) stay below 2^64
constexpr int PAIR_NORM_MAX2 = 24;
constexpr int PAIR_FULL = 0, PAIR_SQR = 1, PAIR_C0 = 2;       // forms of pair_mul             // two products per row and column (squarings, g-factored operands): 24 rows

// One row block of the fused product rule on a window that is addressed THROUGH an offset: column k of the window is
// register (O + k) mod NW.  With O a compile-time constant the slide of the window after a block is a renaming, not NLL
// 64-bit moves per window (round 5: 33-72 of the ~840 instructions of an 18 x 8 block were those moves).
template <int NLL, int U, int T, int O, int FORM, class NM>
PAI_DEV void pair_block(uint64_t (&acc1)[NLL + U], uint64_t (&acc2)[NLL + U], const uint32_t (&a)[NLL], const uint32_t (&b)[NLL],
                        const uint32_t (&cv)[U], const uint32_t (&dv)[U], const NM& nm, uint32_t n0inv, bool lane0) {
    constexpr bool sqr = FORM == PAIR_SQR, c0 = FORM == PAIR_C0;
    constexpr int NW = NLL + U;
#define PAIR_C(k) ((O + (k)) % NW)
    uint32_t low1[U], low2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc1[PAIR_C(j + u)] += (uint64_t)a[j] * cv[u];
        const uint32_t q1 = bcast0<T>(((uint32_t)acc1[PAIR_C(u)] * n0inv) & RMASK);
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc1[PAIR_C(j + u)] += (uint64_t)nm.limb(j) * q1;
        acc1[PAIR_C(u + 1)] += acc1[PAIR_C(u)] >> RB;
        low1[u] = (uint32_t)acc1[PAIR_C(u)] & RMASK;
        acc2[PAIR_C(u)] += lane0 ? (uint64_t)(RMASK - q1) : 0ull;
        if constexpr (!c0) {
#pragma unroll
            for (int j = 0; j < NLL; ++j) acc2[PAIR_C(j + u)] += (uint64_t)a[j] * dv[u];
        }
        if constexpr (!sqr) {
#pragma unroll
            for (int j = 0; j < NLL; ++j) acc2[PAIR_C(j + u)] += (uint64_t)b[j] * cv[u];
        }
        const uint32_t q2 = bcast0<T>(((uint32_t)acc2[PAIR_C(u)] * n0inv) & RMASK);
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc2[PAIR_C(j + u)] += (uint64_t)nm.limb(j) * q2;
        acc2[PAIR_C(u + 1)] += acc2[PAIR_C(u)] >> RB;
        low2[u] = (uint32_t)acc2[PAIR_C(u)] & RMASK;
    }
    // the retired limbs (zero in the group's lane 0) go to the previous lane's top columns; the retired columns become the
    // window's new top
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if constexpr (T > 1) {
            acc1[PAIR_C(NLL + u)] += (uint64_t)from_next_z<T>(low1[u]);
            acc2[PAIR_C(NLL + u)] += (uint64_t)from_next_z<T>(low2[u]);
        }
        acc1[PAIR_C(u)] = 0;
        acc2[PAIR_C(u)] = 0;
    }
#undef PAIR_C
}
// carry-save normalisation of a window at offset O (Rows::normalize in window order)
template <int NLL, int U, int O>
PAI_DEV void pair_normalize(uint64_t (&acc)[NLL + U]) {
    constexpr int NW = NLL + U;
#pragma unroll
    for (int k = NW - 1; k >= 1; --k) {
        const uint64_t keep = (k == NW - 1) ? acc[(O + k) % NW] : (acc[(O + k) % NW] & RMASK);
        acc[(O + k) % NW] = keep + (acc[(O + k - 1) % NW] >> RB);
    }
    acc[O % NW] &= RMASK;
}
// PER = NLL / U + 1 consecutive blocks bring the offset back to 0: one period, fully unrolled
template <int NLL, int U, int T, int B, int FORM, class NM, class PF>
PAI_DEV void pair_period(uint64_t (&acc1)[NLL + U], uint64_t (&acc2)[NLL + U], const uint32_t (&a)[NLL], const uint32_t (&b)[NLL],
                         const uint32_t* c_ptr, const uint32_t* d_ptr, int stride, int blk0, const NM& nm, uint32_t n0inv,
                         bool lane0, PF* pf) {
    constexpr int PER = NLL / U + 1;
    constexpr bool sqr = FORM == PAIR_SQR, c0 = FORM == PAIR_C0;
    if constexpr (B < PER) {
        if constexpr (!std::is_same<PF, NoStream>::value) pf->step(blk0 + B);
        uint32_t cv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cv[u] = c_ptr[((blk0 + B) * U + u) * stride];
            dv[u] = c0 ? 0u : d_ptr[((blk0 + B) * U + u) * stride];
            dv[u] = sqr ? dv[u] << 1 : dv[u];
        }
        pair_block<NLL, U, T, (B * U) % (NLL + U), FORM>(acc1, acc2, a, b, cv, dv, nm, n0inv, lane0);
        // (keeps the scheduler from hoisting the next blocks' operand reads over this block: at 256 registers they spill)
        __builtin_amdgcn_sched_barrier(0);
        // three products per row and column in the second window of a full product: it cannot wait for the period's end
        if constexpr ((B + 1) * U <= PAIR_NORM_MAX && (B + 2) * U > PAIR_NORM_MAX && B + 1 < PER) {
            if constexpr (FORM == PAIR_FULL) pair_normalize<NLL, U, ((B + 1) * U) % (NLL + U)>(acc2);
        }
        pair_period<NLL, U, T, B + 1, FORM>(acc1, acc2, a, b, c_ptr, d_ptr, stride, blk0, nm, n0inv, lane0, pf);
    }
}

// FORM (compile time: every variant is straight-line code, no wave-uniform branches inside the rows):
//   PAIR_FULL  the product rule as above, 5 NL^2;
//   PAIR_SQR   the rows at c_ptr / d_ptr are (a, b) themselves — a d + b c is then 2 a d, one multiply-accumulate per limb
//              pair less: 4 NL^2;
//   PAIR_C0    the right operand has no second digit (g-factored table entries, kernels_pair.hpp): a d drops out, 4 NL^2 as
//              well, and d_ptr is not read.
// Round 5: the window is renamed instead of moved where the geometry allows it (pair_block), the (M - 1) R term joins once
// at the end (it adds M - 1 to the second result digit) instead of one LDS read, select and 64-bit add per row, the
// cross-lane moves need no selects (from_next_z, bcast0), and the two-product forms normalise every 24 rows.
// ROT = false keeps the compact sliding form (the one-off products around a kernel's hot loop: a renamed period is PER
// copies of the block).
template <int NLL, int U, int T, int FORM = PAIR_FULL, bool ROT = true, class NM, class PF = NoStream>
PAI_DEV void pair_mul(uint32_t (&a)[NLL], uint32_t (&b)[NLL], const uint32_t* c_ptr, const uint32_t* d_ptr, int stride,
                      const NM& nm, uint32_t n0inv, PF* pf = nullptr) {
    static_assert(NLL % U == 0 && U <= PAIR_NORM_MAX, "row-block size");
    constexpr bool sqr = FORM == PAIR_SQR, c0 = FORM == PAIR_C0;
    using RW = Rows<NLL, U, T>;
    constexpr int NW = RW::NW;
    uint64_t acc1[NW], acc2[NW];
    RW::zero(acc1);
    RW::zero(acc2);
    const bool lane0 = (group_lane<T>() == 0);
    if (lane0) acc2[0] = 1;
    constexpr int NB = RW::NL / U;
    constexpr int PER = NLL / U + 1;
    if constexpr (ROT && NB % PER == 0 && PER * U <= PAIR_NORM_MAX2) {
#pragma unroll 1
        for (int blk = 0; blk < NB; blk += PER) {
            pair_period<NLL, U, T, 0, FORM>(acc1, acc2, a, b, c_ptr, d_ptr, stride, blk, nm, n0inv, lane0, pf);
            if (blk + PER < NB) { pair_normalize<NLL, U, 0>(acc1); pair_normalize<NLL, U, 0>(acc2); }
        }
    } else {
        constexpr int norm_blocks = ((sqr || c0) ? PAIR_NORM_MAX2 : PAIR_NORM_MAX) / U;
        int since = 0;
#pragma unroll 1
        for (int blk = 0; blk < NB; ++blk) {
            if constexpr (!std::is_same<PF, NoStream>::value) pf->step(blk);
            uint32_t cv[U], dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cv[u] = c_ptr[(blk * U + u) * stride];
                dv[u] = c0 ? 0u : d_ptr[(blk * U + u) * stride];
                dv[u] = sqr ? dv[u] << 1 : dv[u];
            }
            pair_block<NLL, U, T, 0, FORM>(acc1, acc2, a, b, cv, dv, nm, n0inv, lane0);
            // slide: the block left its retired (zeroed) columns at the bottom
#pragma unroll
            for (int j = 0; j < NLL; ++j) { acc1[j] = acc1[j + U]; acc2[j] = acc2[j + U]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc1[NLL + u] = 0; acc2[NLL + u] = 0; }
            if (++since == norm_blocks && blk != NB - 1) { RW::normalize(acc1); RW::normalize(acc2); since = 0; }
        }
    }
    if constexpr (!std::is_same<PF, NoStream>::value) pf->template drain_from<NB>();
    // + (M - 1) R, i.e. M - 1 on the second result digit (lazy columns: finish() carries)
#pragma unroll
    for (int j = 0; j < NLL; ++j) acc2[j] += (uint64_t)nm.limb(j);
    if (lane0) acc2[0] -= 1;
    RW::finish(acc1, a);
    RW::finish(acc2, b);
}

// Plain product with an additive constant: a*b + init, where `init` (this lane's slice, < 2^(29*NL))
// seeds the accumulator.  The low NL limbs are written by the group's lane 0 to lo_ptr[i*lstride]
// (LDS), the high NL limbs are returned distributed over the group in `hi`.
template <int NLL, int U, int T>
PAI_DEV void mul_plain(uint32_t (&hi)[NLL], const uint32_t (&init)[NLL], const uint32_t (&a)[NLL],
                       const uint32_t* b_ptr, int bstride, uint32_t* lo_ptr, int lstride) {
    using RW = Rows<NLL, U, T>;
    uint64_t acc[RW::NW];
#pragma unroll
    for (int j = 0; j < NLL; ++j) acc[j] = init[j];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[NLL + u] = 0;
    constexpr int NB = RW::NL / U;
    constexpr int NORM_BLOCKS = NORM_ROWS / U;
    int since = 0;
    const bool lane0 = (group_lane<T>() == 0);
    NmRegs<1> none{};
#pragma unroll 1
    for (int blk = 0; blk < NB; ++blk) {
        uint32_t bv[U], low[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bv[u] = b_ptr[(blk * U + u) * bstride];
        RW::template block<true, false>(acc, a, bv, none, 0u, low);
        if (lane0) {
#pragma unroll
            for (int u = 0; u < U; ++u) lo_ptr[(blk * U + u) * lstride] = low[u];
        }
        if (++since == NORM_BLOCKS && blk != NB - 1) { RW::normalize(acc); since = 0; }   // finish() takes lazy columns
    }
    RW::finish(acc, hi);
}

// Montgomery reduction of a 2*NL-limb value t (< M*R): r = t * R^-1 mod M (lazy, < 2M).
// lo = this lane's slice of the low NL limbs; the high limbs are read from hi_ptr[i*hstride]
// (LDS [limb][element], limb index relative to NL).
template <int NLL, int U, int T, class NM>
PAI_DEV void mont_redc(uint32_t (&r)[NLL], const uint32_t (&lo)[NLL], const uint32_t* hi_ptr, int hstride,
                       const NM& nm, uint32_t n0inv) {
    using RW = Rows<NLL, U, T>;
    uint64_t acc[RW::NW];
#pragma unroll
    for (int j = 0; j < NLL; ++j) acc[j] = lo[j];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[NLL + u] = 0;
    constexpr int NB = RW::NL / U;
    constexpr int NORM_BLOCKS = NORM_ROWS / U;
    int since = 0;
    const bool top = (group_lane<T>() == T - 1);
    uint32_t bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) bv[u] = 0;
#pragma unroll 1
    for (int blk = 0; blk < NB; ++blk) {
        // the window top of the group's last lane is where the high half enters
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t h = hi_ptr[(blk * U + u) * hstride];
            acc[NLL + u] = top ? (uint64_t)h : 0ull;
        }
        uint32_t low[U];
        RW::template block<false, true>(acc, lo, bv, nm, n0inv, low);
        if (++since == NORM_BLOCKS && blk != NB - 1) { RW::normalize(acc); since = 0; }   // finish() takes lazy columns
    }
    RW::finish(acc, r);
}

// canonicalise x (< 2M, 29-bit limbs, distributed over the group) into [0, M)
template <int NLL, int T, class NM>
PAI_DEV void cond_sub(uint32_t (&x)[NLL], const NM& nm) {
    uint32_t d[NLL];
    int32_t borrow = 0;        // 0 or -1, assuming no borrow-in
#pragma unroll
    for (int j = 0; j < NLL; ++j) {
        int32_t t = (int32_t)x[j] - (int32_t)nm.limb(j) + borrow;   // 29-bit limbs: fits in int32
        d[j] = (uint32_t)t & RMASK;
        borrow = t >> RB;
    }
    if constexpr (T == 64) {
        // one integer per wavefront: borrow look-ahead over the lanes by one 64-bit addition — lane l GENERATES a borrow
        // (bit l of g) or PROPAGATES one it is handed (all its limbs zero: bit l of p); with x = g | p, y = g the carry
        // into bit l of x + y is the borrow into lane l, the carry out the borrow out of the integer
        const uint64_t g = __ballot(borrow != 0);
        bool allz = borrow == 0;
#pragma unroll
        for (int j = 0; j < NLL; ++j) allz = allz && d[j] == 0;
        const uint64_t pm = __ballot(allz);
        const uint64_t xs = g | pm, sum = xs + g;
        const uint64_t cin = sum ^ pm;
        int32_t bo = (int32_t)((cin >> (threadIdx.x & 63)) & 1u) ? -1 : 0;
#pragma unroll
        for (int j = 0; j < NLL; ++j) {
            int32_t t = (int32_t)d[j] + bo;
            d[j] = (uint32_t)t & RMASK;
            bo = t >> RB;
        }
        borrow = (sum < xs) ? -1 : 0;
    } else if constexpr (T > 1) {
        // Every lane hands its borrow to the next one AT ONCE and the receivers re-propagate in one pass.  A received borrow
        // changes a lane's own borrow-out only if all its limbs are zero (then the NEW borrow goes round again: practically never),
        // so the loop runs one pass where the lane-by-lane ripple of rounds 1-5 ran T - 1 (31 on the 32-lane latency geometries).
        int32_t pend = borrow;                               // the borrow this lane has not handed on yet (0 / -1)
        while (true) {
            const int32_t bin = (int32_t)from_prev<T>((uint32_t)pend);      // the group's first lane receives 0
            if (!__any(bin != 0)) break;
            int32_t bo = bin;
#pragma unroll
            for (int j = 0; j < NLL; ++j) {
                int32_t t = (int32_t)d[j] + bo;
                d[j] = (uint32_t)t & RMASK;
                bo = t >> RB;
            }
            pend = borrow == 0 ? bo : 0;                     // a lane that had borrowed out already hands nothing new
            borrow |= bo;
        }
        // the decision is the borrow out of the group's last lane
        int32_t last = (int32_t)__shfl((int)borrow, (int)(((threadIdx.x & 63) & ~(T - 1)) + T - 1), 64);
        borrow = last;
    }
    const bool ge = (borrow == 0);
#pragma unroll
    for (int j = 0; j < NLL; ++j) x[j] = ge ? d[j] : x[j];
}

}  // namespace pai
