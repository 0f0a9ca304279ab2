// Decrypt stage A for the SMALLEST batches on digit pairs, pipelined over the four waves of a workgroup (round 5).
//
// The reference's own benchmark decrypts 16 and 64 ciphertexts (bench/bench_ipcl_python.py:33-42): there the cost of
// ipclPrivateKey.decrypt (classes.cpp:127-133 -> ipcl::PrivateKey::decrypt) is the LATENCY of one chain of ~1023 dependent
// squarings modulo s^2 per (ciphertext, prime) — k_dec_a_rl runs it at ~3.3 us per squaring with the integer spread over a
// wavefront (71 rows of a minus-one Montgomery product).  On digit pairs (mont_padic.hpp: x in Z/s^2 kept as (a, b),
// a + b s == x R mod s^2) a squaring is
//     w = (a^2 + m s) / R                      (first digit: a Montgomery squaring modulo s — HALF the rows),
//     v = (2 a b - m + R s + m' s) / R         (second digit: one product and one reduction modulo s, plus the quotient m)
// and the first-digit chain never needs a second digit.  So the chain runs on TWO waves: W1 squares first digits back to
// back and publishes every a_i together with its quotient digits m_i in a ring of LDS buffers; W2 follows one step behind
// with b_{i+1} = (2 a_i b_i - m_i + ...) / R.  Right to left (as k_dec_a_rl): two more waves, B1 / B2, multiply the pairs
// (a_i, b_i) at the set bits of s - 1 into the accumulator pair the same way (first digit ahead, second digit behind).
// The critical path is ~1023 half-size steps instead of 1023 full-size ones.
//
// The base is s' = s k with k = -s^-1 mod 2^29 (a minus-one context: the quotient digit of a row IS the limb it retires);
// everything is arithmetic modulo s'^2, of which s^2 is a divisor, so the power modulo s^2 — what stage B expects — comes out
// of one plain product a + b s', one reduction (on the 3 x 64 geometry s^2 needs) and a short-quotient Barrett step.
// One (ciphertext, prime) per workgroup; in the chain an integer is ONE limb per lane (s' of at most 64 limbs: 2048- and
// 3072-bit keys) or two (4096-bit keys; ct * pt with base n k at 2048-bit keys).
#pragma once
#include "kernels_paillier.hpp"

namespace pai {

struct DecPPParams {
    const MontCtx* pp[2];        // minus-one contexts of s' = s k (rows r, R = 2^(29 r))
    const uint32_t* kdig[2];     // [nd][2][r]: base-s' digits (Ka, Kb) of R^(i+2) mod s'^2 (the ciphertext's way into digit form)
    const uint32_t* kx[2];       // [nch][NL]: R^-1 R_sq^(j+2) mod (s^2 k2): takes chunk j (base R_sq) of a + b s' into the Montgomery form of sq[]
    const MontCtx* sq[2];        // minus-one contexts of s^2 (exit)
    const MontCtx* fin[2];       // conventional contexts of s^2 (exit: the canonical residue)
    const uint32_t* expo[2];     // s - 1, packed words (VAR: [0] = the exponents, e_words each, one row when e_bcast)
    int ebits[2];                // (VAR: [0] = ebits_max)
    int nd, nch, ct_words, u_words;
    int e_words = 0, e_bcast = 0;
};

constexpr int PP_RMAX = 80;      // limbs of s' (4096-bit keys: 75 rows)
constexpr int PP_RING = 16;      // (a_i, m_i, b_i) slots between the squaring waves and the product waves
constexpr int PP_PRING = 8;      // (A, m) slots between the product waves
constexpr int PP_MAXND = 6;
constexpr int PP_MAXCH = 3;      // base-R_sq chunks of a + b s' at the exit
constexpr int PP_YBUF = 2 * PP_RMAX + 16;
constexpr int PP_EWORDS = 128;   // exponent words (4096 bits)


// G: the geometry of the exit (s^2 / n^2 on 3 x 64); GC: the geometry of the chain — ONE limb per lane where s' fits 64 limbs
// (decryption with keys up to ~3600 bits), else two.
template <class G, class GC>
struct PPLds {
    static constexpr int ROW = GC::NL;                                // words per ring slot: one lane-sliced integer of the chain geometry
    static constexpr int STAGE = 0;                                   // G::LDS_WORDS: operand staging of the exit products
    static constexpr int YBUF = STAGE + G::LDS_WORDS;                 // a + b s' assembled (2 r + 2 limbs)
    static constexpr int RING_A = YBUF + PP_YBUF;
    static constexpr int RING_M = RING_A + PP_RING * ROW;
    static constexpr int RING_B = RING_M + PP_RING * PP_RMAX;
    static constexpr int PRING_A = RING_B + PP_RING * ROW;
    static constexpr int PRING_M = PRING_A + PP_PRING * ROW;
    static constexpr int KD = PRING_M + PP_PRING * PP_RMAX;           // [nd][2][RMAX]
    static constexpr int MLIM = KD + PP_MAXND * 2 * PP_RMAX;          // limbs of s'
    static constexpr int KX = MLIM + PP_RMAX;                         // exit constants, [nch][NL]
    static constexpr int ZERO = KX + PP_MAXCH * G::NL;                // RMAX zero words (the feed of lanes != 0)
    static constexpr int TMPM = ZERO + PP_RMAX;                       // quotient digits of the entry products
    static constexpr int EXPO = TMPM + PP_RMAX;                       // the exponent's words (the product waves scan its bits)
    static constexpr int FLAGS = EXPO + PP_EWORDS;
    static constexpr int WORDS = FLAGS + 16;
    static constexpr int BYTES = WORDS * 4;
    static_assert(YBUF % 4 == 0 && RING_A % 4 == 0 && RING_M % 4 == 0 && RING_B % 4 == 0 && PRING_A % 4 == 0 && PRING_M % 4 == 0 &&
                  KD % 4 == 0 && ZERO % 4 == 0 && TMPM % 4 == 0 && PP_RMAX % 4 == 0 && ROW % 4 == 0, "digit rows are read 16 bytes at a time");
};

// Lazy 64-bit columns (< 2^61 each) of an integer spread over the wave, NLL limbs per lane -> canonical 29-bit limbs, without a
// data-dependent loop: the carry out of a lane is < 2^32 after its own pass, < 8 after the neighbour's first visit, 0 or 1
// after the second, and the 0 / 1 ripple is ONE 64-bit addition of the lanes' generate / propagate ballots (cond_sub's
// look-ahead).  Rows::finish loops "while any lane still has a carry": three or four rounds of compare, branch and a 64-bit
// add per call — a tenth of a chain step once a row costs 12 instructions.
template <int NLL>
PAI_DEV void pp_finish(const uint64_t (&col)[NLL], uint32_t (&r)[NLL]) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < NLL; ++j) {
        const uint64_t t = col[j] + c;
        r[j] = (uint32_t)t & RMASK;
        c = t >> RB;
    }
    uint32_t cout = (uint32_t)c;                                      // < 2^32
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t cin = from_prev<64>(cout);
#pragma unroll
        for (int j = 0; j < NLL; ++j) {
            const uint32_t t = r[j] + cin;                            // < 2^29 + 2^32 - 2^29 in the first pass (r[0] only), < 2^30 after
            r[j] = t & RMASK;
            cin = t >> RB;
        }
        cout = cin;
    }
    bool allones = true;
#pragma unroll
    for (int j = 0; j < NLL; ++j) allones = allones && r[j] == RMASK;
    const uint64_t g = __ballot(cout != 0), pm = __ballot(allones);
    const uint64_t xs = g | pm, sum = xs + g;
    uint32_t cin = (uint32_t)(((sum ^ pm) >> (threadIdx.x & 63)) & 1u);
#pragma unroll
    for (int j = 0; j < NLL; ++j) {
        const uint32_t t = r[j] + cin;
        r[j] = t & RMASK;
        cin = t >> RB;
    }
}
// rl_publish without the exec mask around lane 0's store: every lane stores the same word
PAI_DEV void pp_publish(uint32_t* flag, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One half of the product rule on a minus-one context with k of ONE limb (s' == -1 mod 2^29), the integer spread over the wave
// with NLL limbs per lane (1 where s' fits 64 limbs, else 2):
//   r = (x1 * dig1 [+ x2 * dig2] [+ R - m] + q s') / R  [+ s' - 1]
// x1, x2: this lane's limb slices; dig1, dig2: LDS limbs (stride 1, nrows of them); FEED: m arrives as complemented digits
// (2^29 - 1 - m_i) at lds[fd_off ..] for lane 0 and as zeros for the other lanes (fd_off is per lane); EXPORT: this half's own
// quotient digits leave, complemented, to lds[mq_off ..].
// The window is ONE 64-bit column per limb: with col = lo + 2^29 hi, dividing by 2^29 after the quotient digit q = lo of the
// wave's first column has zeroed that column is   col_k <- lo_(k+1) + hi_k + q npp_k   (lo of the NEXT lane's first column for a
// lane's last one) — per row and lane NLL x (multiply, mask, shift, multiply, add) plus one broadcast and one cross-lane move,
// nothing to slide, zero or normalise (a column lives one row: < 3 * 2^58 + 2^36).  Rounds 5's first version ran blocks of
// three rows on three limbs per lane (64 instructions for 18 multiplies per block, 13 of 64 lanes holding limbs at 2048-bit
// keys): 21 instructions per row against 12 here.  Four rows per loop iteration, their digits read 16 bytes at a time one
// iteration ahead (a lone wave has nobody to hide an LDS round trip behind); the quotient digits leave as the UNIFORM value
// every lane holds after the broadcast — all lanes store the same words: no dump rows, no exec masking.
// nrows is a multiple of 4 (the host rounds R up: a tail of single rows measured slower than the rows it saves) and smaller
// than the limbs of the geometry.
template <class GC, bool TWO, bool FEED, bool EXPORT>
PAI_DEV void pp_half(uint32_t (&r)[GC::NLL], const uint32_t (&x1)[GC::NLL], int dig1, const uint32_t (&x2)[GC::NLL], int dig2,
                     uint32_t* lds, int fd_off, int mq_off, const NmRegs<GC::NLL>& npp, const uint32_t (&mtrue)[GC::NLL], int nrows) {
    constexpr int NLL = GC::NLL;
    static_assert(GC::U == 1 && GC::T == 64, "one limb retired per row, one integer per wavefront");
    const bool lane0 = (threadIdx.x & 63) == 0;
    auto ld4 = [&](int off) -> uint4 { return *reinterpret_cast<const uint4*>(lds + off); };
    uint4 b1 = ld4(dig1), b2 = TWO ? ld4(dig2) : make_uint4(0, 0, 0, 0), f = FEED ? ld4(fd_off) : make_uint4(0, 0, 0, 0);
    // col[] always holds the columns WITH the current row's products: the next row's products (and feed digit) do not depend
    // on this row's quotient digit, so they join the sum the quotient product is added to — the chain from row to row is
    // mask -> broadcast -> one multiply-add.  (The row after the last one reads a zero digit: every digit row is followed by
    // zero words, the limbs of the operand beyond nrows.)
    uint64_t col[NLL];
#pragma unroll
    for (int j = 0; j < NLL; ++j) {
        col[j] = (uint64_t)x1[j] * b1.x;
        if constexpr (TWO) col[j] += (uint64_t)x2[j] * b2.x;
    }
    if constexpr (FEED) col[0] += (uint64_t)f.x + (lane0 ? 1u : 0u);  // R - m = sum (2^29 - 1 - m_i) 2^(29 i) + 1
    // one row: the quotient digit of the current columns, then the columns of the next row (its digits nb1, nb2, nf)
    auto row = [&](uint32_t nb1, uint32_t nb2, uint32_t nfd) -> uint32_t {
        uint32_t lo[NLL];
#pragma unroll
        for (int j = 0; j < NLL; ++j) lo[j] = (uint32_t)col[j] & RMASK;
        const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo[0]);      // (masking AFTER the broadcast, on the scalar side: 1.81 against 1.55 ms)
        const uint32_t t = from_next<64>(lo[0]);
        uint64_t sum[NLL];
#pragma unroll
        for (int j = 0; j < NLL; ++j) {
            sum[j] = (col[j] >> RB) + (uint64_t)x1[j] * nb1;
            if constexpr (TWO) sum[j] += (uint64_t)x2[j] * nb2;
            sum[j] += (uint64_t)(j + 1 < NLL ? lo[j + 1] : t);
        }
        if constexpr (FEED) sum[0] += (uint64_t)nfd;
#pragma unroll
        for (int j = 0; j < NLL; ++j) col[j] = (uint64_t)npp.v[j] * q + sum[j];
        return q;
    };
#pragma unroll 1
    for (int g = 0; g < nrows; g += 4) {
        const uint4 n1 = ld4(dig1 + g + 4);                           // (the group beyond the last: zeros, see above)
        const uint4 n2 = TWO ? ld4(dig2 + g + 4) : make_uint4(0, 0, 0, 0);
        const uint4 nf = FEED ? ld4(fd_off + g + 4) : make_uint4(0, 0, 0, 0);
        const uint32_t c1[5] = {b1.x, b1.y, b1.z, b1.w, n1.x}, c2[5] = {b2.x, b2.y, b2.z, b2.w, n2.x}, cf[5] = {f.x, f.y, f.z, f.w, nf.x};
        uint32_t q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = row(c1[u + 1], c2[u + 1], cf[u + 1]);
        if constexpr (EXPORT) *reinterpret_cast<uint4*>(lds + mq_off + g) = make_uint4(RMASK - q[0], RMASK - q[1], RMASK - q[2], RMASK - q[3]);
        b1 = n1; b2 = n2; f = nf;
    }
    if constexpr (FEED) {                                             // + (s' - 1) R on the numerator
#pragma unroll
        for (int j = 0; j < NLL; ++j) col[j] += (uint64_t)mtrue[j];
        if (lane0) col[0] -= 1;
    }
    pp_finish<NLL>(col, r);
}

// this lane's slices of a ring slot (PP_ROW words: the whole lane-sliced integer, limbs beyond the value are zero)
template <class G>
PAI_DEV void pp_load(uint32_t (&x)[G::NLL], const uint32_t* lds, int off) {
    const int l = (int)threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) x[j] = lds[off + G::NLL * l + j];
}
template <class G>
PAI_DEV void pp_store(uint32_t* lds, int off, const uint32_t (&x)[G::NLL]) {
    const int l = (int)threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < G::NLL; ++j) lds[off + G::NLL * l + j] = x[j];
}

// Dev builds only (-DPP_PROFILE): every wave of workgroup (0, 0) reports its cycles in total and inside rl_wait
#ifdef PP_PROFILE
#define PP_T0() const unsigned long long pp_t0 = __builtin_readcyclecounter(); unsigned long long pp_wait = 0
#define PP_WAIT(expr) do { const unsigned long long w0_ = __builtin_readcyclecounter(); expr; pp_wait += __builtin_readcyclecounter() - w0_; } while (0)
#define PP_DECL(a, b) unsigned long long a = 0, b = 0
#define PP_STAMP(var) var = __builtin_readcyclecounter() - pp_t0
#define PP_REPORT2(role, a, b) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) printf("PP %s: cycles %llu, waiting %llu; stamps %llu %llu\n", role, __builtin_readcyclecounter() - pp_t0, pp_wait, a, b); } while (0)
#define PP_REPORT(role) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) printf("PP %s: cycles %llu, waiting %llu\n", role, __builtin_readcyclecounter() - pp_t0, pp_wait); } while (0)
#else
#define PP_T0() do { } while (0)
#define PP_WAIT(expr) expr
#define PP_REPORT(role) do { } while (0)
#define PP_STAMP(var) do { } while (0)
#define PP_DECL(a, b) do { } while (0)
#define PP_REPORT2(role, a, b) do { } while (0)
#endif

// VAR = false: decrypt stage A (workgroup (i, w): ciphertext i to the power s_w - 1 modulo s_w^2, s_0 = p, s_1 = q).
// VAR = true: ct * pt for the smallest batches (ipclCipherText.__mul__ -> CipherText::operator*(PlainText), classes.cpp /
// the reference's BM_Mul_CTPT): ONE modulus (s = n, the power modulo n^2) and the exponent of each element its own; exponent 0
// gives 1.
template <class G, class GC, bool VAR>
PAI_DEV void pp_chain(const DecPPParams& P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out, int n, uint32_t* lds) {
    static_assert(G::M1 && G::T == 64 && GC::T == 64 && GC::U == 1 && BLOCK_THREADS == 256, "one integer per wavefront, four waves per chain");
    constexpr int NLL = GC::NLL, U = GC::U;          // the chain's geometry; the exit (B2's tail) runs on G
    constexpr int GN = G::NLL, GU = G::U;
    using L = PPLds<G, GC>;
    const int which = VAR ? 0 : (int)blockIdx.y;
    const MontCtx* ctx = P.pp[which];
    const uint32_t* expo = P.expo[which];
    const int ebits = P.ebits[which];
    // (rotating the roles over the waves between the workgroups that share a CU measured slower — 2.08 against 1.97 ms at two
    // workgroups per CU, 3.3 against 2.65 at four: the dispatcher already spreads the chain waves)
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const bool lane0 = lane == 0;
    const int r = (int)ctx->rows, nblk = r / U;
    NmRegs<NLL> npp;
    uint32_t mtrue[NLL];
#pragma unroll
    for (int j = 0; j < NLL; ++j) { npp.v[j] = ctx->npp[NLL * lane + j]; mtrue[j] = ctx->n[NLL * lane + j]; }
    // constants to LDS: digit pairs of R^(i+2), the limbs of s', the exit constant, zeros
    for (int i = threadIdx.x; i < P.nd * 2 * PP_RMAX; i += BLOCK_THREADS) {
        const int e = i / PP_RMAX, k = i - e * PP_RMAX;
        lds[L::KD + i] = k < r ? P.kdig[which][(size_t)e * r + k] : 0u;
    }
    for (int i = threadIdx.x; i < PP_RMAX; i += BLOCK_THREADS) { lds[L::MLIM + i] = ctx->n[i]; lds[L::ZERO + i] = 0u; }
    for (int i = threadIdx.x; i < P.nch * G::NL; i += BLOCK_THREADS) lds[L::KX + i] = P.kx[which][i];
    for (int i = threadIdx.x; i < PP_YBUF; i += BLOCK_THREADS) lds[L::YBUF + i] = 0u;
    // (the quotient-digit rows are read one group beyond their nrows words: zeros)
    for (int i = threadIdx.x; i < PP_RING * PP_RMAX; i += BLOCK_THREADS) lds[L::RING_M + i] = 0u;
    for (int i = threadIdx.x; i < PP_PRING * PP_RMAX; i += BLOCK_THREADS) lds[L::PRING_M + i] = 0u;
    for (int i = threadIdx.x; i < PP_RMAX; i += BLOCK_THREADS) lds[L::TMPM + i] = 0u;
    uint32_t* flags = lds + L::FLAGS;
    uint32_t* headA = flags, *headB = flags + 1, *headP = flags + 2, *tailB1 = flags + 4, *tailB2 = flags + 5, *tailP = flags + 6;
    const int ewords = VAR ? P.e_words : (ebits + 31) / 32;
    // lowest set bit of the exponent at or above i (ebits: none): one LDS word per call, mostly
    auto next_set = [&](int i) -> int {
        while (i < ebits) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[L::EXPO + (i >> 5)]) >> (i & 31);
            if (w) {
                i += __builtin_ctz(w);
                return i < ebits ? i : ebits;
            }
            i = (i | 31) + 1;
        }
        return ebits;
    };
    // lane 0 feeds from the quotient-digit buffer, the other lanes read zeros
    auto feed_off = [&](int m_off) -> int { return lane0 ? m_off : (int)L::ZERO; };
    auto mq_off = [](int m_off) -> int { return m_off; };                    // (the export is a uniform store)
    auto slotA = [](int i) -> int { return L::RING_A + (int)((unsigned)i % PP_RING) * L::ROW; };
    auto slotM = [](int i) -> int { return L::RING_M + (int)((unsigned)i % PP_RING) * PP_RMAX; };
    auto slotB = [](int i) -> int { return L::RING_B + (int)((unsigned)i % PP_RING) * L::ROW; };
    auto slotPA = [](int k) -> int { return L::PRING_A + (int)((unsigned)k % PP_PRING) * L::ROW; };
    auto slotPM = [](int k) -> int { return L::PRING_M + (int)((unsigned)k % PP_PRING) * PP_RMAX; };
    uint32_t none[NLL];
#pragma unroll
    for (int j = 0; j < NLL; ++j) none[j] = 0;
    const int tiles = n;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ei = tile;
        if constexpr (VAR) expo = P.expo[0] + (P.e_bcast ? (size_t)0 : (size_t)ei * P.e_words);
        if (threadIdx.x < 16) flags[threadIdx.x] = 0;
        if ((int)threadIdx.x < ewords) lds[L::EXPO + threadIdx.x] = expo[threadIdx.x];
        __syncthreads();
        if (wave == 0) {
            // ---- W1: the ciphertext's digit form, then the chain of first digits --------------------------------------
            const uint32_t* row = ct + (size_t)ei * P.ct_words;
            uint32_t sa[NLL], sb[NLL];
            PP_T0();
#pragma unroll
            for (int j = 0; j < NLL; ++j) { sa[j] = 0; sb[j] = 0; }
#pragma unroll 1
            for (int i = 0; i < P.nd; ++i) {
                uint32_t c[NLL], w[NLL], v[NLL];
                load_elem_off<GC>(c, row, P.ct_words, r * i);
#pragma unroll
                for (int j = 0; j < NLL; ++j) c[j] = (NLL * lane + j < r) ? c[j] : 0u;
                wave_lds_fence();
                pp_half<GC, false, false, true>(w, c, L::KD + (2 * i) * PP_RMAX, none, 0, lds, 0, mq_off(L::TMPM), npp, mtrue, nblk);
                wave_lds_fence();
                pp_half<GC, false, true, false>(v, c, L::KD + (2 * i + 1) * PP_RMAX, none, 0, lds, feed_off(L::TMPM), 0, npp, mtrue,
                                                      nblk);
                add_limbs<GC>(sa, w);
                add_limbs<GC>(sb, v);
            }
            pp_store<GC>(lds, slotA(0), sa);
            pp_store<GC>(lds, slotB(0), sb);
            pp_publish(headB, 1u);
            pp_publish(headA, 1u);
            uint32_t x[NLL];
#pragma unroll
            for (int j = 0; j < NLL; ++j) x[j] = sa[j];
            uint32_t seen2 = 0, seen3 = 0, seen4 = 0;
            PP_REPORT("W1 (entry)");
#pragma unroll 1
            for (int i = 0; i + 1 < ebits; ++i) {
                if (i >= PP_RING - 1) {                               // slot (i + 1) % RING still holds index i + 1 - RING
                    const uint32_t need = (uint32_t)(i + 2 - PP_RING);
                    if (seen2 < need) PP_WAIT(seen2 = rl_wait(headB, need + 1) - 1);      // W2 is done with index headB - 2
                    if (seen3 < need) PP_WAIT(seen3 = rl_wait(tailB1, need));
                    if (seen4 < need) PP_WAIT(seen4 = rl_wait(tailB2, need));
                }
                uint32_t w[NLL];
                pp_half<GC, false, false, true>(w, x, slotA(i), none, 0, lds, 0, mq_off(slotM(i)), npp, mtrue, nblk);
                pp_store<GC>(lds, slotA(i + 1), w);
#pragma unroll
                for (int j = 0; j < NLL; ++j) x[j] = w[j];
                pp_publish(headA, (uint32_t)(i + 2));
            }
            PP_REPORT("W1");
        } else if (wave == 1) {
            // ---- W2: the second digits, one step behind ---------------------------------------------------------------
            rl_wait(headB, 1u);
            uint32_t b[NLL];
            pp_load<GC>(b, lds, slotB(0));
            uint32_t seen = 0;
            PP_T0();
#pragma unroll 1
            for (int i = 0; i + 1 < ebits; ++i) {
                PP_WAIT(rl_wait(headA, (uint32_t)(i + 2)));           // a_i and m_i
                if (i >= PP_RING - 1) {
                    const uint32_t need = (uint32_t)(i + 2 - PP_RING);
                    if (seen < need) PP_WAIT(seen = rl_wait(tailB2, need));
                }
                uint32_t b2[NLL], v[NLL];
#pragma unroll
                for (int j = 0; j < NLL; ++j) b2[j] = b[j] << 1;      // 2 a b
                pp_half<GC, false, true, false>(v, b2, slotA(i), none, 0, lds, feed_off(slotM(i)), 0, npp, mtrue, nblk);
                pp_store<GC>(lds, slotB(i + 1), v);
#pragma unroll
                for (int j = 0; j < NLL; ++j) b[j] = v[j];
                pp_publish(headB, (uint32_t)(i + 2));
            }
            PP_REPORT("W2");
        } else if (wave == 2) {
            // ---- B1: first digit of the accumulator: the a_i at the set bits of s - 1 -----------------------------------
            uint32_t A[NLL];
            int i = next_set(0), k = 0;
            pp_publish(tailB1, (uint32_t)i);
            bool first = true;
            uint32_t seenP = 0;
            PP_T0();
#pragma unroll 1
            while (i < ebits) {
                PP_WAIT(rl_wait<RL_SLEEP_B>(headA, (uint32_t)(i + 1)));
                if (first) {
                    pp_load<GC>(A, lds, slotA(i));
                    first = false;
                } else {
                    if (k >= PP_PRING - 1) {                          // (one slot stays free for the final A)
                        const uint32_t need = (uint32_t)(k + 2 - PP_PRING);
                        if (seenP < need) seenP = rl_wait(tailP, need);
                    }
                    pp_store<GC>(lds, slotPA(k), A);
                    uint32_t w[NLL];
                    pp_half<GC, false, false, true>(w, A, slotA(i), none, 0, lds, 0, mq_off(slotPM(k)), npp, mtrue, nblk);
#pragma unroll
                    for (int j = 0; j < NLL; ++j) A[j] = w[j];
                    ++k;
                    pp_publish(headP, (uint32_t)k);
                }
                i = next_set(i + 1);
                pp_publish(tailB1, (uint32_t)(i < ebits ? i : ebits + PP_RING));
            }
            if (k >= PP_PRING - 1) {
                const uint32_t need = (uint32_t)(k + 2 - PP_PRING);
                if (seenP < need) seenP = rl_wait(tailP, need);
            }
            pp_store<GC>(lds, slotPA(k), A);                          // the final first digit
            pp_publish(headP, (uint32_t)(k + 1));
            PP_REPORT("B1");
        } else {
            // ---- B2: second digit of the accumulator, then the way out ---------------------------------------------------
            uint32_t Bv[NLL];
            int i = next_set(0), k = 0;
            pp_publish(tailB2, (uint32_t)i);
            bool first = true;
            PP_T0();
            PP_DECL(pp_s1, pp_s2);
#pragma unroll 1
            while (i < ebits) {
                PP_WAIT(rl_wait<RL_SLEEP_B>(headB, (uint32_t)(i + 1)));
                if (first) {
                    pp_load<GC>(Bv, lds, slotB(i));
                    first = false;
                } else {
                    PP_WAIT(rl_wait<RL_SLEEP_B>(headP, (uint32_t)(k + 1)));
                    uint32_t Ao[NLL], v[NLL];
                    pp_load<GC>(Ao, lds, slotPA(k));
                    // v = (A b_i + B a_i - m + R s' + m' s') / R
                    pp_half<GC, true, true, false>(v, Bv, slotA(i), Ao, slotB(i), lds, feed_off(slotPM(k)), 0, npp, mtrue, nblk);
#pragma unroll
                    for (int j = 0; j < NLL; ++j) Bv[j] = v[j];
                    ++k;
                    pp_publish(tailP, (uint32_t)k);
                }
                i = next_set(i + 1);
                pp_publish(tailB2, (uint32_t)(i < ebits ? i : ebits + PP_RING));
            }
            PP_STAMP(pp_s1);
            rl_wait<RL_SLEEP_B>(headP, (uint32_t)(k + 1));
            uint32_t acc2[GN];
            if (VAR && first) {                                       // exponent 0
                set_plain_one<G>(acc2);
            } else {
            uint32_t A[NLL];
            pp_load<GC>(A, lds, slotPA(k));
            // y = A + Bv s' (plain, 2 r limbs): r rows retire the low limbs through lane 0, the window keeps y >> 29 r
            {
                using RW = Rows<NLL, U, 64>;
                uint64_t acc[RW::NW];
#pragma unroll
                for (int j = 0; j < NLL; ++j) acc[j] = A[j];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[NLL + u] = 0;
                NmRegs<1> nonm{};
                int since = 0;
#pragma unroll 1
                for (int blk = 0; blk < nblk; ++blk) {
                    uint32_t bv[U], low[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) bv[u] = lds[L::MLIM + blk * U + u];
                    RW::template block<true, false>(acc, Bv, bv, nonm, 0u, low);
                    if (lane0) {
#pragma unroll
                        for (int u = 0; u < U; ++u) lds[L::YBUF + blk * U + u] = low[u];
                    }
                    if (++since == 8 && blk != nblk - 1) { RW::normalize(acc); since = 0; }
                }
                uint32_t hi[NLL];
                RW::finish(acc, hi);
#pragma unroll
                for (int j = 0; j < NLL; ++j) {
                    if (r + NLL * lane + j < PP_YBUF) lds[L::YBUF + r + NLL * lane + j] = hi[j];
                }
            }
            wave_lds_fence();
            PP_STAMP(pp_s2);
            // into the Montgomery form of the s^2 context chunk by chunk (y = sum_j y_j R_sq^j, y_j < R_sq: every product
            // comes out lazy), then k_dec_a_rl's tail: leave the form, reduce modulo s^2 itself
            const MontCtx* cs = P.sq[which];
            NmRegs<GN> nsq;
#pragma unroll
            for (int j = 0; j < GN; ++j) nsq.v[j] = cs->npp[GN * lane + j];
            const int rows_sq = (int)cs->rows, nblk_sq = rows_sq / GU;
#pragma unroll
            for (int j = 0; j < GN; ++j) acc2[j] = 0;
#pragma unroll 1
            for (int ch = 0; ch < P.nch; ++ch) {
                uint32_t y[GN], t[GN];
#pragma unroll
                for (int j = 0; j < GN; ++j) {
                    const int li = GN * lane + j, idx = ch * rows_sq + li;
                    y[j] = (li < rows_sq && idx < PP_YBUF) ? lds[L::YBUF + idx] : 0u;
                }
                mont_mul_m1<GN, GU, 64>(t, y, lds + L::KX + ch * G::NL, 1, nsq, nblk_sq);
                add_limbs<G>(acc2, t);
            }
            uint32_t one[GN];
            set_plain_one<G>(one);
            mm_times<G>(acc2, one, lds + L::STAGE, nsq, (uint32_t)nblk_sq);
            m1_reduce_to_true_modulus<G>(acc2, lds + L::STAGE, P.fin[which]);
            }
            store_elem<G>(acc2, u_out + ((size_t)which * n + ei) * P.u_words, P.u_words, lds + L::STAGE);
            PP_REPORT2("B2 (stamps: products done, a + b s' done)", pp_s1, pp_s2);
        }
        __syncthreads();
    }
}

template <class G, class GC>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_dec_a_pp(DecPPParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out /*[2][n][u_words]*/, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pp_chain<G, GC, false>(P, ct, u_out, n, lds);
}
template <class G, class GC>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_ctmul_pp(DecPPParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ out /*[n][u_words]*/, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pp_chain<G, GC, true>(P, ct, out, n, lds);
}

}  // namespace pai
