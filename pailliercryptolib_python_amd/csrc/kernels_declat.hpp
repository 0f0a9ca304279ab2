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
// The base is s' = s k with k = -s^-1 mod 2^(29 U) (a minus-one context, Rows::block_m1: the quotient digits of a row block
// ARE the limbs it retires); everything is arithmetic modulo s'^2, of which s^2 is a divisor, so the power modulo s^2 — what
// stage B expects — comes out of one plain product a + b s', one reduction and the conventional tail of k_dec_a_rl.
// One (ciphertext, prime) per workgroup; an integer is 3 limbs x 64 lanes (geo_3x64).
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
constexpr int PP_ROW = 192;      // words per ring slot: one lane-sliced integer of the 3 x 64 geometry (loads / stores without bounds)
constexpr int PP_DSTR = 81;      // per-lane stride of the export dump (odd: conflict-free)
constexpr int PP_RING = 16;      // (a_i, m_i, b_i) slots between the squaring waves and the product waves
constexpr int PP_PRING = 8;      // (A, m) slots between the product waves
constexpr int PP_MAXND = 6;
constexpr int PP_MAXCH = 3;      // base-R_sq chunks of a + b s' at the exit
constexpr int PP_YBUF = 2 * PP_RMAX + 16;
constexpr int PP_EWORDS = 128;   // exponent words (4096 bits)


template <class G>
struct PPLds {
    static constexpr int STAGE = 0;                                   // G::LDS_WORDS: operand staging of the exit products
    static constexpr int YBUF = STAGE + G::LDS_WORDS;                 // a + b s' assembled (2 r + 2 limbs)
    static constexpr int RING_A = YBUF + PP_YBUF;
    static constexpr int RING_M = RING_A + PP_RING * PP_ROW;
    static constexpr int RING_B = RING_M + PP_RING * PP_RMAX;
    static constexpr int PRING_A = RING_B + PP_RING * PP_ROW;
    static constexpr int PRING_M = PRING_A + PP_PRING * PP_ROW;
    static constexpr int DUMP = PRING_M + PP_PRING * PP_RMAX;         // where the lanes other than 0 drop their "quotient digits"
    static constexpr int KD = DUMP + 64 * PP_DSTR;                    // [nd][2][RMAX]
    static constexpr int MLIM = KD + PP_MAXND * 2 * PP_RMAX;          // limbs of s'
    static constexpr int KX = MLIM + PP_RMAX;                         // exit constants, [nch][NL]
    static constexpr int ZERO = KX + PP_MAXCH * G::NL;                          // RMAX zero words (the feed of lanes != 0)
    static constexpr int TMPM = ZERO + PP_RMAX;                       // quotient digits of the entry products
    static constexpr int EXPO = TMPM + PP_RMAX;                       // the exponent's words (the product waves scan its bits)
    static constexpr int FLAGS = EXPO + PP_EWORDS;
    static constexpr int WORDS = FLAGS + 16;
    static constexpr int BYTES = WORDS * 4;
};

// One half of the product rule on a minus-one context, T = 64 (quotient digits through SGPRs):
//   r = (x1 * dig1 [+ x2 * dig2] [+ R - m] + q s') / R  [+ s' - 1]
// x1, x2: this lane's limb slices; dig1, dig2: LDS limbs (stride 1, nblk * U of them); FEED: m arrives as complemented digits
// (2^29 - 1 - m_i) at lds[fd_off ..] for lane 0 and as zeros for the other lanes (fd_off is per lane); EXPORT: this half's own
// quotient digits leave, complemented, to mq (written by lane 0).  HEAVY: operands that make three 2^58 products per row
// and column (two products, or a doubled operand): normalise more often.
// One row block of pp_half on a window split into its low half L (the U columns the block retires) and its high half H
// (NLL == U): afterwards H is the new low half and L — cleared, then holding the first q s' products — the new high half, so two
// consecutive blocks swap the roles of the two register sets and the window never moves.
// The block's own digits arrive in `cur` (read from LDS during the previous block); the next block's digits are read into
// `nxt` here, before the multiplies — a lone wave has nobody to hide an LDS round trip behind, and the two digit sets swap
// roles from block to block like the window halves do (no copies).
template <int U>
struct PPDigits { uint32_t b1[U], b2[U], f[U]; };
template <class G, bool TWO, bool FEED>
PAI_DEV void pp_fetch(PPDigits<G::U>& d, const uint32_t* lds, int d1, int d2, int fd) {
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
        d.b1[u] = lds[d1 + u];
        if constexpr (TWO) d.b2[u] = lds[d2 + u];
        if constexpr (FEED) d.f[u] = lds[fd + u];
    }
}
template <class G, bool TWO, bool FEED, bool EXPORT>
PAI_DEV void pp_block(uint64_t (&L)[G::U], uint64_t (&H)[G::U], const uint32_t (&x1)[G::NLL], const uint32_t (&x2)[G::NLL], uint32_t* lds,
                      const PPDigits<G::U>& cur, PPDigits<G::U>& nxt, int d1, int d2, int fd, int mq, const NmRegs<G::NLL>& npp) {
    constexpr int U = G::U;
    static_assert(G::NLL == U, "window of two halves");
    pp_fetch<G, TWO, FEED>(nxt, lds, d1 + U, d2 + U, fd + U);       // (one block beyond the last: in-bounds reads of unused words)
    const uint32_t (&bv1)[U] = cur.b1;
    const uint32_t (&bv2)[U] = cur.b2;
    const uint32_t (&fv)[U] = cur.f;
    auto col = [&](int k) -> uint64_t& { return k < U ? L[k] : H[k - U]; };
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            col(j + u) += (uint64_t)x1[j] * bv1[u];
            if constexpr (TWO) col(j + u) += (uint64_t)x2[j] * bv2[u];
        }
    }
    if constexpr (FEED) {
#pragma unroll
        for (int u = 0; u < U; ++u) L[u] += (uint64_t)fv[u];
    }
    uint32_t low[U], q[U];
    uint64_t c = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t t = L[u] + c;
        low[u] = (uint32_t)t & RMASK;
        c = t >> RB;
    }
    H[0] += c;
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)low[u]);
    if constexpr (EXPORT) {
        // every lane stores (no exec masking inside the block): lane 0 to the consumer's buffer, the others to a dump
#pragma unroll
        for (int u = 0; u < U; ++u) lds[mq + u] = RMASK - low[u];
    }
    // the window moves up by U columns: H is the low half now, L the (empty) high half
#pragma unroll
    for (int u = 0; u < U; ++u) {
        L[u] = 0;
        H[u] += (uint64_t)from_next<64>(low[u]);
    }
    auto ncol = [&](int k) -> uint64_t& { return k < U ? H[k] : L[k - U]; };
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int j = 0; j < U; ++j) ncol(j + u) += (uint64_t)npp.v[j] * q[u];
    }
}
// Rows::normalize on the split window (low half first)
template <int U>
PAI_DEV void pp_normalize(uint64_t (&L)[U], uint64_t (&H)[U]) {
    uint64_t w[2 * U];
#pragma unroll
    for (int u = 0; u < U; ++u) { w[u] = L[u]; w[U + u] = H[u]; }
#pragma unroll
    for (int k = 2 * U - 1; k >= 1; --k) w[k] = ((k == 2 * U - 1) ? w[k] : (w[k] & RMASK)) + (w[k - 1] >> RB);
    w[0] &= RMASK;
#pragma unroll
    for (int u = 0; u < U; ++u) { L[u] = w[u]; H[u] = w[U + u]; }
}

template <class G, bool TWO, bool FEED, bool EXPORT, bool HEAVY>
PAI_DEV void pp_half(uint32_t (&r)[G::NLL], const uint32_t (&x1)[G::NLL], int dig1, const uint32_t (&x2)[G::NLL], int dig2,
                     uint32_t* lds, int fd_off, int mq_off, const NmRegs<G::NLL>& npp, const uint32_t (&mtrue)[G::NLL], int nblk) {
    // (every LDS operand is an OFFSET from the one base pointer: a run-time choice between LDS pointers costs the address space)
    constexpr int NLL = G::NLL, U = G::U;
    using RW = Rows<NLL, U, 64>;
    uint64_t A[U], B[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { A[u] = 0; B[u] = 0; }
    const bool lane0 = (threadIdx.x & 63) == 0;
    if (FEED && lane0) A[0] = 1;                                      // R - m = sum (2^29 - 1 - m_i) 2^(29 i) + 1
    // rows between normalisations: 2^58 per product and row — three products (two operand pairs, or a doubled operand, plus the
    // quotient's) allow 21 rows, two allow 30
    constexpr int NORMB = (TWO || HEAVY) ? 21 / U : 30 / U;
    int since = 0;
    int blk = 0;
    PPDigits<U> da, db;
    pp_fetch<G, TWO, FEED>(da, lds, dig1, dig2, fd_off);
#pragma unroll 1
    for (; blk + 1 < nblk; blk += 2) {
        pp_block<G, TWO, FEED, EXPORT>(A, B, x1, x2, lds, da, db, dig1 + blk * U, dig2 + blk * U, fd_off + blk * U, mq_off + blk * U, npp);
        if (++since == NORMB) { pp_normalize<U>(B, A); since = 0; }
        pp_block<G, TWO, FEED, EXPORT>(B, A, x1, x2, lds, db, da, dig1 + (blk + 1) * U, dig2 + (blk + 1) * U, fd_off + (blk + 1) * U,
                                       mq_off + (blk + 1) * U, npp);
        if (++since == NORMB && blk + 2 < nblk) { pp_normalize<U>(A, B); since = 0; }
    }
    uint64_t acc[RW::NW];
    if (blk < nblk) {                                                 // an odd block count: the halves end up swapped
        pp_block<G, TWO, FEED, EXPORT>(A, B, x1, x2, lds, da, db, dig1 + blk * U, dig2 + blk * U, fd_off + blk * U, mq_off + blk * U, npp);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[u] = B[u]; acc[U + u] = A[u]; }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[u] = A[u]; acc[U + u] = B[u]; }
    }
    // the window's top U columns are the next lane's lowest (mont_mul_m1)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t top = acc[NLL + u];
        const uint32_t lo = from_prev<64>((uint32_t)top), hi = from_prev<64>((uint32_t)(top >> 32));
        acc[u] += ((uint64_t)hi << 32) | lo;
    }
    if constexpr (FEED) {                                             // + (s' - 1) R on the numerator
#pragma unroll
        for (int j = 0; j < NLL; ++j) acc[j] += (uint64_t)mtrue[j];
        if (lane0) acc[0] -= 1;
    }
    RW::finish(acc, r);
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
#define PP_DECL(a, b) unsigned long long a = 0, b = 0
#define PP_STAMP(var) var = __builtin_readcyclecounter() - pp_t0
#define PP_REPORT2(role, a, b) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) printf("PP %s: cycles %llu, waiting %llu; stamps %llu %llu\n", role, __builtin_readcyclecounter() - pp_t0, pp_wait, a, b); } while (0)
#define PP_REPORT(role) do { } while (0)
#define PP_STAMP(var) do { } while (0)
#define PP_DECL(a, b) do { } while (0)
#define PP_REPORT2(role, a, b) do { } while (0)
#endif

// VAR = false: decrypt stage A (workgroup (i, w): ciphertext i to the power s_w - 1 modulo s_w^2, s_0 = p, s_1 = q).
// VAR = true: ct * pt for the smallest batches (ipclCipherText.__mul__ -> CipherText::operator*(PlainText), classes.cpp /
// the reference's BM_Mul_CTPT): ONE modulus (s = n, the power modulo n^2) and the exponent of each element its own; exponent 0
// gives 1.
template <class G, bool VAR>
PAI_DEV void pp_chain(const DecPPParams& P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out, int n, uint32_t* lds) {
    static_assert(G::M1 && G::T == 64 && G::NLL == G::U && BLOCK_THREADS == 256, "one integer per wavefront, four waves per chain");
    constexpr int NLL = G::NLL, U = G::U;
    using L = PPLds<G>;
    const int which = VAR ? 0 : (int)blockIdx.y;
    const MontCtx* ctx = P.pp[which];
    const uint32_t* expo = P.expo[which];
    const int ebits = P.ebits[which];
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
    // lane 0 feeds from / exports to the quotient-digit buffer, the other lanes read zeros / write to their dump rows
    auto feed_off = [&](int m_off) -> int { return lane0 ? m_off : (int)L::ZERO; };
    auto mq_off = [&](int m_off) -> int { return lane0 ? m_off : (int)L::DUMP + lane * PP_DSTR; };
    auto slotA = [](int i) -> int { return L::RING_A + (i % PP_RING) * PP_ROW; };
    auto slotM = [](int i) -> int { return L::RING_M + (i % PP_RING) * PP_RMAX; };
    auto slotB = [](int i) -> int { return L::RING_B + (i % PP_RING) * PP_ROW; };
    auto slotPA = [](int k) -> int { return L::PRING_A + (k % PP_PRING) * PP_ROW; };
    auto slotPM = [](int k) -> int { return L::PRING_M + (k % PP_PRING) * PP_RMAX; };
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
                load_elem_off<G>(c, row, P.ct_words, r * i);
#pragma unroll
                for (int j = 0; j < NLL; ++j) c[j] = (NLL * lane + j < r) ? c[j] : 0u;
                wave_lds_fence();
                pp_half<G, false, false, true, false>(w, c, L::KD + (2 * i) * PP_RMAX, none, 0, lds, 0, mq_off(L::TMPM), npp, mtrue, nblk);
                wave_lds_fence();
                pp_half<G, false, true, false, false>(v, c, L::KD + (2 * i + 1) * PP_RMAX, none, 0, lds, feed_off(L::TMPM), 0, npp, mtrue,
                                                      nblk);
                add_limbs<G>(sa, w);
                add_limbs<G>(sb, v);
            }
            pp_store<G>(lds, slotA(0), sa);
            pp_store<G>(lds, slotB(0), sb);
            rl_publish(headB, 1u);
            rl_publish(headA, 1u);
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
                pp_half<G, false, false, true, false>(w, x, slotA(i), none, 0, lds, 0, mq_off(slotM(i)), npp, mtrue, nblk);
                pp_store<G>(lds, slotA(i + 1), w);
#pragma unroll
                for (int j = 0; j < NLL; ++j) x[j] = w[j];
                rl_publish(headA, (uint32_t)(i + 2));
            }
            PP_REPORT("W1");
        } else if (wave == 1) {
            // ---- W2: the second digits, one step behind ---------------------------------------------------------------
            rl_wait(headB, 1u);
            uint32_t b[NLL];
            pp_load<G>(b, lds, slotB(0));
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
                pp_half<G, false, true, false, true>(v, b2, slotA(i), none, 0, lds, feed_off(slotM(i)), 0, npp, mtrue, nblk);
                pp_store<G>(lds, slotB(i + 1), v);
#pragma unroll
                for (int j = 0; j < NLL; ++j) b[j] = v[j];
                rl_publish(headB, (uint32_t)(i + 2));
            }
            PP_REPORT("W2");
        } else if (wave == 2) {
            // ---- B1: first digit of the accumulator: the a_i at the set bits of s - 1 -----------------------------------
            uint32_t A[NLL];
            int i = next_set(0), k = 0;
            rl_publish(tailB1, (uint32_t)i);
            bool first = true;
            uint32_t seenP = 0;
            PP_T0();
#pragma unroll 1
            while (i < ebits) {
                PP_WAIT(rl_wait<RL_SLEEP_B>(headA, (uint32_t)(i + 1)));
                if (first) {
                    pp_load<G>(A, lds, slotA(i));
                    first = false;
                } else {
                    if (k >= PP_PRING - 1) {                          // (one slot stays free for the final A)
                        const uint32_t need = (uint32_t)(k + 2 - PP_PRING);
                        if (seenP < need) seenP = rl_wait(tailP, need);
                    }
                    pp_store<G>(lds, slotPA(k), A);
                    uint32_t w[NLL];
                    pp_half<G, false, false, true, false>(w, A, slotA(i), none, 0, lds, 0, mq_off(slotPM(k)), npp, mtrue, nblk);
#pragma unroll
                    for (int j = 0; j < NLL; ++j) A[j] = w[j];
                    ++k;
                    rl_publish(headP, (uint32_t)k);
                }
                i = next_set(i + 1);
                rl_publish(tailB1, (uint32_t)(i < ebits ? i : ebits + PP_RING));
            }
            if (k >= PP_PRING - 1) {
                const uint32_t need = (uint32_t)(k + 2 - PP_PRING);
                if (seenP < need) seenP = rl_wait(tailP, need);
            }
            pp_store<G>(lds, slotPA(k), A);                          // the final first digit
            rl_publish(headP, (uint32_t)(k + 1));
            PP_REPORT("B1");
        } else {
            // ---- B2: second digit of the accumulator, then the way out ---------------------------------------------------
            uint32_t Bv[NLL];
            int i = next_set(0), k = 0;
            rl_publish(tailB2, (uint32_t)i);
            bool first = true;
            PP_T0();
            PP_DECL(pp_s1, pp_s2);
#pragma unroll 1
            while (i < ebits) {
                PP_WAIT(rl_wait<RL_SLEEP_B>(headB, (uint32_t)(i + 1)));
                if (first) {
                    pp_load<G>(Bv, lds, slotB(i));
                    first = false;
                } else {
                    PP_WAIT(rl_wait<RL_SLEEP_B>(headP, (uint32_t)(k + 1)));
                    uint32_t Ao[NLL], v[NLL];
                    pp_load<G>(Ao, lds, slotPA(k));
                    // v = (A b_i + B a_i - m + R s' + m' s') / R
                    pp_half<G, true, true, false, false>(v, Bv, slotA(i), Ao, slotB(i), lds, feed_off(slotPM(k)), 0, npp, mtrue, nblk);
#pragma unroll
                    for (int j = 0; j < NLL; ++j) Bv[j] = v[j];
                    ++k;
                    rl_publish(tailP, (uint32_t)k);
                }
                i = next_set(i + 1);
                rl_publish(tailB2, (uint32_t)(i < ebits ? i : ebits + PP_RING));
            }
            PP_STAMP(pp_s1);
            rl_wait<RL_SLEEP_B>(headP, (uint32_t)(k + 1));
            uint32_t acc2[NLL];
            if (VAR && first) {                                       // exponent 0
                set_plain_one<G>(acc2);
            } else {
            uint32_t A[NLL];
            pp_load<G>(A, lds, slotPA(k));
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
            NmRegs<NLL> nsq;
#pragma unroll
            for (int j = 0; j < NLL; ++j) nsq.v[j] = cs->npp[NLL * lane + j];
            const int rows_sq = (int)cs->rows, nblk_sq = rows_sq / U;
#pragma unroll
            for (int j = 0; j < NLL; ++j) acc2[j] = 0;
#pragma unroll 1
            for (int ch = 0; ch < P.nch; ++ch) {
                uint32_t y[NLL], t[NLL];
#pragma unroll
                for (int j = 0; j < NLL; ++j) {
                    const int li = NLL * lane + j, idx = ch * rows_sq + li;
                    y[j] = (li < rows_sq && idx < PP_YBUF) ? lds[L::YBUF + idx] : 0u;
                }
                mont_mul_m1<NLL, U, 64>(t, y, lds + L::KX + ch * G::NL, 1, nsq, nblk_sq);
                add_limbs<G>(acc2, t);
            }
            uint32_t one[NLL];
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

template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_dec_a_pp(DecPPParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ u_out /*[2][n][u_words]*/, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pp_chain<G, false>(P, ct, u_out, n, lds);
}
template <class G>
__global__ void __launch_bounds__(BLOCK_THREADS, 1)
k_ctmul_pp(DecPPParams P, const uint32_t* __restrict__ ct, uint32_t* __restrict__ out /*[n][u_words]*/, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    pp_chain<G, true>(P, ct, out, n, lds);
}

}  // namespace pai
