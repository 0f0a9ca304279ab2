/* AVX512-IFMA multi-buffer Montgomery exponentiation (TEST INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg and
 * tests/test_oracle.py).  Included at the end of oracle/paillier_ref.c when the compiler targets a CPU with
 * AVX512-IFMA (gcc -march=native defines __AVX512IFMA__), so that it shares that file's scalar helpers.
 *
 * This is the CPU algorithm the reference's README.md:32 names — IPP-Crypto's mbx_exp{1024,2048,3072,4096}_mb8,
 * reached through ipcl::modExp (bindings/ipcl_bindings_classes.cpp:57,130,325) — restated from its published
 * description, in our own code (IPP-Crypto is not in /root/reference and cannot be fetched):
 *   - eight independent exponentiations run side by side, one per 64-bit lane of a 512-bit register
 *     ("multi-buffer", mb8); limb i of all eight operands forms one __m512i;
 *   - radix 2^52 limbs so that vpmadd52luq / vpmadd52huq (52x52 -> low / high 52 bits, accumulated into 64-bit
 *     lanes) do the multiply-accumulate with 12 bits of lazy-carry headroom per lane;
 *   - "almost" Montgomery multiplication: R = 2^(52 L) > 4 M, operands and results stay below 2 M and no
 *     conditional subtraction happens between products (one canonicalisation at the end);
 *   - fixed 5-bit windows over a 32-entry table per lane (the width README.md:32's kernel uses).
 * A dedicated squaring (every limb pair once, doubled, then a separate reduction) saves a quarter of the
 * multiply instructions, as IPP-Crypto's ifma_ams52x* routines do.
 * Results are checked bit for bit against the scalar 64-bit CIOS path of this file's host (tests/test_oracle.py).
 */
#include <immintrin.h>

#define IFMA_MAXL 160 /* 52-bit limbs: 8192-bit moduli (n^2 of a 4096-bit key) + 2 bits; lazy sums stay < 2^52 * 4 * 160 < 2^62 */
#define IFMA_WIN 5
typedef __m512i v8;

typedef struct {
    int L;                 /* 52-bit limbs, 52 L >= bits(M) + 2 */
    int L64;               /* 64-bit limbs of the packed form */
    v8 n[IFMA_MAXL];       /* modulus, broadcast to the 8 lanes */
    v8 r2[IFMA_MAXL];      /* R^2 mod M */
    v8 k0;                 /* -M^-1 mod 2^52 */
} ifma_ctx;

static const u64 M52 = (1ull << 52) - 1;

/* packed 64-bit limbs -> 52-bit limbs */
static void to52(u64* d, int L, const u64* s, int L64) {
    for (int i = 0; i < L; ++i) {
        const int bit = 52 * i, k = bit >> 6, sh = bit & 63;
        u64 v = k < L64 ? s[k] >> sh : 0;
        if (sh > 12 && k + 1 < L64) v |= s[k + 1] << (64 - sh);
        d[i] = v & M52;
    }
}
/* normalised 52-bit limbs -> packed 64-bit limbs */
static void from52(u64* d, int L64, const u64* s, int L) {
    memset(d, 0, sizeof(u64) * L64);
    for (int i = 0; i < L; ++i) {
        const int bit = 52 * i, k = bit >> 6, sh = bit & 63;
        if (k < L64) d[k] |= s[i] << sh;
        if (sh > 12 && k + 1 < L64) d[k + 1] |= s[i] >> (64 - sh);
    }
}

static void ifma_ctx_init(ifma_ctx* c, int bits, const u64* n64, int L64, const u64* r2_52 /* R^2 mod M as 52-bit limbs */,
                          u64 k0) {
    c->L = (bits + 2 + 51) / 52;
    c->L64 = L64;
    u64 t[IFMA_MAXL];
    to52(t, c->L, n64, L64);
    for (int i = 0; i < c->L; ++i) {
        c->n[i] = _mm512_set1_epi64((long long)t[i]);
        c->r2[i] = _mm512_set1_epi64((long long)r2_52[i]);
    }
    c->k0 = _mm512_set1_epi64((long long)k0);
}

/* r = a * b * R^-1 mod M (almost: < 2 M for a, b < 2 M).  Limbs of a, b are < 2^52; r is normalised.
 * One pass per limb of b: the running sum moves down one limb per row (the low limb is a multiple of 2^52 after
 * q * M has been added) and carries stay lazy in the 12 spare bits — at most four 52-bit summands per limb and
 * row, at most 160 rows => < 2^62. */
static inline void ifma_amm(v8* r, const v8* a, const v8* b, const ifma_ctx* c) {
    const int L = c->L;
    const v8 mask = _mm512_set1_epi64((long long)M52), zero = _mm512_setzero_si512();
    v8 acc[IFMA_MAXL + 1];
    for (int j = 0; j <= L; ++j) acc[j] = zero;
    for (int i = 0; i < L; ++i) {
        const v8 bi = b[i];
        v8 t0 = _mm512_madd52lo_epu64(acc[0], a[0], bi);
        const v8 q = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t0, c->k0), mask);
        t0 = _mm512_madd52lo_epu64(t0, c->n[0], q);                      /* low 52 bits are zero now */
        v8 carry = _mm512_srli_epi64(t0, 52);
        for (int j = 1; j < L; ++j) {
            v8 t = acc[j];
            t = _mm512_madd52lo_epu64(t, a[j], bi);
            t = _mm512_madd52hi_epu64(t, a[j - 1], bi);
            t = _mm512_madd52lo_epu64(t, c->n[j], q);
            t = _mm512_madd52hi_epu64(t, c->n[j - 1], q);
            acc[j - 1] = j == 1 ? _mm512_add_epi64(t, carry) : t;
        }
        v8 t = acc[L];
        t = _mm512_madd52hi_epu64(t, a[L - 1], bi);
        t = _mm512_madd52hi_epu64(t, c->n[L - 1], q);
        acc[L - 1] = L == 1 ? _mm512_add_epi64(t, carry) : t;
        acc[L] = zero;
    }
    v8 cy = zero;
    for (int j = 0; j < L; ++j) {
        const v8 t = _mm512_add_epi64(acc[j], cy);
        r[j] = _mm512_and_si512(t, mask);
        cy = _mm512_srli_epi64(t, 52);
    }
}

/* r = a^2 * R^-1 mod M: every limb pair once, doubled, the squares of the limbs on top, then a separate Montgomery
 * reduction (3 L^2 instead of 4 L^2 multiply instructions). */
static inline void ifma_ams(v8* r, const v8* a, const ifma_ctx* c) {
    const int L = c->L;
    const v8 mask = _mm512_set1_epi64((long long)M52), zero = _mm512_setzero_si512();
    v8 res[2 * IFMA_MAXL + 1];
    for (int j = 0; j <= 2 * L; ++j) res[j] = zero;
    for (int i = 0; i < L - 1; ++i) {                                   /* off-diagonal products a_i * a_j, j > i */
        const v8 ai = a[i];
        v8 hi_prev = zero;
        for (int j = i + 1; j < L; ++j) {
            v8 t = _mm512_add_epi64(res[i + j], hi_prev);
            t = _mm512_madd52lo_epu64(t, a[j], ai);
            hi_prev = _mm512_madd52hi_epu64(zero, a[j], ai);
            res[i + j] = t;
        }
        res[i + L] = _mm512_add_epi64(res[i + L], hi_prev);
    }
    for (int j = 0; j < 2 * L; ++j) res[j] = _mm512_slli_epi64(res[j], 1);
    for (int i = 0; i < L; ++i) {                                       /* diagonal */
        res[2 * i] = _mm512_madd52lo_epu64(res[2 * i], a[i], a[i]);
        res[2 * i + 1] = _mm512_madd52hi_epu64(res[2 * i + 1], a[i], a[i]);
    }
    for (int i = 0; i < L; ++i) {                                       /* reduction, one row per low limb */
        const v8 q = _mm512_and_si512(_mm512_madd52lo_epu64(zero, res[i], c->k0), mask);
        v8 t0 = _mm512_madd52lo_epu64(res[i], c->n[0], q);
        v8 hi_prev = _mm512_madd52hi_epu64(zero, c->n[0], q);
        res[i + 1] = _mm512_add_epi64(res[i + 1], _mm512_srli_epi64(t0, 52));
        for (int j = 1; j < L; ++j) {
            v8 t = _mm512_add_epi64(res[i + j], hi_prev);
            t = _mm512_madd52lo_epu64(t, c->n[j], q);
            hi_prev = _mm512_madd52hi_epu64(zero, c->n[j], q);
            res[i + j] = t;
        }
        res[i + L] = _mm512_add_epi64(res[i + L], hi_prev);
    }
    v8 cy = zero;
    for (int j = 0; j < L; ++j) {
        const v8 t = _mm512_add_epi64(res[L + j], cy);
        r[j] = _mm512_and_si512(t, mask);
        cy = _mm512_srli_epi64(t, 52);
    }
}

/* out_k = base_k ^ e_k mod M for the 8 lanes.  base: 8 rows of L64 packed limbs (< M); e: 8 rows of e_stride limbs
 * (e_stride 0 = one shared exponent); out: 8 rows of L64 limbs, canonical. */
static void ifma_exp_mb8(u64* out, const u64* base, const u64* e, int e_stride, int ebits, const ifma_ctx* c,
                         const u64* n64) {
    const int L = c->L, L64 = c->L64;
    static __thread v8 tbl[1 << IFMA_WIN][IFMA_MAXL];
    v8 x[IFMA_MAXL], one[IFMA_MAXL];
    u64 lane[8][IFMA_MAXL] __attribute__((aligned(64)));
    for (int k = 0; k < 8; ++k) to52(lane[k], L, base + (size_t)k * L64, L64);
    for (int j = 0; j < L; ++j)
        x[j] = _mm512_set_epi64((long long)lane[7][j], (long long)lane[6][j], (long long)lane[5][j], (long long)lane[4][j],
                                (long long)lane[3][j], (long long)lane[2][j], (long long)lane[1][j], (long long)lane[0][j]);
    for (int j = 0; j < L; ++j) one[j] = _mm512_setzero_si512();
    one[0] = _mm512_set1_epi64(1);
    ifma_amm(tbl[0], one, c->r2, c);                                    /* R mod M */
    ifma_amm(tbl[1], x, c->r2, c);                                      /* base * R */
    for (int k = 2; k < (1 << IFMA_WIN); ++k) {
        if ((k & 1) == 0) ifma_ams(tbl[k], tbl[k / 2], c);
        else ifma_amm(tbl[k], tbl[k - 1], tbl[1], c);
    }
    const int nwin = (ebits + IFMA_WIN - 1) / IFMA_WIN;
    const v8 lane_id = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
    for (int wi = nwin - 1; wi >= 0; --wi) {
        u64 wv[8];
        for (int k = 0; k < 8; ++k) {
            const u64* ek = e + (size_t)k * e_stride;
            unsigned v = 0;
            for (int b = IFMA_WIN - 1; b >= 0; --b) {
                const int bit = wi * IFMA_WIN + b;
                v = (v << 1) | (bit < ebits ? (unsigned)((ek[bit >> 6] >> (bit & 63)) & 1) : 0u);
            }
            wv[k] = v;
        }
        /* per-lane table entry: element index (w_k * IFMA_MAXL + j) * 8 + k of the u64 view of tbl */
        const v8 w = _mm512_loadu_si512((const void*)wv);
        const v8 idx0 = _mm512_add_epi64(_mm512_slli_epi64(_mm512_mullo_epi64(w, _mm512_set1_epi64(IFMA_MAXL)), 3), lane_id);
        v8 y[IFMA_MAXL];
        for (int j = 0; j < L; ++j)
            y[j] = _mm512_i64gather_epi64(_mm512_add_epi64(idx0, _mm512_set1_epi64(8 * j)), (const void*)tbl, 8);
        if (wi == nwin - 1) {
            for (int j = 0; j < L; ++j) x[j] = y[j];
        } else {
            for (int s = 0; s < IFMA_WIN; ++s) ifma_ams(x, x, c);
            ifma_amm(x, x, y, c);                                        /* entry 0 is R mod M: a multiplication by one */
        }
    }
    ifma_amm(x, x, one, c);                                              /* leave the Montgomery domain: < 2 M... */
    for (int j = 0; j < L; ++j) tbl[0][j] = x[j];                        /* table no longer needed: reuse as scratch */
    for (int k = 0; k < 8; ++k) {
        u64 l52[IFMA_MAXL], p[MAXL];
        for (int j = 0; j < L; ++j) l52[j] = ((const u64*)&tbl[0][j])[k];
        from52(p, L64, l52, L);
        if (cmp_n(p, n64, L64) >= 0) sub_n(p, p, n64, L64);             /* ... so one conditional subtraction */
        memcpy(out + (size_t)k * L64, p, sizeof(u64) * L64);
    }
}

int orc_ifma_available(void) { return __builtin_cpu_supports("avx512ifma") ? 1 : 0; }

/* out[i] = base[i]^e mod M (e shared: e_stride 0, or per element), 8 elements per call of the mb8 kernel */
int orc_ifma_modexp_batch(int N, int bits, int L64, const u64* n, const u64* r2_52, u64 k0, const u64* base, const u64* e,
                          int e_stride, int ebits, u64* out, int threads) {
    if ((bits + 2 + 51) / 52 > IFMA_MAXL || L64 > MAXL) return -1;
    ifma_ctx c;
    ifma_ctx_init(&c, bits, n, L64, r2_52, k0);
    const int groups = (N + 7) / 8;
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads_or(threads))
    for (int g = 0; g < groups; ++g) {
        u64 b8[8 * MAXL], e8[8 * MAXL], o8[8 * MAXL];
        const int es = e_stride ? e_stride : 0;
        for (int k = 0; k < 8; ++k) {
            const int i = g * 8 + k < N ? g * 8 + k : N - 1;
            memcpy(b8 + (size_t)k * L64, base + (size_t)i * L64, sizeof(u64) * L64);
            if (es) memcpy(e8 + (size_t)k * es, e + (size_t)i * es, sizeof(u64) * es);
        }
        ifma_exp_mb8(o8, b8, es ? e8 : e, es, ebits, &c, n);
        for (int k = 0; k < 8 && g * 8 + k < N; ++k) memcpy(out + (size_t)(g * 8 + k) * L64, o8 + (size_t)k * L64, sizeof(u64) * L64);
    }
    return 0;
}

typedef struct {
    int bits;
    const u64* r2_52;
    u64 k0;
} ifma_mod;

/* DJN encryption with the obfuscator hs^r on the mb8 kernel; the cheap tail ((1 + m n) * obf mod n^2) stays scalar. */
int orc_ifma_encrypt_djn_batch(int N, int Ln, const u64* n, const u64* nsq, u64 nsq0inv, const u64* nsq_r2, const u64* hs,
                               const u64* m, const u64* r, int Lr, int rbits, u64* ct, const ifma_mod* im, int threads) {
    const int L2 = 2 * Ln;
    if (L2 > MAXL || (im->bits + 2 + 51) / 52 > IFMA_MAXL) return -1;
    mctx c = {L2, nsq, nsq0inv, nsq_r2, NULL};
    ifma_ctx ic;
    ifma_ctx_init(&ic, im->bits, nsq, L2, im->r2_52, im->k0);
    const int groups = (N + 7) / 8;
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads_or(threads))
    for (int g = 0; g < groups; ++g) {
        u64 b8[8 * MAXL], e8[8 * MAXL], o8[8 * MAXL];
        for (int k = 0; k < 8; ++k) {
            const int i = g * 8 + k < N ? g * 8 + k : N - 1;
            memcpy(b8 + (size_t)k * L2, hs, sizeof(u64) * L2);
            memcpy(e8 + (size_t)k * Lr, r + (size_t)i * Lr, sizeof(u64) * Lr);
        }
        ifma_exp_mb8(o8, b8, e8, Lr, rbits, &ic, nsq);
        for (int k = 0; k < 8 && g * 8 + k < N; ++k) {
            const int i = g * 8 + k;
            u64 c0[MAXL + 1], t[MAXL];
            mul_full(c0, m + (size_t)i * Ln, Ln, n, Ln);
            for (int w = 0; w < L2; ++w) {
                if (++c0[w]) break;
            }
            mont_mul(t, c0, o8 + (size_t)k * L2, &c);
            mont_mul(ct + (size_t)i * L2, t, nsq_r2, &c);
        }
    }
    return 0;
}

/* CRT decryption with the two half-size exponentiations (ct mod s^2)^(s-1) on the mb8 kernel */
int orc_ifma_decrypt_crt_batch(int N, int Ln, int Lh, const prime_ctx* pc, const prime_ctx* qc, const u64* pinvqR,
                               const u64* ct, u64* m_out, const ifma_mod* imp, const ifma_mod* imq, int threads) {
    if (Ln > MAXL) return -1;
    const prime_ctx* pcs[2] = {pc, qc};
    const ifma_mod* ims[2] = {imp, imq};
    ifma_ctx ic[2];
    for (int w = 0; w < 2; ++w) ifma_ctx_init(&ic[w], ims[w]->bits, pcs[w]->s2, Ln, ims[w]->r2_52, ims[w]->k0);
    const int groups = (N + 7) / 8;
#pragma omp parallel for schedule(dynamic, 2) num_threads(nthreads_or(threads))
    for (int g = 0; g < groups; ++g) {
        u64 ms[2][8][MAXL];
        for (int w = 0; w < 2; ++w) {
            const prime_ctx* k = pcs[w];
            mctx c2 = {Ln, k->s2, k->s2_0inv, k->s2_r2, NULL};
            mctx c1 = {Lh, k->s, k->s_0inv, k->s_r2, NULL};
            u64 x8[8 * MAXL], u8_[8 * MAXL];
            for (int j = 0; j < 8; ++j) {
                const int i = g * 8 + j < N ? g * 8 + j : N - 1;
                reduce_wide(x8 + (size_t)j * Ln, ct + (size_t)i * 2 * Ln, &c2);
            }
            ifma_exp_mb8(u8_, x8, k->e, 0, k->ebits, &ic[w], k->s2);
            for (int j = 0; j < 8; ++j) {
                u64 um1[MAXL], l[MAXL];
                memcpy(um1, u8_ + (size_t)j * Ln, sizeof(u64) * Lh);
                for (int t = 0; t < Lh; ++t) {
                    if (um1[t]--) break;
                }
                mul_low(l, um1, k->sinv2, Lh);
                mont_mul(ms[w][j], l, k->hR, &c1);
            }
        }
        mctx cq = {Lh, qc->s, qc->s_0inv, qc->s_r2, NULL};
        for (int j = 0; j < 8 && g * 8 + j < N; ++j) {
            u64 d[MAXL], t[MAXL], prod[2 * MAXL];
            if (sub_n(d, ms[1][j], ms[0][j], Lh)) add_n(d, d, qc->s, Lh);
            mont_mul(t, d, pinvqR, &cq);
            mul_full(prod, pc->s, Lh, t, Lh);
            u64 carry = 0;
            for (int w = 0; w < 2 * Lh; ++w) {
                u128 s = (u128)prod[w] + (w < Lh ? ms[0][j][w] : 0) + carry;
                prod[w] = (u64)s;
                carry = (u64)(s >> 64);
            }
            u64* mo = m_out + (size_t)(g * 8 + j) * Ln;
            for (int w = 0; w < Ln; ++w) mo[w] = w < 2 * Lh ? prod[w] : 0;
        }
    }
    return 0;
}
