/* CPU oracle in plain C (TEST INFRASTRUCTURE ONLY — never linked into, called from or shipped with
 * the product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it).
 *
 * It restates, with the textbook word-serial algorithms, the arithmetic that the reference delegates
 * to the un-vendored intel/pailliercryptolib + IPP-Crypto (see oracle/paillier_oracle.py header and
 * SURVEY.md App. D): 64-bit-limb CIOS Montgomery multiplication, fixed 5-bit-window exponentiation
 * (the window IPP-Crypto's mbx_exp uses, README.md:32), DJN/standard Paillier encryption and CRT
 * decryption.  All constants that need division (R^2, R^3, inverses) are computed by the Python
 * side with big ints and passed in, so this file contains no division routine.  Results are pinned
 * against oracle/paillier_oracle.py (CPython pow) by tests/test_oracle.py; the ciphertext bits
 * themselves are "parity unpinned" by the reference (no known-answer vectors exist upstream).
 *
 * Numbers are little-endian arrays of uint64_t limbs; batches are row-major.
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/paillier_ref.c -o oracle/_build/libpaillier_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define MAXL 136 /* up to 8704-bit moduli */
#define WIN 5

typedef struct {
    int L;          /* limbs */
    const u64* n;   /* modulus (odd) */
    u64 n0inv;      /* -n^-1 mod 2^64 */
    const u64* r2;  /* R^2 mod n, R = 2^(64 L) */
    const u64* r3;  /* R^3 mod n (may be NULL when unused) */
} mctx;

static int cmp_n(const u64* a, const u64* b, int L) {
    for (int i = L - 1; i >= 0; --i)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
static u64 sub_n(u64* r, const u64* a, const u64* b, int L) {
    u64 borrow = 0;
    for (int i = 0; i < L; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1;
    }
    return borrow;
}
static u64 add_n(u64* r, const u64* a, const u64* b, int L) {
    u64 c = 0;
    for (int i = 0; i < L; ++i) {
        u128 t = (u128)a[i] + b[i] + c;
        r[i] = (u64)t;
        c = (u64)(t >> 64);
    }
    return c;
}

/* r = a*b*R^-1 mod n, a,b < n, r < n (CIOS) */
static void mont_mul(u64* r, const u64* a, const u64* b, const mctx* c) {
    const int L = c->L;
    u64 t[MAXL + 2];
    memset(t, 0, sizeof(u64) * (L + 2));
    for (int i = 0; i < L; ++i) {
        u64 carry = 0;
        const u64 bi = b[i];
        for (int j = 0; j < L; ++j) {
            u128 s = (u128)a[j] * bi + t[j] + carry;
            t[j] = (u64)s;
            carry = (u64)(s >> 64);
        }
        u128 s = (u128)t[L] + carry;
        t[L] = (u64)s;
        t[L + 1] = (u64)(s >> 64);
        const u64 q = t[0] * c->n0inv;
        s = (u128)q * c->n[0] + t[0];
        carry = (u64)(s >> 64);
        for (int j = 1; j < L; ++j) {
            s = (u128)q * c->n[j] + t[j] + carry;
            t[j - 1] = (u64)s;
            carry = (u64)(s >> 64);
        }
        s = (u128)t[L] + carry;
        t[L - 1] = (u64)s;
        t[L] = t[L + 1] + (u64)(s >> 64);
    }
    if (t[L] || cmp_n(t, c->n, L) >= 0) sub_n(r, t, c->n, L);
    else memcpy(r, t, sizeof(u64) * L);
}

/* out = base^e mod n; base plain < n; e has ebits significant bits; out plain */
static void mont_exp(u64* out, const u64* base, const u64* e, int ebits, const mctx* c) {
    const int L = c->L;
    u64 tbl[1 << WIN][MAXL];
    u64 one[MAXL];
    memset(one, 0, sizeof(u64) * L);
    one[0] = 1;
    mont_mul(tbl[0], one, c->r2, c);   /* R mod n */
    mont_mul(tbl[1], base, c->r2, c);
    for (int k = 2; k < (1 << WIN); ++k) mont_mul(tbl[k], tbl[k - 1], tbl[1], c);
    const int nwin = (ebits + WIN - 1) / WIN;
    u64 x[MAXL];
    int started = 0;
    for (int wi = nwin - 1; wi >= 0; --wi) {
        unsigned wv = 0;
        for (int b = WIN - 1; b >= 0; --b) {
            int bit = wi * WIN + b;
            unsigned v = (bit < ebits) ? (unsigned)((e[bit >> 6] >> (bit & 63)) & 1) : 0;
            wv = (wv << 1) | v;
        }
        if (!started) {
            memcpy(x, tbl[wv], sizeof(u64) * L);
            started = 1;
        } else {
            for (int s = 0; s < WIN; ++s) mont_mul(x, x, x, c);
            if (wv) mont_mul(x, x, tbl[wv], c);
        }
    }
    if (!started) memcpy(x, tbl[0], sizeof(u64) * L);
    mont_mul(out, x, one, c);
}

/* r = t mod n for a 2L-limb t (< n*R): REDC then * R^2 */
static void reduce_wide(u64* r, const u64* t2, const mctx* c) {
    const int L = c->L;
    u64 t[2 * MAXL + 1];
    memcpy(t, t2, sizeof(u64) * 2 * L);
    t[2 * L] = 0;
    for (int i = 0; i < L; ++i) {
        const u64 q = t[i] * c->n0inv;
        u64 carry = 0;
        for (int j = 0; j < L; ++j) {
            u128 s = (u128)q * c->n[j] + t[i + j] + carry;
            t[i + j] = (u64)s;
            carry = (u64)(s >> 64);
        }
        for (int k = i + L; carry && k <= 2 * L; ++k) {
            u128 s = (u128)t[k] + carry;
            t[k] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
    u64 x[MAXL];
    if (t[2 * L] || cmp_n(t + L, c->n, L) >= 0) sub_n(x, t + L, c->n, L);
    else memcpy(x, t + L, sizeof(u64) * L);
    mont_mul(r, x, c->r2, c);    /* (t R^-1) R^2 R^-1 = t mod n */
}

static void mul_full(u64* r, const u64* a, int La, const u64* b, int Lb) {
    memset(r, 0, sizeof(u64) * (La + Lb));
    for (int i = 0; i < La; ++i) {
        u64 carry = 0;
        for (int j = 0; j < Lb; ++j) {
            u128 s = (u128)a[i] * b[j] + r[i + j] + carry;
            r[i + j] = (u64)s;
            carry = (u64)(s >> 64);
        }
        r[i + Lb] = carry;
    }
}
static void mul_low(u64* r, const u64* a, const u64* b, int L) {
    memset(r, 0, sizeof(u64) * L);
    for (int i = 0; i < L; ++i) {
        u64 carry = 0;
        for (int j = 0; j + i < L; ++j) {
            u128 s = (u128)a[i] * b[j] + r[i + j] + carry;
            r[i + j] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
}

static int nthreads_or(int t) {
#ifdef _OPENMP
    return t > 0 ? t : omp_get_max_threads();
#else
    (void)t;
    return 1;
#endif
}

int orc_max_threads(void) { return nthreads_or(0); }

/* out[i] = base[i]^e mod n ; e shared (e_stride 0) or per element (e_stride = limbs per exponent) */
int orc_modexp_batch(int N, int L, const u64* n, u64 n0inv, const u64* r2, const u64* base, const u64* e,
                     int e_stride, int ebits, u64* out, int threads) {
    if (L > MAXL) return -1;
    mctx c = {L, n, n0inv, r2, NULL};
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads_or(threads))
    for (int i = 0; i < N; ++i) mont_exp(out + (size_t)i * L, base + (size_t)i * L, e + (size_t)i * e_stride, ebits, &c);
    return 0;
}

/* out[i] = a[i]*b[i] mod n */
int orc_modmul_batch(int N, int L, const u64* n, u64 n0inv, const u64* r2, const u64* a, const u64* b, u64* out,
                     int threads) {
    if (L > MAXL) return -1;
    mctx c = {L, n, n0inv, r2, NULL};
#pragma omp parallel for schedule(static) num_threads(nthreads_or(threads))
    for (int i = 0; i < N; ++i) {
        u64 t[MAXL];
        mont_mul(t, a + (size_t)i * L, b + (size_t)i * L, &c);
        mont_mul(out + (size_t)i * L, t, r2, &c);
    }
    return 0;
}

/* DJN encryption, canonical algorithm: ct = (1 + m n) * hs^r mod n^2.
 * Ln limbs for n and m; nsq context has 2*Ln limbs; r has Lr limbs (rbits significant). */
int orc_encrypt_djn_batch(int N, int Ln, const u64* n, const u64* nsq, u64 nsq0inv, const u64* nsq_r2, const u64* hs,
                          const u64* m, const u64* r, int Lr, int rbits, u64* ct, int threads) {
    const int L2 = 2 * Ln;
    if (L2 > MAXL) return -1;
    mctx c = {L2, nsq, nsq0inv, nsq_r2, NULL};
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads_or(threads))
    for (int i = 0; i < N; ++i) {
        u64 obf[MAXL], c0[MAXL + 1], t[MAXL];
        mont_exp(obf, hs, r + (size_t)i * Lr, rbits, &c);
        mul_full(c0, m + (size_t)i * Ln, Ln, n, Ln);          /* m*n < n^2 */
        for (int k = 0; k < L2; ++k) {                           /* + 1 */
            if (++c0[k]) break;
        }
        mont_mul(t, c0, obf, &c);
        mont_mul(ct + (size_t)i * L2, t, nsq_r2, &c);
    }
    return 0;
}

/* CRT decryption.  Per prime s in {p, q}: context of s^2 (Ln limbs, with r3 unused), exponent s-1,
 * sinv2 = s^-1 mod 2^(64 Lh), context of s (Lh limbs), hR = h_s * R mod s.  pinvqR = p^-1 R mod q. */
typedef struct {
    const u64 *s2, *s2_r2;
    u64 s2_0inv;
    const u64* e;
    int ebits;
    const u64 *s, *s_r2;
    u64 s_0inv;
    const u64 *sinv2, *hR;
} prime_ctx;

int orc_decrypt_crt_batch(int N, int Ln, int Lh, const prime_ctx* pc, const prime_ctx* qc, const u64* pinvqR,
                          const u64* ct, u64* m_out, int threads) {
    if (2 * Ln > 2 * MAXL || Ln > MAXL) return -1;
    const prime_ctx* pcs[2] = {pc, qc};
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads_or(threads))
    for (int i = 0; i < N; ++i) {
        u64 ms[2][MAXL];
        for (int w = 0; w < 2; ++w) {
            const prime_ctx* k = pcs[w];
            mctx c2 = {Ln, k->s2, k->s2_0inv, k->s2_r2, NULL};
            mctx c1 = {Lh, k->s, k->s_0inv, k->s_r2, NULL};
            u64 x[MAXL], u[MAXL], um1[MAXL], l[MAXL];
            reduce_wide(x, ct + (size_t)i * 2 * Ln, &c2);
            mont_exp(u, x, k->e, k->ebits, &c2);
            memcpy(um1, u, sizeof(u64) * Lh);                    /* (u - 1) mod 2^(64 Lh) */
            for (int j = 0; j < Lh; ++j) {
                if (um1[j]--) break;
            }
            mul_low(l, um1, k->sinv2, Lh);                       /* exact quotient (u-1)/s */
            mont_mul(ms[w], l, k->hR, &c1);                      /* l * h mod s */
        }
        /* t = (mq - mp) * pinvq mod q ; m = mp + p t */
        const prime_ctx* kq = qc;
        mctx cq = {Lh, kq->s, kq->s_0inv, kq->s_r2, NULL};
        u64 d[MAXL], t[MAXL], prod[2 * MAXL];
        if (sub_n(d, ms[1], ms[0], Lh)) add_n(d, d, kq->s, Lh);
        mont_mul(t, d, pinvqR, &cq);
        mul_full(prod, pc->s, Lh, t, Lh);
        u64 carry = 0;
        for (int j = 0; j < 2 * Lh; ++j) {
            u128 s = (u128)prod[j] + (j < Lh ? ms[0][j] : 0) + carry;
            prod[j] = (u64)s;
            carry = (u64)(s >> 64);
        }
        u64* mo = m_out + (size_t)i * Ln;
        for (int j = 0; j < Ln; ++j) mo[j] = j < 2 * Lh ? prod[j] : 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Optional second opinion / stronger CPU baseline: the same two operations through libgmp
 * (mpz_powm: GMP's assembly-tuned sliding-window Montgomery exponentiation), resolved with dlopen
 * at run time so that nothing is needed at build time.  GMP is a stand-in for the reference's
 * IPP-Crypto path, which cannot be obtained here; it is the best general-purpose CPU big-integer
 * library present on the box.
 * ---------------------------------------------------------------------------------------------- */
#include <dlfcn.h>
typedef struct { int alloc; int size; unsigned long* d; } mpz_s;
typedef mpz_s mpz_tt[1];
static struct {
    void* h;
    void (*init)(mpz_s*);
    void (*clear)(mpz_s*);
    void (*import)(mpz_s*, size_t, int, size_t, int, size_t, const void*);
    void* (*export_)(void*, size_t*, int, size_t, int, size_t, const mpz_s*);
    void (*powm)(mpz_s*, const mpz_s*, const mpz_s*, const mpz_s*);
    void (*mul)(mpz_s*, const mpz_s*, const mpz_s*);
    void (*mod)(mpz_s*, const mpz_s*, const mpz_s*);
    void (*add)(mpz_s*, const mpz_s*, const mpz_s*);
    void (*sub)(mpz_s*, const mpz_s*, const mpz_s*);
    void (*add_ui)(mpz_s*, const mpz_s*, unsigned long);
    void (*sub_ui)(mpz_s*, const mpz_s*, unsigned long);
    void (*divexact)(mpz_s*, const mpz_s*, const mpz_s*);
} G;

int orc_gmp_available(void) {
    if (G.h) return 1;
    const char* names[] = {"libgmp.so.10", "libgmp.so", NULL};
    void* h = NULL;
    for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return 0;
#define SYM(field, name) *(void**)(&G.field) = dlsym(h, name); if (!G.field) return 0
    SYM(init, "__gmpz_init"); SYM(clear, "__gmpz_clear"); SYM(import, "__gmpz_import"); SYM(export_, "__gmpz_export");
    SYM(powm, "__gmpz_powm"); SYM(mul, "__gmpz_mul"); SYM(mod, "__gmpz_mod"); SYM(add, "__gmpz_add");
    SYM(sub, "__gmpz_sub"); SYM(add_ui, "__gmpz_add_ui"); SYM(sub_ui, "__gmpz_sub_ui"); SYM(divexact, "__gmpz_divexact");
#undef SYM
    G.h = h;
    return 1;
}
static void z_in(mpz_s* z, const u64* p, int L) { G.import(z, (size_t)L, -1, 8, 0, 0, p); }
static void z_out(u64* p, int L, const mpz_s* z) {
    size_t cnt = 0;
    memset(p, 0, sizeof(u64) * L);
    G.export_(p, &cnt, -1, 8, 0, 0, z);
}

/* ct = (1 + m n) * hs^r mod n^2 */
int orc_gmp_encrypt_djn_batch(int N, int Ln, const u64* n, const u64* nsq, const u64* hs, const u64* m, const u64* r,
                              int Lr, u64* ct, int threads) {
    if (!orc_gmp_available()) return -2;
#pragma omp parallel num_threads(nthreads_or(threads))
    {
        mpz_tt zn, znsq, zhs, zm, zr, zo, zc;
        G.init(zn); G.init(znsq); G.init(zhs); G.init(zm); G.init(zr); G.init(zo); G.init(zc);
        z_in(zn, n, Ln); z_in(znsq, nsq, 2 * Ln); z_in(zhs, hs, 2 * Ln);
#pragma omp for schedule(dynamic, 4)
        for (int i = 0; i < N; ++i) {
            z_in(zm, m + (size_t)i * Ln, Ln);
            z_in(zr, r + (size_t)i * Lr, Lr);
            G.powm(zo, zhs, zr, znsq);
            G.mul(zc, zm, zn);
            G.add_ui(zc, zc, 1);
            G.mul(zc, zc, zo);
            G.mod(zc, zc, znsq);
            z_out(ct + (size_t)i * 2 * Ln, 2 * Ln, zc);
        }
        G.clear(zn); G.clear(znsq); G.clear(zhs); G.clear(zm); G.clear(zr); G.clear(zo); G.clear(zc);
    }
    return 0;
}

/* CRT decryption with hp, hq, pinvq supplied */
int orc_gmp_decrypt_crt_batch(int N, int Ln, int Lh, const u64* p, const u64* q, const u64* hp, const u64* hq,
                              const u64* pinvq, const u64* ct, u64* m_out, int threads) {
    if (!orc_gmp_available()) return -2;
#pragma omp parallel num_threads(nthreads_or(threads))
    {
        mpz_tt zp, zq, zp2, zq2, zpm1, zqm1, zhp, zhq, zpi, zc, zx, zmp, zmq;
        mpz_s* all[] = {zp, zq, zp2, zq2, zpm1, zqm1, zhp, zhq, zpi, zc, zx, zmp, zmq};
        for (unsigned k = 0; k < sizeof(all) / sizeof(all[0]); ++k) G.init(all[k]);
        z_in(zp, p, Lh); z_in(zq, q, Lh); z_in(zhp, hp, Lh); z_in(zhq, hq, Lh); z_in(zpi, pinvq, Lh);
        G.mul(zp2, zp, zp); G.mul(zq2, zq, zq); G.sub_ui(zpm1, zp, 1); G.sub_ui(zqm1, zq, 1);
#pragma omp for schedule(dynamic, 4)
        for (int i = 0; i < N; ++i) {
            z_in(zc, ct + (size_t)i * 2 * Ln, 2 * Ln);
            G.mod(zx, zc, zp2); G.powm(zx, zx, zpm1, zp2); G.sub_ui(zx, zx, 1); G.divexact(zx, zx, zp);
            G.mul(zx, zx, zhp); G.mod(zmp, zx, zp);
            G.mod(zx, zc, zq2); G.powm(zx, zx, zqm1, zq2); G.sub_ui(zx, zx, 1); G.divexact(zx, zx, zq);
            G.mul(zx, zx, zhq); G.mod(zmq, zx, zq);
            G.sub(zx, zmq, zmp); G.mul(zx, zx, zpi); G.mod(zx, zx, zq);
            G.mul(zx, zx, zp); G.add(zx, zx, zmp);
            z_out(m_out + (size_t)i * Ln, Ln, zx);
        }
        for (unsigned k = 0; k < sizeof(all) / sizeof(all[0]); ++k) G.clear(all[k]);
    }
    return 0;
}

/* AVX512-IFMA multi-buffer (mb8) exponentiation: the CPU algorithm README.md:32 names, restated in our own code */
#ifdef __AVX512IFMA__
#include "paillier_ifma.c"
#else
int orc_ifma_available(void) { return 0; }
#endif
