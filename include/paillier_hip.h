/* paillier_hip.h — C ABI of libpaillier_hip.so, the MI355X (gfx950) Paillier engine.
 *
 * This library is the drop-in replacement for the native layer under the reference's Python API:
 * the pybind11 module `ipcl_bindings` (src/ipcl_python/bindings/ipcl_bindings.cpp:21-63) plus the
 * un-vendored ipcl / IPP-Crypto libraries it forwards to.  Each entry point cites the reference
 * interface it replaces.  Conventions:
 *
 *  - every function returns 0 on success, a negative PAI_E_* code on failure; pai_last_error()
 *    returns a thread-local message for the last failure (C++ exceptions never cross the ABI);
 *  - big integers are little-endian arrays of native-endian uint32_t words (the wire order of
 *    pyByte2BN / BN2bytes, ipcl_bindings.cpp:100-138), batches are row-major [N][words];
 *  - pointers named d_* are DEVICE pointers on the key's device; h_* are HOST pointers;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous
 *    with respect to the host unless stated otherwise;
 *  - a key handle owns device scratch (window tables, quotient-digit columns) that its operations reuse.  The
 *    library orders the users of that scratch itself: operations of one handle issued on the SAME stream are
 *    ordered by the stream, operations issued on DIFFERENT streams are chained with an event (the later call makes
 *    its stream wait for the earlier one), so any mix of threads and streams on one handle is safe, and handles
 *    of the same key material on different devices are independent.  Only pai_ct_invert (it reports
 *    non-invertible inputs; pai_ct_invert_async + pai_pubkey_status is the asynchronous form), pai_pubkey_status,
 *    pai_pubkey_trim and pai_ct_pow2 without a hint return after their stream work has completed; everything else,
 *    pai_decrypt and pai_modexp_fixed (its host exponent is copied to pinned staging) included, is asynchronous;
 *  - a call never changes the calling thread's current HIP device (it is restored on return);
 *  - there is NO CPU fallback: without a usable gfx950 device key creation fails with
 *    PAI_E_NODEVICE.
 *
 * Word counts for a key of `key_bits` bits: n_words = ceil(key_bits/32) (plaintext residues mod n),
 * ct_words = 2*n_words (ciphertexts mod n^2), r_words = ceil(randbits/32) (DJN randomness).
 */
#ifndef PAILLIER_HIP_H_
#define PAILLIER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAI_OK 0
#define PAI_E_INVALID (-1)    /* bad argument */
#define PAI_E_NODEVICE (-2)   /* no gfx950 device / HIP runtime failure at set-up */
#define PAI_E_HIP (-3)        /* HIP runtime error during a call */
#define PAI_E_UNSUPPORTED (-4)/* key size outside the compiled geometries (moduli up to 8192 bits) */
#define PAI_E_INTERNAL (-5)

typedef struct pai_pubkey pai_pubkey;      /* replaces ipclPublicKey  (ipcl_bindings_classes.cpp:12-91)  */
typedef struct pai_privkey pai_privkey;    /* replaces ipclPrivateKey (ipcl_bindings_classes.cpp:93-163) */
typedef struct pai_modulus pai_modulus;    /* a bare odd modulus for pai_modmul / pai_modexp_*           */

/* ---- library ---------------------------------------------------------------------------------- */
int pai_version(void);                                   /* 100*major + minor */
int pai_device_count(int* count);                        /* number of visible HIP devices */
const char* pai_last_error(void);

/* Per-kernel timing (HIP events on the caller's stream).  While enabled, pai_encrypt / pai_decrypt
 * become synchronous and record the duration of each kernel they launch; pai_profile_last(i, ...)
 * returns entry i of the calling thread's last call (PAI_E_INVALID past the end). */
int pai_profile_enable(int on);
int pai_profile_last(int index, char* name_out, size_t name_cap, float* ms_out);

/* ---- device memory helpers (for callers that do not bring their own allocator) ---------------- */
int pai_malloc(int device, size_t bytes, void** d_ptr);
int pai_free(int device, void* d_ptr);
int pai_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes, void* stream);
int pai_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes, void* stream);
int pai_stream_sync(int device, void* stream);

/* Small host operands of asynchronous calls without a copy on the stream (round 6).  The reference's pybind layer takes host
 * vectors by value (bindings/ipcl_bindings_classes.cpp:318-325: the exponents of CipherText * PlainText, the shifts the
 * Python layer computes for ipcl_python.py:570-741); at its own benchmark sizes (16 / 64 elements, bench/bench_ipcl_python.py:
 * 22-78) an H2D copy per operand costs as much as the kernel.  pai_host_stage copies `parts` host arrays (together at most
 * PAI_HOST_STAGE_MAX bytes, each part 16-byte aligned) into one slot of a pinned, device-mapped ring of the calling thread
 * and returns, per part, a pointer a kernel may READ through directly.  Contract: the pointers may be passed as read-only
 * device operands (exponents, shifts, codec inputs) of calls enqueued on `stream` BEFORE the calling thread's next
 * pai_host_stage for that device; the library keeps the slot intact until those calls have executed. */
#define PAI_HOST_STAGE_MAX 4096
int pai_host_stage(int device, int parts, const void* const* h_src, const size_t* bytes, void* stream, void** d_ptrs);
/* pai_ct_add_aligned / pai_ct_mul (below) with their small operand — the shifts, the exponents — still on the HOST (at most
 * PAI_HOST_STAGE_MAX bytes): staged as by pai_host_stage and read by the kernel in place, in one call. */
int pai_ct_add_aligned_host(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* h_delta,
                            size_t N, uint32_t* d_out, void* stream);
int pai_ct_mul_host(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* h_e, int e_words, int ebits_max, int e_bcast, size_t N,
                    uint32_t* d_out, void* stream);

/* ---- container operations on device rows (no arithmetic) -----------------------------------------
 * ipclPlainText / ipclCipherText __getitem__ with a slice and rotate (bindings/ipcl_bindings_classes.cpp:224-262,328-366;
 * rotate is what the reference's reductions are built from, ipcl_python.py:810-827).  Rows of `row_words` words.
 * slice:  d_out[i] = d_src[start + i * step] for i < count (step >= 1);  rotate:  d_out[i] = d_src[(i + shift) mod N]
 * (any sign of shift; d_out must not alias d_src).  Asynchronous on `stream`. */
int pai_buf_slice(int device, const uint32_t* d_src, int row_words, size_t start, size_t count, size_t step, uint32_t* d_out,
                  void* stream);
int pai_buf_rotate(int device, const uint32_t* d_src, int row_words, size_t N, long long shift, uint32_t* d_out, void* stream);

/* ---- keys ------------------------------------------------------------------------------------- */
/* ipclPublicKey(n, bits, enableDJN) — classes.cpp:24-27 — and the pickle form
 * (scheme, n, bits, hs, randbits) — ipcl_bindings.cpp:66-98.  h_hs == NULL selects the standard
 * scheme (obfuscator r^n); otherwise the DJN scheme (obfuscator hs^r, r of `randbits` bits) and the
 * fixed-base table for hs is built on the device by the first call that obfuscates (pai_encrypt with randomness /
 * pai_obfuscate): a handle that only adds, multiplies or decrypts never allocates it. */
int pai_pubkey_create(const uint32_t* h_n, int n_words, int key_bits, const uint32_t* h_hs, int hs_words,
                      int randbits, int device, pai_pubkey** out);
void pai_pubkey_destroy(pai_pubkey* pk);
/* Extension for many-key (federated) deployments: frees what the handle holds beyond its constants — the DJN fixed-base
 * tables (8.6 GB at 2048-bit keys; the next obfuscating call rebuilds them, sized from the memory that is free THEN) and
 * the grow-only scratch of the batch operations.  Synchronises the device.  *freed_bytes (optional): device memory returned. */
int pai_pubkey_trim(pai_pubkey* pk, size_t* freed_bytes);
/* fills any non-NULL out-parameter */
int pai_pubkey_info(const pai_pubkey* pk, int* key_bits, int* n_words, int* ct_words, int* r_words,
                    int* randbits, int* is_djn, int* device);
/* The DJN fixed-base table this handle holds right now (zeros before its first obfuscating call and after a trim or an
 * eviction): device bytes, window width in bits, number of windows.  The first keys of a device get the big table (1/32 of
 * the device memory), later ones the small operating point (PAI_FB_BIG_KEYS / PAI_FB_SMALL_TABLE_MB). */
int pai_pubkey_table_info(const pai_pubkey* pk, size_t* table_bytes, int* window_bits, int* windows);
/* The batch sizes at which this key's calls change kernel family on its device (csrc/path_ranges.hpp: every range is a number
 * of elements per compute unit times the device's CU count; PAI_LATENCY_MAX / PAI_LAT_ADD_MAX / PAI_TUNE overrides included):
 * op 0 = pai_decrypt, 1 = pai_encrypt (DJN), 2 = pai_ct_mul, 3 = pai_ct_add*.  An edge E separates N = E from N = E + 1.
 * Writes at most `cap` edges (ascending) and the full count.  For tests and probes that want to stand on both sides of a switch
 * (tests/test_gpu_path_edges.py); no reference counterpart. */
int pai_path_edges(const pai_pubkey* pk, int op, size_t* edges, int cap, int* count);

/* ipclKeypair.generate_keypair(n_length, enable_DJN) — bindings/ipcl_bindings.cpp:12-15 -> ipcl::generateKeypair
 * (timed by the reference's BM_KeyGen, bench/bench_ipcl_python.py:13-19).  Host-only (no device work): two random primes
 * of key_bits/2 bits with the two top bits set (so p*q has exactly key_bits bits), p != q; djn != 0 adds upstream's DJN
 * constraints p = q = 3 (mod 4), gcd(p-1, q-1) = 2.  Incremental search over a sieve of the primes below 2^16, Miller-
 * Rabin with base 2 and 24 random bases, the two primes searched on two host threads.  key_bits: a multiple of 64,
 * 128..8192.  h_seed == NULL: the kernel CSPRNG (getrandom); a seed makes the key reproducible (tests only — such a
 * key is NOT secret).  h_p, h_q receive key_bits/64 words each, p < q not guaranteed. */
int pai_keygen(int key_bits, int djn, const uint64_t* h_seed, uint32_t* h_p, uint32_t* h_q);
/* Host big-integer modular exponentiation for key set-up (the DJN base hs = (-x^2)^n mod n^2 of ipcl::PublicKey's
 * constructor, classes.cpp:24-27): h_out = h_base ^ h_exp mod h_mod for an odd modulus of mod_words words (<= 260);
 * h_base: mod_words words, < modulus; h_out: mod_words words.  Host-only, synchronous. */
int pai_host_modexp(const uint32_t* h_base, const uint32_t* h_exp, int exp_words, const uint32_t* h_mod, int mod_words,
                    uint32_t* h_out);

/* ipclPrivateKey(pubkey, p, q) — classes.cpp:96-101.  p and q may come in either order; n == p*q is
 * checked.  Derives p^2, q^2, hp, hq, p^-1 mod q. */
int pai_privkey_create(const pai_pubkey* pk, const uint32_t* h_p, int p_words, const uint32_t* h_q, int q_words,
                       pai_privkey** out);
void pai_privkey_destroy(pai_privkey* sk);

/* ---- hot path --------------------------------------------------------------------------------- */
/* ipclPublicKey.encrypt(pt, make_secure=false) — classes.cpp:53-60; L3 raw_encrypt, ipcl_python.py:103-106.
 * d_ct[i] = (1 + d_m[i] * n) mod n^2.   d_m: [N][n_words], each < n.   d_ct: [N][ct_words]. */
int pai_raw_encrypt(const pai_pubkey* pk, const uint32_t* d_m, size_t N, uint32_t* d_ct, void* stream);
/* PaillierEncryptedNumber + plaintext — ipcl_python.py:495-504 raw-encrypts the plaintext (classes.cpp:53-60 with make_secure=false)
 * and :365-381 / classes.cpp:318-321 multiply the two ciphertexts; in ONE pass here:
 * d_out[i] = d_ct[i] * (1 + d_m[i] * n) mod n^2, wire form in and out (round 6: the reference's BM_Add_CTPT is two launches and an
 * intermediate array otherwise).  d_m: [N][n_words] residues < n (already encoded AT the ciphertext's exponents: pai_fp_encode_at).
 * d_out may alias d_ct.  Small batches run on the latency geometry (three sequential products), others on lane groups. */
int pai_ct_add_plain(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_m, size_t N, uint32_t* d_out, void* stream);

/* ipclPublicKey.encrypt(pt, make_secure=true) — classes.cpp:53-60 with the randomness made explicit.
 * DJN keys:      d_r: [N][r_words], r_i < 2^randbits;   ct_i = (1 + m_i n) * hs^{r_i} mod n^2.
 * standard keys: d_r: [N][n_words], 0 < r_i < n;         ct_i = (1 + m_i n) * r_i^n   mod n^2. */
int pai_encrypt(const pai_pubkey* pk, const uint32_t* d_m, const uint32_t* d_r, size_t N, uint32_t* d_ct,
                void* stream);

/* ipclPublicKey.apply_obfuscator(CipherText) — classes.cpp:77-83: d_ct[i] <- d_ct[i] * obf(r_i) mod n^2. */
int pai_obfuscate(const pai_pubkey* pk, uint32_t* d_ct, const uint32_t* d_r, size_t N, void* stream);

/* ipclPrivateKey.decrypt(CipherText) — classes.cpp:127-133 (CRT).  d_m: [N][n_words]. */
int pai_decrypt(pai_privkey* sk, const uint32_t* d_ct, size_t N, uint32_t* d_m, void* stream);

/* ipclCipherText.__add__(ct, ct) — classes.cpp:318-321: d_out[i] = d_a[i] * d_b[i] mod n^2.
 * b_bcast != 0: d_b holds one ciphertext used for every i (size-1 broadcast).  d_out may alias d_a.
 * Operands are residues modulo n^2 (as everywhere in this ABI); the result is the canonical residue.  (Batches beyond the
 * small-batch range run one most-significant-limb-first product per element, csrc/mont_msb.hpp, which would also accept rows that
 * are not reduced; the small-batch route does not promise that.) */
int pai_ct_add(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N,
               uint32_t* d_out, void* stream);

/* ipclCipherText.__mul__(ct, pt) — classes.cpp:324-325: d_out[i] = d_ct[i] ^ e_i mod n^2, e_i >= 0.
 * d_e: [N][e_words] (or one row if e_bcast); ebits_max bounds the bit length of every e_i. */
int pai_ct_mul(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_e, int e_words, int ebits_max,
               int e_bcast, size_t N, uint32_t* d_out, void* stream);

/* Replaces the per-element gmpy2.invert of PaillierEncryptedNumber.__invert_ct (ipcl_python.py:272-276):
 * d_out[i] = d_ct[i]^-1 mod n^2 (batched: simultaneous inversion as a product tree + one extended GCD per top-level
 * product).  Synchronous; fails with PAI_E_INVALID if some ciphertext shares a factor with n.  d_out may alias d_ct. */
int pai_ct_invert(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream);
/* The same work without the synchronisation: returns as soon as the kernels are queued on `stream`; a non-invertible input
 * sets bit 0 of the handle's sticky status word (pai_pubkey_status below) instead of failing the call. */
int pai_ct_invert_async(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream);
/* The asynchronous form whose outcome travels with the RESULT instead of the handle: *d_flag (one device word the caller owns,
 * zeroed by the caller) receives bit 0 when an input of THIS call is not invertible (the output rows are then undefined).  The
 * Python layer keeps the word with the container built from d_out (and with everything computed from it) and reads it — one
 * synchronisation — when that container is exported or decrypted, so the error is raised on the faulty object. */
int pai_ct_invert_flag(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, int32_t* d_flag, void* stream);

/* __raw_add with its exponent alignment fused (ipcl_python.py:490-526 + :570-741): delta_i = exponent(a_i) - exponent(b_i)
 * (base-2 fixed-point exponents, int32 on the device); the operand with the LOWER exponent is raised first:
 * d_out[i] = delta_i > 0 ? d_a[i] * d_b[i]^(2^delta_i) : d_a[i]^(2^-delta_i) * d_b[i]   (mod n^2).
 * The result's exponent is max(exponent(a_i), exponent(b_i)) (the caller's bookkeeping).  One pass over the data instead
 * of two pai_ct_pow2 passes and a pai_ct_add.  b_bcast != 0: one ciphertext b for every i.  d_out may alias d_a or d_b. */
int pai_ct_add_aligned(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                       size_t N, uint32_t* d_out, void* stream);

/* n-ary ciphertext sum in one pass — the aggregation sum_j E(x_j) over k parties' arrays, which the reference spells as a chain
 * of PaillierEncryptedNumber.__add__ calls with their exponent alignments (ipcl_python.py:365-381, 490-526, 570-741;
 * tests/ipcl_python_test.py:21-38):  d_out[i] = prod_{j<k} op_j[i]^(2^raise_j[i]) mod n^2, 2 <= k <= 16.
 * h_ops: HOST array of k device pointers to [N][ct_words] rows; h_raise: NULL, or a HOST array of k device pointers (entries may
 * be NULL) to int32 [N] squaring counts >= 0 — the caller raises every operand to the per-element maximum exponent of the sum.
 * Lazy-domain bookkeeping as for pai_ct_mont_mul: operand 0 holds x R^tag0, the others x R^tag, the result x R^dom_out (any tags
 * with |.| small: wire form is 0; tag0 + (k-1)(tag-1) is the tag that costs no extra product; a raised operand 0 needs tag0 ==
 * tag).  k - 1 Montgomery products per element (+1 per raised operand and per squaring), one store.  d_out may alias an operand.
 * Asynchronous on `stream`. */
int pai_ct_addn(const pai_pubkey* pk, const uint32_t* const* h_ops, const int32_t* const* h_raise, int k, int tag0, int tag,
                int dom_out, size_t N, uint32_t* d_out, void* stream);

/* Lazy Montgomery domain for chains of additions (extension).  A ciphertext buffer may hold x R^k mod n^2
 * instead of x (R = 2^bits of pai_pubkey_mont_bits; the caller tracks the integer k per buffer — k = 0 is the wire form).
 * pai_ct_mont_mul: d_out[i] = d_a[i] * d_b[i] * R^-1 mod n^2 (canonical residue), ONE Montgomery product per element where
 * pai_ct_add needs two (or the 1.4 x dearer most-significant-limb-first product): operands with tags ka, kb give ciphertext-addition with tag ka + kb - 1.  Multiplying by the
 * broadcast constant R^(1 + k' - k) mod n^2 (b_bcast != 0) moves a buffer from tag k to tag k', e.g. back to the wire
 * form before decryption, export or pickling — the bits at every boundary are those of CipherText::operator+
 * (classes.cpp:318-321).  d_out may alias d_a. */
int pai_ct_mont_mul(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N, uint32_t* d_out,
                    void* stream);
int pai_pubkey_mont_bits(const pai_pubkey* pk, int* bits);
/* pai_ct_add_aligned on buffers that share a tag k: d_entry = one packed row holding R^(2 - k) mod n^2 (for k = 0 this
 * is what pai_ct_add_aligned uses); the result carries tag k. */
int pai_ct_add_aligned_dom(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                           size_t N, uint32_t* d_out, const uint32_t* d_entry, void* stream);

/* The reductions of PaillierEncryptedNumber.sum / __matmul__ (ipcl_python.py:746-762, 810-880): upstream pads to a
 * power of two with E_raw(0) = 1 and runs log2 steps of CipherText::rotate + operator+ (__padded_ct, :810-827),
 * i.e. computes the product of a group of ciphertexts modulo n^2.  Here: d_ct holds `count` rows read as
 * [count/groups][groups] (member-major), and d_out[g] = prod_l d_ct[l*groups + g] mod n^2 for g < groups — a
 * product tree over halves, one Montgomery product per node.  The product is order-independent, so the bits equal
 * upstream's rotate-and-add result.  count must be a positive multiple of groups; groups == 1 is sum(). */
int pai_ct_prod(const pai_pubkey* pk, const uint32_t* d_ct, size_t count, size_t groups, uint32_t* d_out, void* stream);

/* The matrix products PaillierEncryptedNumber.__matmul__ / __rmatmul__ / dot (ipcl_python.py:829-930): every output
 * element is a sum of ciphertext * plaintext terms aligned to a common exponent, i.e. the product of powers
 *     d_out[r * M + j] = prod_{l < K}  base(r, l, j)^(e[r][l][j])  mod n^2,
 * base(r, l, j) = d_ct[r * K + l], or d_ct_inv[r * K + l] (the caller's pai_ct_invert of d_ct) where d_sign[l * M + j] != 0
 * (negative multipliers, ipcl_python.py:426-437); e = |mantissa| << alignment shift, e_words words each, little endian,
 * laid out [R][K][M][e_words]; ebits_max bounds every exponent.  d_sign and d_ct_inv are both NULL or both given.
 * One chain of squarings per output element and chunk of members instead of one per term (Straus; windows of 2..7 bits
 * over per-base power tables built on the fly, the width chosen from the shape; on base-n digit pairs for keys up to 2048 bits,
 * on lane groups above).  PAI_E_UNSUPPORTED when the power tables (2^w entries per base and sign) do not fit — the caller then takes
 * the term-by-term route (pai_ct_mul + pai_ct_add_aligned / pai_ct_prod).  The bits equal that route's: the result is the
 * canonical residue of the same product. */
int pai_ct_multiexp(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_ct_inv, size_t R, size_t K, size_t M,
                    const uint32_t* d_e, int e_words, int ebits_max, const uint8_t* d_sign, uint32_t* d_out, void* stream);

/* Exponent alignment, ipcl_python.py:570-741 (ct * 2^delta as ciphertext^(2^delta)):
 * for delta_i > 0: d_ct[i] <- d_ct[i]^(2^delta_i) mod n^2; other elements are left untouched.
 * Batches of >= 16384 elements (PAI_POW2_DIGIT_MIN) on keys up to 2048 bits read the largest shift back first (this
 * synchronises `stream`) and run shifts of 8..62 as ct^e with the one-bit exponent 2^delta on the base-n digit engine. */
int pai_ct_pow2(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N,
                void* stream);
/* The same with the caller's knowledge of max_i delta_i (the Python layer builds delta on the host and knows it): no
 * read-back, the call is asynchronous for every batch size.  max_delta must be >= every delta_i (0: nothing to do). */
int pai_ct_pow2_hint(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N, int max_delta,
                     void* stream);
/* Sticky status word of a handle's ASYNCHRONOUS calls, read with a synchronisation of `stream` (and cleared when
 * clear != 0): bit 0 — pai_ct_invert_async met a ciphertext that is not invertible modulo n^2 (its output rows are then
 * undefined); bit 1 — a pai_ct_pow2_hint call was given a max_delta below a shift of its batch on the digit-engine path
 * (batches >= PAI_POW2_DIGIT_MIN on keys up to 2048 bits, hints the digit path serves; the raised ciphertexts of that call are
 * then wrong — a hint outside that range runs the lane-group kernel, which is correct for any shift, and flags nothing).  The
 * Python layer computes its hints from host arrays and uses pai_ct_invert_flag, so it never depends on this word. */
int pai_pubkey_status(const pai_pubkey* pk, int* status_out, int clear, void* stream);


/* ---- data formats either side of the path --------------------------------------------------------------
 * Fixed-point codec of bindings/fixedpoint.py:54-115 for float64 arrays (the hot Python loops of
 * ipcl_python.py:136-140 and :229-243), on the device so that 8 B instead of 256 B per element cross PCIe.
 * Both need n > 2^66 (any real key).  encode: d_x[N] finite doubles (the caller rejects NaN/Inf like the
 * reference's int()) -> residues d_m[N][n_words] and base-2 exponents d_expo[N].  decode: residues ->
 * signed mantissas d_mant[N] with d_flag[i] = 0, or d_flag[i] = 1 when element i needs the exact big-integer
 * path (|mantissa| >= 2^63, overflow zone or corrupt residue: the host path raises the reference's errors). */
int pai_fp_encode_f64(const pai_pubkey* pk, const double* d_x, size_t N, uint32_t* d_m, int32_t* d_expo, void* stream);
/* the same for int64 arrays: exponent 0, residue = x mod n (fixedpoint.py:72-74,89-96) */
int pai_fp_encode_i64(const pai_pubkey* pk, const int64_t* d_x, size_t N, uint32_t* d_m, int32_t* d_expo, void* stream);
/* Encoders with a target exponent per element (d_target[N], or one value when target_bcast), for the plaintext operand
 * of ct + plaintext (ipcl_python.py:495-504 raw-encrypts it, :570-741 then raise the lower-exponent side by
 * ct^(2^delta)): for a RAW encryption (1 + m n)^(2^d) = 1 + (m 2^d mod n) n, so an element whose own exponent e is below
 * its target t is encoded at t directly (mantissa shifted left by t - e; same ciphertext bits, same exponent, no
 * squarings) whenever |mantissa| 2^(t-e) < 2^(bits(n) - 2); other elements keep e (d_expo tells).  is_f64: d_x is
 * double[N], else int64[N]. */
int pai_fp_encode_at(const pai_pubkey* pk, const void* d_x, int is_f64, size_t N, const int32_t* d_target, int target_bcast,
                     uint32_t* d_m, int32_t* d_expo, void* stream);
int pai_fp_decode_i64(const pai_pubkey* pk, const uint32_t* d_m, size_t N, int64_t* d_mant, int32_t* d_flag, void* stream);

/* Obfuscator randomness, replacing upstream ipcl's per-element getRandomBN inside
 * PublicKey::encrypt (called at classes.cpp:57): d_r[N][r_words] <- ChaCha20 key stream (RFC 8439 block
 * function; h_key8 = 256-bit key from the OS CSPRNG, h_nonce3 = 96-bit nonce, 32-bit block counter starting
 * at counter0, carried into nonce word 0), top word of every row masked to randbits.  Standard-scheme keys get rows of
 * bits(n) random bits: candidates for r, of which the caller keeps those in [1, n) (rejection sampling). */
int pai_draw_r(const pai_pubkey* pk, const uint32_t* h_key8, const uint32_t* h_nonce3, uint32_t counter0, size_t N,
               uint32_t* d_r, void* stream);

/* ---- multi-GPU (one node) --------------------------------------------------------------------------------
 * The hot operations are element-wise (encrypt / decrypt / add / mul forward whole std::vector<BigNumber> batches:
 * classes.cpp:53-60, 127-133, 318-325), so a batch shards by contiguous blocks with no exchange inside an operation:
 * shard g of G owns rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G))) (trailing shards may be empty). */
int pai_shard_plan(size_t N, int nshards, int shard, size_t* begin, size_t* count);
/* Single-process fan-out over the visible devices (one key handle per device): gathers row shards living on
 * devices[i] (rows[i] rows of row_words words each, back to back in shard order) into d_out on dst_device, or
 * scatters d_in on src_device into the shards — peer copies over xGMI, all shards in flight at once; returns when
 * the copies have completed.  (One-process-per-GPU jobs gather with RCCL instead: sharding.py.) */
int pai_gather(int nshards, const int* devices, const void* const* d_shards, const size_t* rows, int row_words,
               int dst_device, void* d_out);
int pai_scatter(int nshards, const int* devices, void* const* d_shards, const size_t* rows, int row_words,
                int src_device, const void* d_in);

/* ---- generic modular building blocks (arbitrary odd modulus up to 8192 bits) ------------------- */
int pai_modulus_create(const uint32_t* h_m, int m_words, int device, pai_modulus** out);
void pai_modulus_destroy(pai_modulus* m);
/* out[i] = a[i] * b[i] mod M; rows of w32 words (w32 = ceil(bits(M)/32)) */
int pai_modmul(pai_modulus* m, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N, uint32_t* d_out,
               void* stream);
/* out[i] = base[i] ^ E mod M, E given on the HOST (wave-uniform exponent, fixed 5-bit window) */
int pai_modexp_fixed(pai_modulus* m, const uint32_t* d_base, const uint32_t* h_e, int e_words, size_t N,
                     uint32_t* d_out, void* stream);
/* out[i] = base[i] ^ e[i] mod M, per-element exponents on the DEVICE */
int pai_modexp_var(pai_modulus* m, const uint32_t* d_base, int base_bcast, const uint32_t* d_e, int e_words,
                   int ebits_max, int e_bcast, size_t N, uint32_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAILLIER_HIP_H_ */
