// Table of kernel launchers for one lane-group geometry (NLL limbs per lane x T lanes per integer).
// One translation unit per geometry instantiates it (geo_*.hip) so the build parallelises.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "kernels_paillier.hpp"
#include "mont_msb.hpp"

namespace pai {

constexpr int MODEXP_WINDOW = 5;     // fixed-window width of the uniform-exponent kernels

struct GeoOps {
    int nll, t, u, nl, epb;
    int lds_bytes;        // one operand buffer + modulus copy
    int lds_bytes_b;      // stage-B decrypt kernel (two operand buffers)
    // mode: MODMUL_FULL (a*b mod M) or MODMUL_MONT (a*b*R^-1 mod M), kernels_modexp.hpp
    void (*modmul)(hipStream_t, int grid, const MontCtx*, const uint32_t* a, const uint32_t* b, uint32_t* out,
                   int n, int w32, int b_bcast, int mode, const MontCtx* fin);
    void (*modexp_fixed)(hipStream_t, int grid, const MontCtx*, const uint32_t* base, int base_w32,
                         const uint32_t* expo, int ewords, int ebits, uint32_t* out, int out_w32, int n,
                         uint32_t* table, int keep_mont);
    void (*modexp_var)(hipStream_t, int grid, const MontCtx*, const uint32_t* base, int base_w32, int base_shift,
                       const uint32_t* expo, int ew, int ebits_max, int exp_bcast, uint32_t* out, int out_w32,
                       int n, int keep_mont, int out_raw);
    // windowed variant: table scratch of var_table_words(blocks, wbits) words
    void (*modexp_var_win)(hipStream_t, int grid, const MontCtx*, const uint32_t* base, int base_w32, const uint32_t* expo,
                           int ew, int ebits_max, int exp_bcast, uint32_t* out, int out_w32, int n, uint32_t* table, int wbits,
                           const MontCtx* fin /* minus-one context pairs on the wide-group geometries, else NULL */);
    void (*encrypt)(hipStream_t, int grid, EncParams, const uint32_t* m, const uint32_t* r, const uint32_t* ct_in,
                    uint32_t* ct_out, int n, int mode);
    // T[j][hi * 2^h + lo] = S[2 j + 1][hi] * S[2 j][lo]: second level of the fixed-base table build (raw Montgomery rows)
    void (*fb_expand)(hipStream_t, int grid, const MontCtx*, const uint32_t* S, uint32_t* T, int J, int h);
    void (*dec_a)(hipStream_t, int gridx, DecAParams, const uint32_t* ct, uint32_t* u_out, int n, uint32_t* table);
    void (*dec_b)(hipStream_t, int grid, DecBParams, const uint32_t* u_in, uint32_t* m_out, int n);
    void (*pow2)(hipStream_t, int grid, const MontCtx*, uint32_t* ct, const int32_t* delta, int delta_bcast, int n,
                 int w32, const MontCtx* fin);
    // out_i = a_i * b_i with the lower-exponent side raised by ^(2^|delta_i|) first (delta = exponent(a) - exponent(b))
    // out[j] = base^(2^(h j)) mod M, j < nsnap, plain packed rows: one chain of squarings (fin: see modexp_var_win)
    void (*sq_chain)(hipStream_t, const MontCtx*, const MontCtx* fin, const uint32_t* base, int w32, uint32_t* out, int h, int nsnap);
    void (*add_aligned)(hipStream_t, int grid, const MontCtx*, const uint32_t* a, const uint32_t* b, int b_bcast,
                        const int32_t* delta, uint32_t* out, int n, int w32, const uint32_t* entry, const MontCtx* fin);
    // n-ary sum (k_addn): out_i = prod_j op_j[i]^(2^raise_j[i]); rpow = the key's table of R^m, |m| <= RPOW_SPAN, NL limbs per row
    void (*addn)(hipStream_t, int grid, const MontCtx*, AddnArgs, uint32_t* out, int n, int w32, const uint32_t* rpow);
    // words of table scratch needed by modexp_fixed / dec_a for `blocks` resident workgroups
    size_t (*table_words)(size_t blocks);
    // ct = w + v n [or ct_in (w + v n)] from the plain digit pairs of the lane-group pair kernels (rows [2][wv_words])
    void (*pair_finish)(hipStream_t, int grid, EncParams, const uint32_t* wv, int wv_words, const uint32_t* ct_in,
                        uint32_t* ct_out, int n, int mul_ct);
    // multi-exponentiation (k_mexp_table / k_mexp): power tables per (base, sign), then chunks of members per output
    void (*mexp_table)(hipStream_t, int grid, const MontCtx*, const uint32_t* ct, const uint32_t* ct_inv, int w32, uint32_t* table,
                       int nentries, int nsigns, int wbits);
    void (*mexp)(hipStream_t, int grid, const MontCtx*, MexpParams, const uint32_t* table, const uint32_t* e, const uint8_t* sign,
                 uint32_t* out, int nlanes);
    // out = a * b mod M on wire-form rows by one most-significant-limb-first product (k_modmul_msb; NULL on the latency geometries)
    void (*modmul_msb)(hipStream_t, int grid, const MsbCtx*, const uint32_t* a, const uint32_t* b, uint32_t* out, int n, int w32);
};

const GeoOps* geo_ops_36x1();
const GeoOps* geo_ops_36x2();
const GeoOps* geo_ops_28x4();
const GeoOps* geo_ops_36x4();
const GeoOps* geo_ops_28x8();
const GeoOps* geo_ops_36x8();
// latency geometries (an integer spread over 16 / 32 / 64 lanes): small batches
const GeoOps* geo_ops_3x16();
const GeoOps* geo_ops_3x32();
const GeoOps* geo_ops_3x64();
const GeoOps* geo_ops_9x32();

// stage A of the smallest decryptions on digit pairs, pipelined over four waves (kernels_declat.hpp; 3 x 64 geometry only)
struct DecPPParams;
void launch_dec_a_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* u_out, int chain_limbs);
void launch_ctmul_pp(hipStream_t s, int n, const DecPPParams& P, const uint32_t* ct, uint32_t* out, int chain_limbs);

// smallest geometry whose capacity covers a modulus of `bits` bits (R = 2^(29 NL) > 4 M), or nullptr
const GeoOps* geo_for_bits(int bits);
// the same for the latency geometries
const GeoOps* geo_latency_for_bits(int bits);

// Wide engine (one integer per lane, mont_wide.hpp): limb count serving a modulus of `bits` bits (0 = none),
// table scratch words, and the stage-A decrypt launcher (grid = gridx x 2 primes, 256 elements per block).
int wide_nl_for_bits(int bits);
size_t wide_table_words(int nl, size_t blocks);
bool launch_dec_a_wide(int nl, hipStream_t s, int gridx, const DecAParams& P, const uint32_t* ct, uint32_t* u_out, int n,
                       uint32_t* table);

// p-adic digit engine for CRT-decrypt stage A (mont_padic.hpp / kernels_padic.hpp)
struct DecPadicParams;
constexpr int PADIC_SLIDE_BITS = 6;    // sliding-window width of the decrypt exponent schedule (k_dec_a per 2^20: 486 / 478 / 489 ms at 5 / 6 / 7 bits)
constexpr int PADIC_TBL_ENTRIES = 1 << (PADIC_SLIDE_BITS - 1);    // odd powers
int padic_nl_for_prime_bits(int bits);
size_t padic_table_words(int nl, size_t blocks);
size_t padic_scratch_words(int nl, size_t blocks);
int padic_blocks_per_cu(int nl);           // resident workgroups per CU of the stage-A kernel for this limb count
bool launch_dec_a_padic(int nl, hipStream_t s, int gridx, const DecPadicParams& P, const uint32_t* ct,
                        uint32_t* u_out, int n, uint32_t* table);

// digit engine with base n for raw/DJN encryption (kernels_padic_enc.hpp)
struct EncPadicParams;
int padic_enc_nl_for_n_bits(int bits);
// optional precomputed window bases of a fixed-base table: plain residues modulo n^2, [windows][base_words], from k_sq_chain
struct FbBases {
    const uint32_t* bases_plain = nullptr;
    int base_words = 0;
    const uint32_t* kdig = nullptr;      // digit pairs of R^(i+2) mod n^2 (the ciphertext -> digit-form constants)
    int nd = 0;
};
bool launch_fb_table_padic(int nl, hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig,
                           const uint32_t* one_dig, uint32_t* table, int J, int wb, const FbBases& fb);
bool launch_fb_expand_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S,
                            uint32_t* T, int J, int h, uint32_t* mscratch);
struct PowPadicParams;
bool launch_pow_padic(int nl, hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n);
struct CtMulPadicParams;
size_t ctmul_padic_table_words(int nl, int wbits, size_t blocks);
bool launch_ctmul_padic(int nl, hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e,
                        uint32_t* out, int n);
struct MexpPadicParams;
bool launch_mexp_table_padic(int nl, hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* ct, const uint32_t* ct_inv, int nlanes);
bool launch_mexp_padic(int nl, hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* e, const uint8_t* sign, uint32_t* out, int nlanes);
// g-factoring of a finished digit-form table (kernels_padic_enc.hpp): passes 1 and 2 over `count` entries in chunks of K
bool padic_enc_gform_supported();
bool launch_fb_g_prefix_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K,
                              uint32_t* pref, uint32_t* tot, int tw, uint32_t* mscratch);
bool launch_fb_g_finish_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K,
                              const uint32_t* pref, const uint32_t* inv, int tw, uint32_t* mscratch);
bool launch_encrypt_padic(int nl, hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r,
                          const uint32_t* ct_in, uint32_t* ct_out, int n, int mode);

// digit pairs with base n on the lane-group engine (kernels_pair.hpp): DJN obfuscator / encryption for n of 2049 .. 4156 bits
struct PairParams;
struct PairCtMulParams;
int pair_nl_for_n_bits(int bits);                 // 112 / 144 limbs, 0 = not served
int pair_nl_for_prime_bits(int bits);             // 36 / 56 / 72 limbs (primes of keys up to 2048 / 3072 / 4096 bits), 0 = not served
int pair_epb(int nl);                             // elements per workgroup
bool launch_pair_fb_chain(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* bases,
                          const uint32_t* one_pair, uint32_t* S, int nwin, int h, const FbBases& fb);
bool launch_pair_fb_expand(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S,
                           uint32_t* T, int J, int h);
bool launch_pair_g_prefix(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K,
                          uint32_t* pref, uint32_t* tot, int tw);
bool launch_pair_g_finish(int nl, hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K,
                          const uint32_t* pref, const uint32_t* inv, int tw);
bool launch_pair_fixed_base(int nl, hipStream_t s, int grid, const PairParams& P, const uint32_t* m, const uint32_t* r,
                            uint32_t* wv_out, int n, int with_m);
bool launch_pair_ctmul(int nl, hipStream_t s, int grid, const PairCtMulParams& P, const uint32_t* ct, const uint32_t* e,
                       uint32_t* wv_out, int n);

// x = a^-1 mod M for `count` values of `words` 32-bit words each (words in {64,128,192,256}); *fail counts
// non-invertible inputs.  Returns false if `words` has no instantiation.
bool launch_inv_eea(hipStream_t s, int words, const uint32_t* mod, const uint32_t* a, uint32_t* out, int count,
                    int max_steps, int* fail);

}  // namespace pai
