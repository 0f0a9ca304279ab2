// Instantiations of the base-n digit-pair kernels (kernels_padic_enc.hpp): fixed-base table construction, raw / DJN
// encryption, apply_obfuscator, ciphertext * plaintext and the standard scheme's r^n.  72 limbs here (n of 1400..2048
// bits), 36 limbs in padic_enc36_kernels.hip (n of 700..1024 bits).  Own translation units (see padic_dec_kernels.hip).
#include "padic_enc_launch.hpp"

namespace pai {

using L72 = EncLaunch<72, 8>;     // rows per block of the 72-limb products: 8
// k_encrypt_padic runs 12-row blocks: with the fused product rule (two accumulator windows, 320+ registers, parked in
// AGPRs between row blocks) fewer, larger blocks mean fewer window round trips — 67.2 vs 69.8 ms per 2^20; ct x pt
// prefers 8 rows (4.87 vs 5.07 ms per 65536) and r^n does not care (144-150 ms per 65536 either way)
using L72E = EncLaunch<72, 12>;

int padic_enc_nl_for_n_bits(int bits) {
    if (bits >= 700 && RB * 36 >= bits + 20) return 36;
    if (bits >= 1400 && RB * 72 >= bits + 20) return 72;
    return 0;
}
bool launch_fb_table_padic(int nl, hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig,
                           const uint32_t* one_dig, uint32_t* table, int J, int wb, const FbBases& fb) {
    if (nl == 72) L72::fb_table(s, nctx, nm1, hs_dig, one_dig, table, J, wb, fb);
    else if (nl == 36) enc36_fb_table(s, nctx, nm1, hs_dig, one_dig, table, J, wb, fb);
    else return false;
    return true;
}
bool launch_fb_expand_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S,
                            uint32_t* T, int J, int h, uint32_t* mscratch) {
    if (nl == 72) L72::fb_expand(s, grid, nctx, nm1, S, T, J, h, mscratch);
    else if (nl == 36) enc36_fb_expand(s, grid, nctx, nm1, S, T, J, h, mscratch);
    else return false;
    return true;
}
bool launch_encrypt_padic(int nl, hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r,
                          const uint32_t* ct_in, uint32_t* ct_out, int n, int mode) {
    if (nl == 72) L72E::encrypt(s, grid, P, m, r, ct_in, ct_out, n, mode);
    else if (nl == 36) enc36_encrypt(s, grid, P, m, r, ct_in, ct_out, n, mode);
    else return false;
    return true;
}
bool padic_enc_gform_supported() { return PAI_ENC_GFORM_OK; }
bool launch_fb_g_prefix_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K,
                              uint32_t* pref, uint32_t* tot, int tw, uint32_t* mscratch) {
    if (nl == 72) L72E::g_prefix(s, grid, nctx, table, count, K, pref, tot, tw, mscratch);
    else if (nl == 36) enc36_g_prefix(s, grid, nctx, table, count, K, pref, tot, tw, mscratch);
    else return false;
    return true;
}
bool launch_fb_g_finish_padic(int nl, hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K,
                              const uint32_t* pref, const uint32_t* inv, int tw, uint32_t* mscratch) {
    if (nl == 72) L72E::g_finish(s, grid, nctx, table, count, K, pref, inv, tw, mscratch);
    else if (nl == 36) enc36_g_finish(s, grid, nctx, table, count, K, pref, inv, tw, mscratch);
    else return false;
    return true;
}
size_t ctmul_padic_table_words(int nl, int wbits, size_t blocks) { return ((size_t)1 << wbits) * 2 * nl * blocks * BLOCK_THREADS; }
bool launch_ctmul_padic(int nl, hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e,
                        uint32_t* out, int n) {
    if (nl == 72) L72::ctmul(s, grid, P, ct, e, out, n);
    else if (nl == 36) enc36_ctmul(s, grid, P, ct, e, out, n);
    else return false;
    return true;
}
bool launch_pow_padic(int nl, hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n) {
    if (nl == 72) L72::pow(s, grid, P, base, out, n);
    else if (nl == 36) enc36_pow(s, grid, P, base, out, n);
    else return false;
    return true;
}

bool launch_mexp_table_padic(int nl, hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* ct, const uint32_t* ct_inv, int nlanes) {
    if (nl == 72) L72::mexp_table(s, grid, P, ct, ct_inv, nlanes);
    else if (nl == 36) enc36_mexp_table(s, grid, P, ct, ct_inv, nlanes);
    else return false;
    return true;
}
bool launch_mexp_padic(int nl, hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* e, const uint8_t* sign, uint32_t* out, int nlanes) {
    if (nl == 72) L72::mexp(s, grid, P, e, sign, out, nlanes);
    else if (nl == 36) enc36_mexp(s, grid, P, e, sign, out, nlanes);
    else return false;
    return true;
}

}  // namespace pai
