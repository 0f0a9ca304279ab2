// 36-limb instantiations of the base-n digit-pair kernels: n of 700..1024 bits (1024-bit keys), 12-row blocks.
#include "padic_enc_launch.hpp"

namespace pai {

using L36 = EncLaunch<36, 12>;
void enc36_fb_table(hipStream_t s, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* hs_dig, const uint32_t* one_dig,
                    uint32_t* table, int J, int wb, const FbBases& fb) { L36::fb_table(s, nctx, nm1, hs_dig, one_dig, table, J, wb, fb); }
void enc36_fb_expand(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S, uint32_t* T, int J, int h,
                     uint32_t* mscratch) { L36::fb_expand(s, grid, nctx, nm1, S, T, J, h, mscratch); }
void enc36_encrypt(hipStream_t s, int grid, const EncPadicParams& P, const uint32_t* m, const uint32_t* r, const uint32_t* ct_in,
                   uint32_t* ct_out, int n, int mode) { L36::encrypt(s, grid, P, m, r, ct_in, ct_out, n, mode); }
void enc36_ctmul(hipStream_t s, int grid, const CtMulPadicParams& P, const uint32_t* ct, const uint32_t* e, uint32_t* out, int n) {
    L36::ctmul(s, grid, P, ct, e, out, n);
}
void enc36_g_prefix(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K, uint32_t* pref, uint32_t* tot,
                    int tw, uint32_t* mscratch) { L36::g_prefix(s, grid, nctx, table, count, K, pref, tot, tw, mscratch); }
void enc36_g_finish(hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K, const uint32_t* pref,
                    const uint32_t* inv, int tw, uint32_t* mscratch) { L36::g_finish(s, grid, nctx, table, count, K, pref, inv, tw, mscratch); }
void enc36_pow(hipStream_t s, int grid, const PowPadicParams& P, const uint32_t* base, uint32_t* out, int n) { L36::pow(s, grid, P, base, out, n); }

void enc36_mexp_table(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* ct, const uint32_t* ct_inv, int nlanes) {
    L36::mexp_table(s, grid, P, ct, ct_inv, nlanes);
}
void enc36_mexp(hipStream_t s, int grid, const MexpPadicParams& P, const uint32_t* e, const uint8_t* sign, uint32_t* out, int nlanes) {
    L36::mexp(s, grid, P, e, sign, out, nlanes);
}

}  // namespace pai
