// Instantiations of the lane-group digit-pair kernels (kernels_pair.hpp): 112 limbs on 4 lanes x 28 (n up to 3228 bits:
// 3072-bit keys; one wave per SIMD — two accumulator windows, both digits and the modulus slice are ~240 registers) and
// 144 limbs on 8 lanes x 18 (n up to 4156 bits: 4096-bit keys; ~150 registers, two waves per SIMD.  4 lanes x 36 needs
// ~310 registers: 1.6 KB of scratch per lane with the table prefetch).
#include "geo_ops.hpp"
#include "kernels_pair.hpp"

namespace pai {

template <class G>
struct PairLaunch {
    static constexpr int BYTES = PairLds<G>::BYTES;
    static void set_lds(const void* fn) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES); }
    static void chain(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* bases,
                      const uint32_t* one_pair, uint32_t* S, int nwin, int h, const FbBases& fb) {
        set_lds((const void*)k_pair_fb_chain<G>);
        hipLaunchKernelGGL(k_pair_fb_chain<G>, dim3(grid), dim3(BLOCK_THREADS), BYTES, s, nctx, nm1, bases, one_pair, S, nwin, h,
                           fb.bases_plain, fb.base_words, fb.kdig, fb.nd);
    }
    static void expand(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S, uint32_t* T, int J, int h) {
        set_lds((const void*)k_pair_fb_expand<G>);
        hipLaunchKernelGGL(k_pair_fb_expand<G>, dim3(grid), dim3(BLOCK_THREADS), BYTES, s, nctx, nm1, S, T, J, h);
    }
    static void fixed_base(hipStream_t s, int grid, const PairParams& P, const uint32_t* m, const uint32_t* r, uint32_t* wv_out,
                           int n, int with_m) {
        auto go = [&](auto kernel, int bytes) {
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK_THREADS), bytes, s, P, m, r, wv_out, n, with_m);
        };
        if (P.fb_gform) go(k_pair_fixed_base<G, true>, PairLds<G>::BYTES_FBG);
        else go(k_pair_fixed_base<G, false>, PairLds<G>::BYTES_FB);
    }
    static void g_prefix(hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K, uint32_t* pref,
                         uint32_t* tot, int tw) {
        (void)hipFuncSetAttribute((const void*)k_pair_g_prefix<G>, hipFuncAttributeMaxDynamicSharedMemorySize, PairGLds<G>::BYTES);
        hipLaunchKernelGGL(k_pair_g_prefix<G>, dim3(grid), dim3(BLOCK_THREADS), PairGLds<G>::BYTES, s, nctx, table, count, K, pref, tot, tw);
    }
    static void g_finish(hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K, const uint32_t* pref,
                         const uint32_t* inv, int tw) {
        (void)hipFuncSetAttribute((const void*)k_pair_g_finish<G>, hipFuncAttributeMaxDynamicSharedMemorySize, PairGLds<G>::BYTES);
        hipLaunchKernelGGL(k_pair_g_finish<G>, dim3(grid), dim3(BLOCK_THREADS), PairGLds<G>::BYTES, s, nctx, table, count, K, pref, inv, tw);
    }
    static void ctmul(hipStream_t s, int grid, const PairCtMulParams& P, const uint32_t* ct, const uint32_t* e, uint32_t* wv_out, int n) {
        (void)hipFuncSetAttribute((const void*)k_pair_ctmul<G>, hipFuncAttributeMaxDynamicSharedMemorySize, PairLds<G>::BYTES_CT);
        hipLaunchKernelGGL(k_pair_ctmul<G>, dim3(grid, P.nctx1 ? 2 : 1), dim3(BLOCK_THREADS), PairLds<G>::BYTES_CT, s, P, ct, e, wv_out, n);
    }
};
#ifndef PAIR_G112
#define PAIR_G112 Geo<28, 4, 7, false>     // 7-row blocks: 17.9 ms per 65536 at 3072-bit keys (4 rows: 18.9; 8 lanes x 14: 18.5)
#endif
#ifndef PAIR_G144
#define PAIR_G144 Geo<18, 8, 6, false>     // 44.3 ms at 4096-bit keys (3 / 9 rows: 44.1 / 44.0; 4 lanes x 36 without prefetch: 58.1)
#endif
using G112 = PAIR_G112;
using G144 = PAIR_G144;
static_assert(G112::NL == 112 && G144::NL == 144, "pair geometries");
using P112 = PairLaunch<G112>;
using P144 = PairLaunch<G144>;
// the table-conversion kernels run single Montgomery products (mont_mul: row blocks must divide its 24-row normalisation
// interval); the table layout does not depend on the row-block size
// 36 limbs on 4 lanes x 9: the PRIMES of keys up to 2048 bits — decryption stage A of mid-size batches (k_pair_ctmul with modulus
// s, exponent s - 1: paillier_capi.hip, mid_decrypt)
using G36 = Geo<9, 4, 3, false>;
using P36 = PairLaunch<G36>;
// ... 56 limbs on 4 x 14 and 72 limbs on 4 x 18: the primes of 3072- and 4096-bit keys
using G56 = Geo<14, 4, 7, false>;
using P56 = PairLaunch<G56>;
using G72 = Geo<18, 4, 6, false>;
using P72 = PairLaunch<G72>;
using P112C = PairLaunch<Geo<G112::NLL, G112::T, 4, false>>;
using P144C = PairLaunch<Geo<G144::NLL, G144::T, 6, false>>;

int pair_nl_for_n_bits(int bits) {
    if (bits <= 2048) return 0;                    // the one-element-per-lane digit engine serves those
    if (RB * 112 >= bits + 20) return 112;
    if (RB * 144 >= bits + 20) return 144;
    return 0;
}
int pair_nl_for_prime_bits(int bits) { return RB * 36 >= bits + 20 ? 36 : (RB * 56 >= bits + 20 ? 56 : (RB * 72 >= bits + 20 ? 72 : 0)); }
int pair_epb(int nl) { return nl == 112 ? G112::EPB : (nl == 144 ? G144::EPB : (nl == 36 || nl == 56 || nl == 72 ? G36::EPB : 0)); }
bool launch_pair_fb_chain(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* bases,
                          const uint32_t* one_pair, uint32_t* S, int nwin, int h, const FbBases& fb) {
    if (nl == 112) P112::chain(s, grid, nctx, nm1, bases, one_pair, S, nwin, h, fb);
    else if (nl == 144) P144::chain(s, grid, nctx, nm1, bases, one_pair, S, nwin, h, fb);
    else return false;
    return true;
}
bool launch_pair_fb_expand(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* S,
                           uint32_t* T, int J, int h) {
    if (nl == 112) P112::expand(s, grid, nctx, nm1, S, T, J, h);
    else if (nl == 144) P144::expand(s, grid, nctx, nm1, S, T, J, h);
    else return false;
    return true;
}
bool launch_pair_fixed_base(int nl, hipStream_t s, int grid, const PairParams& P, const uint32_t* m, const uint32_t* r,
                            uint32_t* wv_out, int n, int with_m) {
    if (nl == 112) P112::fixed_base(s, grid, P, m, r, wv_out, n, with_m);
    else if (nl == 144) P144::fixed_base(s, grid, P, m, r, wv_out, n, with_m);
    else if (nl == 36) P36::fixed_base(s, grid, P, m, r, wv_out, n, with_m);         // (mid-size batches at keys up to 2048 bits: the
    else if (nl == 56) P56::fixed_base(s, grid, P, m, r, wv_out, n, with_m);         //  one-element-per-lane engine's table has the
    else if (nl == 72) P72::fixed_base(s, grid, P, m, r, wv_out, n, with_m);         //  same layout)
    else return false;
    return true;
}

bool launch_pair_g_prefix(int nl, hipStream_t s, int grid, const MontCtx* nctx, const uint32_t* table, size_t count, int K,
                          uint32_t* pref, uint32_t* tot, int tw) {
    if (nl == 112) P112C::g_prefix(s, grid, nctx, table, count, K, pref, tot, tw);
    else if (nl == 144) P144C::g_prefix(s, grid, nctx, table, count, K, pref, tot, tw);
    else return false;
    return true;
}
bool launch_pair_g_finish(int nl, hipStream_t s, int grid, const MontCtx* nctx, uint32_t* table, size_t count, int K,
                          const uint32_t* pref, const uint32_t* inv, int tw) {
    if (nl == 112) P112C::g_finish(s, grid, nctx, table, count, K, pref, inv, tw);
    else if (nl == 144) P144C::g_finish(s, grid, nctx, table, count, K, pref, inv, tw);
    else return false;
    return true;
}

bool launch_pair_ctmul(int nl, hipStream_t s, int grid, const PairCtMulParams& P, const uint32_t* ct, const uint32_t* e,
                       uint32_t* wv_out, int n) {
    if (nl == 112) P112::ctmul(s, grid, P, ct, e, wv_out, n);
    else if (nl == 144) P144::ctmul(s, grid, P, ct, e, wv_out, n);
    else if (nl == 36) P36::ctmul(s, grid, P, ct, e, wv_out, n);
    else if (nl == 56) P56::ctmul(s, grid, P, ct, e, wv_out, n);
    else if (nl == 72) P72::ctmul(s, grid, P, ct, e, wv_out, n);
    else return false;
    return true;
}

}  // namespace pai
