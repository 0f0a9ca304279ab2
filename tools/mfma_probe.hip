// Time-boxed probe ("next" item 6): can the matrix cores take the wave-uniform-modulus half of a Montgomery
// reduction off the vector multiplier?  Microbenchmark only — nothing in the library uses this.
//
// One element per lane (64 per wave), as in the digit-pair decrypt kernel.  The SEPARATED reduction of a double-width
// value T modulo a wave-uniform p is   m = T_lo * p' mod R,   t = (T + m p) / R   — two products by constants.  With 7-bit
// digits (signed int8 operands stay non-negative) both are  [digits of the constant as a Toeplitz matrix] x [digit
// columns of the 64 elements]  on v_mfma_i32_32x32x32_i8: the constant is the A operand (rows = output digit positions), the
// elements are the B operand (columns), so inputs and outputs stay "one element per lane" up to a v_permlane32_swap.
//   R = 2^1120 = 160 digits = 5 K-blocks of 32;  T_lo arrives as 39 limbs of 29 bits, t leaves as 39 limbs of 29 bits.
// Per reduction: limbs -> spread digits (VALU), 20 lane-half swaps, 30 MFMAs (lower-triangular p' Toeplitz), 80 swaps of
// the accumulators, carry-normalisation of 160 column sums into digits (VALU), 20 swaps, 30 MFMAs (the high half of
// m p), 80 swaps, recombination of 160 column sums into 29-bit limbs (VALU).
// Against it: what ONE reduction costs on the vector multiplier today: 36^2 v_mad_u64_u32 (digit-serial, lazy 64-bit columns).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
// prints one JSON line: cycles per reduction and wave for both forms, a correctness check of the MFMA form against a
// host big-integer evaluation, and the clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ND = 160;          // 7-bit digits per operand
constexpr int NKB = ND / 32;     // K blocks
constexpr int NLIMB = 39;        // 29-bit limbs covering 1120 bits (39 * 29 = 1131)
constexpr int NDW = ND / 4;      // dwords of spread digits (4 digits, one per byte)

__device__ __forceinline__ void swap32(int& lo_src, int& hi_src) {
    // after: first = {lower lanes: first, upper lanes: second of the lane 32 below}; second = {lower: first of the lane 32 above, upper: second}
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)lo_src, (unsigned)hi_src, false, false);
    lo_src = (int)r[0];
    hi_src = (int)r[1];
}

// 28-bit field -> 4 bytes of 7 bits
__device__ __forceinline__ uint32_t spread28(uint32_t f) {
    const uint32_t g = ((f & 0xfffc000u) << 2) | (f & 0x3fffu);
    return ((g & 0x3f803f80u) << 1) | (g & 0x007f007fu);
}

// limbs (29-bit) -> spread digit dwords
__device__ __forceinline__ void limbs_to_digits(const uint32_t (&x)[NLIMB], int (&d)[NDW]) {
#pragma unroll
    for (int w = 0; w < NDW; ++w) {
        const int bit = 28 * w, j = bit / 29, off = bit % 29;
        uint32_t f = x[j] >> off;
        if (off > 1 && j + 1 < NLIMB) f |= x[j + 1] << (29 - off);
        d[w] = (int)spread28(f & 0xfffffffu);
    }
}

// one MFMA pass: out[cb][I] (I = row block) += sum_Kb Afrag[dmap(I, Kb)] x B_cb[Kb]
// LOW: rows 0..159 (I = 0..4, delta = I - Kb in 0..4); HIGH: rows 160..319 (I' = 5..9, delta = I' - Kb in 1..5)
template <bool HIGH>
__device__ __forceinline__ void toeplitz_mfma(const v4i (&afrag)[6], const v4i (&b0)[NKB], const v4i (&b1)[NKB], v16i (&acc0)[NKB],
                                              v16i (&acc1)[NKB]) {
#pragma unroll
    for (int I = 0; I < NKB; ++I) {
        v16i c0 = {}, c1 = {};
#pragma unroll
        for (int Kb = 0; Kb < NKB; ++Kb) {
            const int delta = (HIGH ? I + NKB : I) - Kb;
            if (delta < (HIGH ? 1 : 0) || delta > 5) continue;
            if (!HIGH && delta > 4) continue;
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[delta], b0[Kb], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[delta], b1[Kb], c1, 0, 0, 0);
        }
        acc0[I] = c0;
        acc1[I] = c1;
    }
}

// spread digit dwords of this lane's element -> B operands of the two column blocks (elements 0..31 / 32..63)
__device__ __forceinline__ void digits_to_b(int (&d)[NDW], v4i (&b0)[NKB], v4i (&b1)[NKB]) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int first = d[8 * kb + i], second = d[8 * kb + 4 + i];
            swap32(first, second);
            b0[kb][i] = first;
            b1[kb][i] = second;
        }
    }
}

// accumulators -> the 160 column sums of this lane's own element: col[32 I + row]
__device__ __forceinline__ void acc_to_cols(v16i (&acc0)[NKB], v16i (&acc1)[NKB], int (&col)[ND]) {
#pragma unroll
    for (int I = 0; I < NKB; ++I) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int a = acc0[I][r], b = acc1[I][r];
            swap32(a, b);
            const int row = (r & 3) + 8 * (r >> 2);
            col[32 * I + row] = a;
            col[32 * I + row + 4] = b;
        }
    }
}

// column sums (each < 2^22) -> spread 7-bit digits, carries propagated, truncated to ND digits
__device__ __forceinline__ void cols_to_digits(const int (&col)[ND], int (&d)[NDW]) {
    uint64_t c = 0;
#pragma unroll
    for (int w = 0; w < NDW; ++w) {
        uint64_t v = c + (uint32_t)col[4 * w];
        v += (uint64_t)(uint32_t)col[4 * w + 1] << 7;
        v += (uint64_t)(uint32_t)col[4 * w + 2] << 14;
        v += (uint64_t)(uint32_t)col[4 * w + 3] << 21;
        d[w] = (int)spread28((uint32_t)v & 0xfffffffu);
        c = v >> 28;
    }
}

// column sums -> 29-bit limbs (the value's bits [0, 1131))
__device__ __forceinline__ void cols_to_limbs(const int (&col)[ND], uint32_t (&x)[NLIMB]) {
    uint32_t f28[NDW];
    uint64_t c = 0;
#pragma unroll
    for (int w = 0; w < NDW; ++w) {
        uint64_t v = c + (uint32_t)col[4 * w];
        v += (uint64_t)(uint32_t)col[4 * w + 1] << 7;
        v += (uint64_t)(uint32_t)col[4 * w + 2] << 14;
        v += (uint64_t)(uint32_t)col[4 * w + 3] << 21;
        f28[w] = (uint32_t)v & 0xfffffffu;
        c = v >> 28;
    }
#pragma unroll
    for (int j = 0; j < NLIMB; ++j) {
        const int bit = 29 * j, w = bit / 28, off = bit % 28;
        uint32_t v = w < NDW ? f28[w] >> off : 0u;
        if (w + 1 < NDW) v |= f28[w + 1] << (28 - off);
        x[j] = v & 0x1fffffffu;
    }
}

// A fragments: afrag[delta][lane] = 16 bytes: digit[32 delta + (lane & 31) - 16 (lane >> 5) - j], j = 0..15
__global__ void __launch_bounds__(64, 1)
k_mfma_reduce(const v4i* __restrict__ frag_pinv, const v4i* __restrict__ frag_p, const uint32_t* __restrict__ tlo, uint32_t* __restrict__ out,
              int iters, unsigned long long* cycles) {
    const int lane = threadIdx.x;
    v4i fa[6], fb[6];
#pragma unroll
    for (int dl = 0; dl < 6; ++dl) { fa[dl] = frag_pinv[dl * 64 + lane]; fb[dl] = frag_p[dl * 64 + lane]; }
    uint32_t x[NLIMB];
#pragma unroll
    for (int j = 0; j < NLIMB; ++j) x[j] = tlo[((size_t)blockIdx.x * 64 + lane) * NLIMB + j];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        int d[NDW];
        limbs_to_digits(x, d);
        v4i b0[NKB], b1[NKB];
        digits_to_b(d, b0, b1);
        v16i a0[NKB], a1[NKB];
        toeplitz_mfma<false>(fa, b0, b1, a0, a1);              // m = T_lo p' (low 160 digits)
        int col[ND];
        acc_to_cols(a0, a1, col);
        cols_to_digits(col, d);
        digits_to_b(d, b0, b1);
        toeplitz_mfma<true>(fb, b0, b1, a0, a1);               // (m p) >> 1120
        acc_to_cols(a0, a1, col);
        cols_to_limbs(col, x);                                  // feeds the next iteration: a dependent chain, as in a modexp
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < NLIMB; ++j) out[((size_t)blockIdx.x * 64 + lane) * NLIMB + j] = x[j];
    if (lane == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// what the same ONE reduction costs on the vector multiplier today: digit-serial Montgomery, 36^2 multiply-accumulates of the
// quotient digits with the modulus limbs (SGPR operands, lazy 64-bit columns, carries every 12 rows — the digit-pair
// engine's arrangement; the quotient digit is one v_mul_lo per row, no separate T_lo p' product exists there).  A squaring
// of the digit-pair engine contains two such reductions (w and v): 2 x 36^2 = 57 % of its multiplies.
__global__ void __launch_bounds__(64, 1)
k_valu_reduce(const uint32_t* __restrict__ pl, const uint32_t* __restrict__ tlo, uint32_t* __restrict__ out, int iters, unsigned long long* cycles) {
    const int lane = threadIdx.x;
    constexpr int NL = 36;
    uint32_t x[NL], p[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) { x[j] = tlo[((size_t)blockIdx.x * 64 + lane) * NLIMB + j]; p[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl[j]); }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int half = 0; half < 1; ++half) {          // ONE reduction: q p accumulated digit-serially (36^2 multiply-accumulates)
            uint64_t acc[NL + 12];
#pragma unroll
            for (int j = 0; j < NL + 12; ++j) acc[j] = x[j % NL];
#pragma unroll 1
            for (int blk = 0; blk < NL / 12; ++blk) {
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const uint32_t q = ((uint32_t)acc[u] * 0x12345671u) & 0x1fffffffu;
#pragma unroll
                    for (int j = 0; j < NL; ++j) acc[j + u] += (uint64_t)p[j] * q;
                    acc[u + 1] += acc[u] >> 29;
                }
#pragma unroll
                for (int j = 0; j < NL; ++j) acc[j] = acc[j + 12];
#pragma unroll
                for (int u = 0; u < 12; ++u) acc[NL + u] = 0;
#pragma unroll
                for (int j = NL + 11; j >= 1; --j) acc[j] = (acc[j] & 0x1fffffffu) + (acc[j - 1] >> 29);
                acc[0] &= 0x1fffffffu;
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) x[j] = (uint32_t)acc[j] & 0x1fffffffu;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < NL; ++j) out[((size_t)blockIdx.x * 64 + lane) * NLIMB + j] = x[j];
    if (lane == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---- host big integers in 7-bit digits ---------------------------------------------------------------------------------
using Dig = std::vector<int>;
static Dig mul_digits(const Dig& a, const Dig& b, size_t keep_lo, size_t from) {   // columns [from, from + keep_lo) of a b, carry-propagated from `from` on
    // (the carry out of the columns below `from` is, in a Montgomery reduction, known without computing them: T_lo != 0)
    std::vector<long long> c(a.size() + b.size() + 1, 0);
    for (size_t i = 0; i < a.size(); ++i)
        for (size_t j = 0; j < b.size(); ++j) c[i + j] += (long long)a[i] * b[j];
    long long carry = 0;
    Dig out;
    for (size_t i = from; i < c.size(); ++i) {          // carries start at `from`: what the columns [from, ...) alone hold
        long long t = c[i] + carry;
        int dgt = (int)(t & 127);
        carry = t >> 7;
        if (i >= from && out.size() < keep_lo) out.push_back(dgt);
    }
    while (out.size() < keep_lo) out.push_back(0);
    return out;
}

int main(int argc, char** argv) {
    // mfma_probe [iters] [mode]: mode 1 = MFMA form only, 2 = VALU form only (long runs for power sampling), 0 = both
    const int iters_arg = argc > 1 ? atoi(argv[1]) : 200, mode = argc > 2 ? atoi(argv[2]) : 0;
    srand(12345);
    Dig pinv(ND), p(ND);
    for (int i = 0; i < ND; ++i) { pinv[i] = rand() & 127; p[i] = rand() & 127; }
    p[ND - 1] = 127;
    auto frags = [&](const Dig& c) {
        std::vector<int> f(6 * 64 * 4, 0);
        for (int dl = 0; dl < 6; ++dl)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 16; ++j) {
                    const int idx = 32 * dl + (lane & 31) - 16 * (lane >> 5) - j;
                    const int v = (idx >= 0 && idx < ND) ? c[idx] : 0;
                    f[(dl * 64 + lane) * 4 + j / 4] |= v << (8 * (j % 4));
                }
        return f;
    };
    std::vector<int> fpi = frags(pinv), fp = frags(p);
    const int NB = 1024;                                  // one wave per SIMD
    std::vector<uint32_t> tl((size_t)NB * 64 * NLIMB);
    for (auto& v : tl) v = ((uint32_t)rand() << 15 ^ (uint32_t)rand()) & 0x1fffffffu;
    for (size_t e = 0; e < (size_t)NB * 64; ++e) tl[e * NLIMB + NLIMB - 1] &= (1u << (1120 - 29 * 38)) - 1u;   // < 2^1120
    int *d_fpi, *d_fp;
    uint32_t *d_tl, *d_out, *d_pl;
    unsigned long long* d_cyc;
    CK(hipMalloc(&d_fpi, fpi.size() * 4));
    CK(hipMalloc(&d_fp, fp.size() * 4));
    CK(hipMalloc(&d_tl, tl.size() * 4));
    CK(hipMalloc(&d_out, tl.size() * 4));
    CK(hipMalloc(&d_pl, 64 * 4));
    CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_fpi, fpi.data(), fpi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_fp, fp.data(), fp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tl, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pl, tl.data(), 64 * 4, hipMemcpyHostToDevice));
    // ---- correctness: one iteration, a handful of elements, against the host digit arithmetic ----
    hipLaunchKernelGGL(k_mfma_reduce, dim3(1), dim3(64), 0, 0, (const v4i*)d_fpi, (const v4i*)d_fp, d_tl, d_out, 1, d_cyc);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> got(64 * NLIMB);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int e = 0; e < 64; ++e) {
        Dig t(ND);
        for (int i = 0; i < ND; ++i) {
            const int bit = 7 * i, j = bit / 29, off = bit % 29;
            uint64_t v = tl[(size_t)e * NLIMB + j] >> off;
            if (j + 1 < NLIMB) v |= (uint64_t)tl[(size_t)e * NLIMB + j + 1] << (29 - off);
            t[i] = (int)(v & 127);
        }
        Dig m = mul_digits(t, pinv, ND, 0);
        Dig hi = mul_digits(m, p, ND + 2, ND);            // digits 160.. of m p
        for (int j = 0; j < NLIMB; ++j) {
            uint64_t want = 0;
            for (int b = 0; b < 29; ++b) {
                const int bit = 29 * j + b, dg = bit / 7, o = bit % 7;
                if (dg < (int)hi.size()) want |= (uint64_t)((hi[dg] >> o) & 1) << b;
            }
            if (got[(size_t)e * NLIMB + j] != (uint32_t)want) { ++bad; break; }
        }
    }
    // ---- timing ----
    const int iters = iters_arg;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_mfma = 0, ms_valu = 0;
    unsigned long long cyc_mfma = 0, cyc_valu = 0;
    for (int rep = 0; rep < 3; ++rep) {
        if (mode != 2) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma_reduce, dim3(NB), dim3(64), 0, 0, (const v4i*)d_fpi, (const v4i*)d_fp, d_tl, d_out, iters, d_cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_mfma, e0, e1));
            CK(hipMemcpy(&cyc_mfma, d_cyc, 8, hipMemcpyDeviceToHost));
        }
        if (mode != 1) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_valu_reduce, dim3(NB), dim3(64), 0, 0, d_pl, d_tl, d_out, iters, d_cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_valu, e0, e1));
            CK(hipMemcpy(&cyc_valu, d_cyc, 8, hipMemcpyDeviceToHost));
        }
    }
    printf("{\"probe\": \"separated Montgomery reduction of 64 elements per wave: MFMA i8 Toeplitz vs v_mad_u64_u32\", \"mismatching_elements_of_64\": %d, "
           "\"waves\": %d, \"iters\": %d, \"mfma_ms\": %.3f, \"valu_ms\": %.3f, \"mfma_us_per_reduction\": %.3f, \"valu_us_per_reduction\": %.3f, "
           "\"mfma_refclk_per_reduction\": %.0f, \"valu_refclk_per_reduction\": %.0f, \"speedup_mfma_over_valu\": %.3f}\n",
           bad, NB, iters, ms_mfma, ms_valu, 1e3 * ms_mfma / iters, 1e3 * ms_valu / iters, (double)cyc_mfma / iters, (double)cyc_valu / iters,
           (ms_mfma > 0 && ms_valu > 0) ? ms_valu / ms_mfma : 0.0);
    return 0;
}
