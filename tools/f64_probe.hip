// FP64-FMA limb probe (round 6; VERDICT r05 "next" #1): one digit-pair SQUARING modulo s^2 with the limbs held
// as doubles and every limb product split into an exact (hi, lo) pair by two v_fma_f64 — the GPU's 53-bit multiplier in place
// of v_mad_u64_u32 on 29-bit limbs (mont_padic.hpp: Padic<36,12>::sqr, 4 554 MACs per squaring of a 1024-bit s).
//
// Representation.  x in Z/s^2 as (a, b), a + b s == x R (mod s^2), R = 2^(LB NL); a limb is a double holding a SIGNED integer
// |limb| <= 2^(LB-1) (round-to-nearest splits give centred digits for free: no round-toward-zero MODE switch, no +R s term
// for positivity; |a|, |b| stay < 0.63 s through any number of squarings because |m| <= R/2).
// Product rule (mont_padic.hpp:6-13):  w = (a^2 + m s)/R,  v = (2 a b - m + m' s)/R.
// Limb product x*y (|x y| < 2^(2 LB)):  with H a running sum that is a multiple of 2^LB near C1 = 1.5 * 2^(LB+52)
//        h = fma(x, y, H)            -- H + [x y rounded to a multiple of 2^LB]      (exact: one rounding, at 2^LB)
//        l = fma(x, y, H - h)        -- the rounding error, |l| <= 2^(LB-1)          (exact; H - h is exact by Sterbenz)
//        L += l ; H = h              -- the column's low sum; the high sum rides in the FMA addend
// i.e. FOUR FP64 instructions per limb product (fma, add, fma, add).  Two-FMA-only forms do not exist: the low part needs the
// individual high part (H - h), and its sum does not fit the addend of the next product's FMA (|H - h| ~ 2^(2LB) against
// L < 2^53).  Columns are produced by product scanning; a column's value is L_c + (H_{c-1} - C1)/2^LB + carry.
// Quotient digit of column c: r = centred low digit of the column, m_c = centred low digit of r * (-s^-1 mod 2^LB) — three more
// FMAs and three adds.
//
// Geometries: <22, 47> (every bound below 2^53 in the worst case: 34 low parts of <= 2^46) and the optimistic <21, 49> the
// review priced (3.5 * 21^2 = 1 544 limb products; its low sums can exceed 2^53 in the worst case — a product kernel could not
// ship it without a mid-column split — but random operands stay exact, so it serves as the lower bound of the cost).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/f64_probe.hip -o tools/f64_probe
// Usage:  tools/f64_probe [iters=64] [dump.json|-] [only]     (tools/f64_probe_check.py verifies the dump against CPython integers)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include <string>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int I, int N, class F> __host__ __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int LB> struct K {
    static constexpr double B = (double)(1ull << LB);
    static constexpr double BINV = 1.0 / (double)(1ull << LB);
    static constexpr double C1 = 1.5 * (double)(1ull << 52) * (double)(1ull << LB);
    static constexpr double M52 = 1.5 * (double)(1ull << 52);
};

__host__ __device__ __forceinline__ void mac(double x, double y, double& H, double& L) {
    const double h = __builtin_fma(x, y, H);
    L += __builtin_fma(x, y, H - h);
    H = h;
}

// one reduction pass: columns of  P + q s  with quotient digits q (written to q[]), result limbs to out[].
// PROD(c, H, L) adds the products of column c of P (0 <= c < 2 NL - 1); SUB: subtract qsub[c] in column c < NL.
template <int NL, int LB, bool SUB, class PROD>
__host__ __device__ __forceinline__ void reduce_pass(PROD prod, const double (&s)[NL], double sinv, const double (&qsub)[NL],
                                            double (&q)[NL], double (&out)[NL]) {
    using C = K<LB>;
    double carry = 0.0, hprev = 0.0;
    static_for<0, 2 * NL>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        double H = C::C1, L = 0.0;
        if constexpr (c < 2 * NL - 1) prod(cc, H, L);
        static_for<0, NL>([&](auto ii) {               // q_i s_(c-i), digits already known
            constexpr int i = decltype(ii)::value, j = c - i;
            if constexpr (j >= 1 && j < NL) mac(q[i], s[j], H, L);      // j = 0 is the digit this column determines
        });
        double acc = L + hprev + carry;
        if constexpr (SUB && c < NL) acc -= qsub[c];
        if constexpr (c < NL) {
            const double t = __builtin_fma(acc, C::BINV, C::M52) - C::M52;      // round(acc / B)
            const double r = __builtin_fma(-t, C::B, acc);                      // centred low digit
            const double ph = __builtin_fma(r, sinv, C::C1);
            const double qc = __builtin_fma(r, sinv, C::C1 - ph);               // centred low digit of r * sinv
            q[c] = qc;
            const double h = __builtin_fma(qc, s[0], H);
            acc += __builtin_fma(qc, s[0], H - h);
            H = h;
            carry = acc * C::BINV;                                              // exact: acc == 0 (mod B)
        } else {
            const double t = __builtin_fma(acc, C::BINV, C::M52) - C::M52;
            out[c - NL] = __builtin_fma(-t, C::B, acc);
            carry = t;
        }
        hprev = (H - C::C1) * C::BINV;
    });
    out[NL - 1] += carry * C::B;                       // the sign / overflow stays in the top limb
}

template <int NL, int LB>
__host__ __device__ __forceinline__ void sqr_chain(const double (&s)[NL], double sinv, double (&a)[NL], double (&b)[NL], int iters) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        double a2[NL], m[NL], mp[NL], w[NL], v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) a2[i] = a[i] + a[i];
        reduce_pass<NL, LB, false>([&](auto cc, double& H, double& L) {
            static_for<0, NL>([&](auto ii) {
                constexpr int c = decltype(cc)::value, i = decltype(ii)::value, j = c - i;
                if constexpr (j >= 0 && j < NL && i <= j) {
                    if constexpr (i == j) mac(a[i], a[i], H, L); else mac(a2[i], a[j], H, L);
                }
            });
        }, s, sinv, m, m, w);
        reduce_pass<NL, LB, true>([&](auto cc, double& H, double& L) {
            static_for<0, NL>([&](auto ii) {
                constexpr int c = decltype(cc)::value, i = decltype(ii)::value, j = c - i;
                if constexpr (j >= 0 && j < NL) mac(a2[i], b[j], H, L);
            });
        }, s, sinv, m, mp, v);
#pragma unroll
        for (int i = 0; i < NL; ++i) { a[i] = w[i]; b[i] = v[i]; }
    }
}

template <int NL, int LB>
__global__ void __launch_bounds__(256, 1)
k_sqr_chain(const double* __restrict__ sg, double sinv, const double* __restrict__ in, double* __restrict__ out, int iters) {
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
    double a[NL], b[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) { a[i] = in[(size_t)i * n + slot]; b[i] = in[(size_t)(NL + i) * n + slot]; }
    double s[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) s[i] = sg[i];                                   // wave-uniform: scalar loads, SGPR pairs
    sqr_chain<NL, LB>(s, sinv, a, b, iters);
#pragma unroll
    for (int i = 0; i < NL; ++i) { out[(size_t)i * n + slot] = a[i]; out[(size_t)(NL + i) * n + slot] = b[i]; }
}

// ---- host: a tiny unsigned big integer on 64-bit words, only to build a random odd modulus and its limbs --------------------
static std::string hex_of_limbs(const std::vector<double>& v, size_t stride, size_t off, int nl) {
    std::string r = "[";
    for (int i = 0; i < nl; ++i) { char buf[64]; snprintf(buf, sizeof buf, "%s%.0f", i ? "," : "", v[(size_t)i * stride + off]); r += buf; }
    return r + "]";
}

template <int NL, int LB>
static void run(const char* name, int iters, int ncu, FILE* dump, int wps = 1) {
    const int grid = ncu * wps, n = grid * 256;
    std::mt19937_64 rng(11);
    const uint64_t mask = (1ull << LB) - 1;
    // modulus: 1024 random bits, odd, top bit set, as unsigned LB-bit limbs
    std::vector<uint64_t> su(NL, 0);
    {
        int bits = 1024;
        for (int i = 0; i < NL && bits > 0; ++i) { const int t = bits < LB ? bits : LB; su[i] = rng() & ((1ull << t) - 1); bits -= t; if (bits == 0) su[i] |= 1ull << (t - 1); }
        su[0] |= 1;
    }
    uint64_t inv = su[0]; for (int i = 0; i < 6; ++i) inv *= 2 - su[0] * inv;     // s^-1 mod 2^64
    int64_t sinv = (int64_t)((0 - inv) & mask); if (sinv >= (int64_t)(1ull << (LB - 1))) sinv -= (int64_t)(1ull << LB);
    std::vector<double> s(NL); for (int i = 0; i < NL; ++i) s[i] = (double)su[i];
    // operands: random limbs below the modulus' top (centred form is not required on entry)
    std::vector<double> in((size_t)2 * NL * n), out((size_t)2 * NL * n);
    for (int d = 0; d < 2; ++d)
        for (int i = 0; i < NL; ++i)
            for (int e = 0; e < n; ++e) {
                uint64_t lim = rng() & mask;
                if ((i + 1) * LB > 1023) { const int keep = 1023 - i * LB; lim = keep > 0 ? (lim & ((1ull << keep) - 1)) : 0; }
                in[((size_t)d * NL + i) * n + e] = (double)lim;
            }
    if (getenv("F64_PROBE_HOST")) {                    // the same templates on the host: the algorithm's check without a GPU
        const int picks[6] = {0, 1, 63, 64, n / 2 + 17, n - 1};
        double sh[NL]; for (int i = 0; i < NL; ++i) sh[i] = s[i];
        for (int k = 0; k < 6; ++k) {
            double a[NL], b[NL];
            for (int i = 0; i < NL; ++i) { a[i] = in[(size_t)i * n + picks[k]]; b[i] = in[((size_t)NL + i) * n + picks[k]]; }
            sqr_chain<NL, LB>(sh, (double)sinv, a, b, 3);
            for (int i = 0; i < NL; ++i) { out[(size_t)i * n + picks[k]] = a[i]; out[((size_t)NL + i) * n + picks[k]] = b[i]; }
        }
        if (dump) {
            fprintf(dump, "{\"NL\": %d, \"LB\": %d, \"squarings\": 3, \"where\": \"host\", \"s\": %s, \"cases\": [", NL, LB, hex_of_limbs(s, 1, 0, NL).c_str());
            for (int k = 0; k < 6; ++k) {
                const int e = picks[k];
                fprintf(dump, "%s{\"a\": %s, \"b\": %s, \"w\": %s, \"v\": %s}", k ? "," : "", hex_of_limbs(in, n, e, NL).c_str(),
                        hex_of_limbs(in, n, (size_t)NL * n + e, NL).c_str(), hex_of_limbs(out, n, e, NL).c_str(),
                        hex_of_limbs(out, n, (size_t)NL * n + e, NL).c_str());
            }
            fprintf(dump, "]}\n");
        }
        return;
    }
    double *ds, *din, *dout;
    CK(hipMalloc(&ds, NL * 8)); CK(hipMalloc(&din, in.size() * 8)); CK(hipMalloc(&dout, in.size() * 8));
    CK(hipMemcpy(ds, s.data(), NL * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // check run: 3 squarings
    hipLaunchKernelGGL((k_sqr_chain<NL, LB>), dim3(grid), dim3(256), 0, 0, ds, (double)sinv, din, dout, 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
    if (dump) {
        fprintf(dump, "{\"NL\": %d, \"LB\": %d, \"squarings\": 3, \"s\": %s, \"cases\": [", NL, LB, hex_of_limbs(s, 1, 0, NL).c_str());
        const int picks[6] = {0, 1, 63, 64, n / 2 + 17, n - 1};
        for (int k = 0; k < 6; ++k) {
            const int e = picks[k];
            fprintf(dump, "%s{\"a\": %s, \"b\": %s, \"w\": %s, \"v\": %s}", k ? "," : "", hex_of_limbs(in, n, e, NL).c_str(),
                    hex_of_limbs(in, n, (size_t)NL * n + e, NL).c_str(), hex_of_limbs(out, n, e, NL).c_str(),
                    hex_of_limbs(out, n, (size_t)NL * n + e, NL).c_str());
        }
        fprintf(dump, "]}\n");
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_sqr_chain<NL, LB>), dim3(grid), dim3(256), 0, 0, ds, (double)sinv, din, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double prods = 0.5 * NL * (NL + 1) + 3.0 * NL * NL;
    printf("{\"probe\": \"%s\", \"NL\": %d, \"limb_bits\": %d, \"waves_per_simd\": %d, \"elements\": %d, \"iters\": %d, \"ms\": %.3f, \"ns_per_squaring_per_elem\": %.4f, "
           "\"limb_products_per_squaring\": %.0f, \"T_limb_products_s\": %.3f, \"T_fp64_instr_s_at_4_per_product\": %.2f}\n",
           name, NL, LB, wps, n, iters, ms, ms * 1e6 / ((double)n * iters), prods, prods * n * iters / (ms * 1e-3) / 1e12,
           4.0 * prods * n * iters / (ms * 1e-3) / 1e12);
    fflush(stdout);
    CK(hipFree(ds)); CK(hipFree(din)); CK(hipFree(dout));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    FILE* dump = (argc > 2 && argv[2][0] != '-') ? fopen(argv[2], "w") : nullptr;
    int ncu = 256;
    if (!getenv("F64_PROBE_HOST")) { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    const int only = argc > 3 ? atoi(argv[3]) : 0;      // 0: all; 1: 22x47 one wave; 2: 22x47 two waves (power runs)
    if (only == 0 || only == 1) run<22, 47>("f64 digit-pair sqr 22 x 47 bits", iters, ncu, dump);
    if (only == 0) run<21, 49>("f64 digit-pair sqr 21 x 49 bits", iters, ncu, dump);
    if (only == 0 || only == 2) run<22, 47>("f64 digit-pair sqr 22 x 47 bits", iters, ncu, nullptr, 2);
    if (only == 0) run<21, 49>("f64 digit-pair sqr 21 x 49 bits", iters, ncu, nullptr, 2);
    if (dump) fclose(dump);
    return 0;
}
