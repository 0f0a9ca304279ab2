// Sustained (seconds, power-limited) v_mad_u64_u32 rate on gfx950: the roofline denominator of bench.py at the power cap, next
// to the burst figure of tools/ubench_valu.hip (an 18 ms run).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_sustained.hip -o tools/ubench_sustained
// Usage: tools/ubench_sustained [seconds_per_config=3] [form=0|1|2]
//   form 0: asm v_mad_u64_u32 with the carry-out in vcc, 16 independent chains per lane (what ubench_valu.hip measures)
//   form 1: compiler-scheduled  acc[k] += (u64)(u32)acc[(k+5)&15] * b  (carry-out in whatever SGPR pair the compiler picks, the
//           multiplier taken from another chain's previous value: the dependency pattern of a product loop)
//   form 2: form 1 with a 4:1 mix of multiplies and 32-bit adds/ands (the product kernels' instruction mix: ~79 % multiplies)
// One launch is sized to ~20-40 ms; launches repeat back to back for the requested time; every launch is timed with HIP events
// and the rate of the first launch, of all launches, and of the last second is printed (one JSON line per waves-per-SIMD setting).
// tools/ubench_sustained.sh samples rocm-smi power / sclk beside it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NACC = 16;
constexpr int REP = 4;

template <int FORM>
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, int iters, uint32_t seed_a, uint32_t seed_b) {
    uint64_t acc[NACC];
    uint32_t a = seed_a * (threadIdx.x * 2 + 1) + 12345u, b = seed_b + threadIdx.x * 7u;
    uint32_t t = a ^ b;
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = ((uint64_t)(a + k) << 20) | (b ^ k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if constexpr (FORM == 0) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
                } else {
                    acc[k] += (uint64_t)(uint32_t)acc[(k + 5) & 15] * (uint32_t)(b + r);
                    if constexpr (FORM == 2) {
                        if ((k & 3) == 3) t = (t + (uint32_t)acc[k]) & 0x1fffffffu;
                    }
                }
            }
        }
    }
    uint64_t s = t;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s ^= acc[k];
    if (s == 0x123456789abcdefull) out[0] = (uint32_t)s;
}

template <int FORM>
static void run(int waves_per_simd, double seconds, uint32_t* d_out, int ncu, int form) {
    int blocks = ncu * waves_per_simd;
    int iters = 60000 / waves_per_simd * (waves_per_simd == 1 ? 1 : 2);      // ~20-40 ms per launch
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms_all;
    std::vector<double> t_end;
    auto T0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < seconds) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mad<FORM>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u, 5u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
        ms_all.push_back(ms); t_end.push_back(el);
    }
    double macs_per_launch = (double)blocks * 256.0 * iters * NACC * REP;
    double sum_all = 0, sum_last = 0; int n_last = 0;
    for (size_t i = 0; i < ms_all.size(); ++i) {
        sum_all += ms_all[i];
        if (t_end[i] > el - 1.0) { sum_last += ms_all[i]; ++n_last; }
    }
    double r_first = macs_per_launch / (ms_all[0] * 1e-3), r_all = macs_per_launch * ms_all.size() / (sum_all * 1e-3),
           r_last = macs_per_launch * n_last / (sum_last * 1e-3);
    printf("{\"op\": \"v_mad_u64_u32\", \"form\": %d, \"waves_per_simd\": %d, \"seconds\": %.2f, \"launches\": %zu, \"ms_first\": %.3f, "
           "\"ms_last\": %.3f, \"T_mac_s_first_launch\": %.3f, \"T_mac_s_all\": %.3f, \"T_mac_s_last_second\": %.3f, "
           "\"kernel_busy_frac\": %.3f}\n",
           form, waves_per_simd, el, ms_all.size(), ms_all[0], ms_all.back(), r_first / 1e12, r_all / 1e12, r_last / 1e12,
           sum_all * 1e-3 / el);
    fflush(stdout);
}

int main(int argc, char** argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    int form = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    uint32_t* d_out; CK(hipMalloc(&d_out, 4096));
    for (int w : {1, 2, 4, 8}) {
        if (form == 0) run<0>(w, seconds, d_out, ncu, form);
        else if (form == 1) run<1>(w, seconds, d_out, ncu, form);
        else run<2>(w, seconds, d_out, ncu, form);
    }
    CK(hipFree(d_out));
    return 0;
}
