// Data-format kernels either side of the Paillier hot path (HBM-bound, one row per lane):
//   k_fp_encode_f64   float64 -> (residue mod n as packed words, base-2 exponent)     fixedpoint.py:54-96
//   k_fp_encode_i64   int64 -> residue mod n, exponent 0                                  fixedpoint.py:72-74,89-96
//   k_fp_decode_i64   residue mod n -> signed 64-bit mantissa (+ "needs the exact host path" flag)  fixedpoint.py:98-115
//   k_draw_r          obfuscator randomness r < 2^randbits for DJN keys: ChaCha20 key stream (RFC 8439 block
//                     function, 256-bit key drawn from the OS CSPRNG by the caller), one 64-byte block per lane
// They move 8 B instead of 256 B per element over PCIe on the way in and out of the device.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pai {

// Encoding of a finite double x (the caller rejects NaN/Inf as the reference's int() would):
//   |x| < 1e-200 (zeros and subnormals included)  ->  mantissa 0, exponent 0            (fixedpoint.py:64-65,72-74)
//   otherwise  exponent = 53 - frexp(x).exp,  mantissa = x * 2^exponent = +-(2^52 | fraction bits), exactly
//   residue = mantissa mod n  (n > 2^66, so |mantissa| <= max_int always holds)
__global__ void __launch_bounds__(256)
k_fp_encode_f64(const double* __restrict__ x, const uint32_t* __restrict__ n_words_ptr, int nw, uint32_t* __restrict__ out,
                int32_t* __restrict__ expo, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double v = x[i];
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const bool tiny = fabs(v) < 1e-200;
    const int e = (int)((bits >> 52) & 0x7FF);
    const uint64_t mant = tiny ? 0ull : ((bits & 0xFFFFFFFFFFFFFull) | (1ull << 52));
    const bool neg = !tiny && (bits >> 63);
    expo[i] = tiny ? 0 : (1075 - e);
    uint32_t* row = out + i * (size_t)nw;
    if (!neg) {
        row[0] = (uint32_t)mant;
        row[1] = (uint32_t)(mant >> 32);
        for (int k = 2; k < nw; ++k) row[k] = 0;
    } else {
        // n - mant, mant < 2^53: 64-bit subtract, then a single borrow ripples through the upper words
        const uint64_t n_lo = (uint64_t)n_words_ptr[0] | ((uint64_t)n_words_ptr[1] << 32);
        const uint64_t lo = n_lo - mant;
        uint32_t borrow = mant > n_lo ? 1u : 0u;
        row[0] = (uint32_t)lo;
        row[1] = (uint32_t)(lo >> 32);
        for (int k = 2; k < nw; ++k) {
            const uint32_t w = n_words_ptr[k];
            row[k] = w - borrow;
            borrow = (borrow && w == 0) ? 1u : 0u;
        }
    }
}

// Encoding of an integer (fixedpoint.py:72-74,89-96): exponent 0, residue = x mod n.
__global__ void __launch_bounds__(256)
k_fp_encode_i64(const int64_t* __restrict__ x, const uint32_t* __restrict__ n_words_ptr, int nw, uint32_t* __restrict__ out,
                int32_t* __restrict__ expo, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int64_t v = x[i];
    // the reference's tiny-value test `np.abs(scalar) < 1e-200` (fixedpoint.py:64) wraps for -2^63 (int64 abs overflow:
    // the "absolute value" is negative), so that one value encodes as 0 there — reproduced for parity
    if (v == INT64_MIN) v = 0;
    const bool neg = v < 0;
    const uint64_t mant = neg ? (0ull - (uint64_t)v) : (uint64_t)v;
    expo[i] = 0;
    uint32_t* row = out + i * (size_t)nw;
    if (!neg) {
        row[0] = (uint32_t)mant;
        row[1] = (uint32_t)(mant >> 32);
        for (int k = 2; k < nw; ++k) row[k] = 0;
    } else {
        const uint64_t n_lo = (uint64_t)n_words_ptr[0] | ((uint64_t)n_words_ptr[1] << 32);
        const uint64_t lo = n_lo - mant;
        uint32_t borrow = mant > n_lo ? 1u : 0u;
        row[0] = (uint32_t)lo;
        row[1] = (uint32_t)(lo >> 32);
        for (int k = 2; k < nw; ++k) {
            const uint32_t w = n_words_ptr[k];
            row[k] = w - borrow;
            borrow = (borrow && w == 0) ? 1u : 0u;
        }
    }
}

// The two encoders with a per-element TARGET exponent (ct + plaintext, ipcl_python.py:495-504 + 570-741): the reference
// raw-encrypts the plaintext at its own exponent e and then raises that ciphertext to 2^(t - e) when the other operand's
// exponent t is larger.  For a raw encryption this is arithmetic on the plaintext: (1 + m n)^(2^d) = 1 + (m 2^d mod n) n
// (mod n^2), so encoding the plaintext at exponent t directly — mantissa shifted left by d — yields the SAME ciphertext
// bits and exponent with no squaring at all.  Done here when the shifted magnitude stays below n (|mant| 2^d < 2^(nbits-2));
// otherwise the element keeps its own exponent and the ciphertext path raises it as before.  A zero mantissa takes any
// larger target (1^anything = 1).  IS_F64: x holds doubles, else int64.
template <bool IS_F64>
__global__ void __launch_bounds__(256)
k_fp_encode_at(const void* __restrict__ x_, const uint32_t* __restrict__ n_words_ptr, int nw, int nbits,
               const int32_t* __restrict__ target, int t_bcast, uint32_t* __restrict__ out, int32_t* __restrict__ expo, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint64_t mant;
    bool neg;
    int e0, mbits;
    if constexpr (IS_F64) {
        const double v = reinterpret_cast<const double*>(x_)[i];
        const uint64_t bits = (uint64_t)__double_as_longlong(v);
        const bool tiny = fabs(v) < 1e-200;
        mant = tiny ? 0ull : ((bits & 0xFFFFFFFFFFFFFull) | (1ull << 52));
        neg = !tiny && (bits >> 63);
        e0 = tiny ? 0 : (1075 - (int)((bits >> 52) & 0x7FF));
        mbits = 53;
    } else {
        int64_t v = reinterpret_cast<const int64_t*>(x_)[i];
        if (v == INT64_MIN) v = 0;                          // see k_fp_encode_i64
        neg = v < 0;
        mant = neg ? (0ull - (uint64_t)v) : (uint64_t)v;
        e0 = 0;
        mbits = mant ? 64 - __clzll((long long)mant) : 0;        // the magnitude's own bit length (as fixedpoint.align_encoded)
    }
    int shift = 0;
    const int t = target[t_bcast ? 0 : i];
    const long long dist = (long long)t - (long long)e0;      // (64-bit: any int32 pair)
    if (dist > 0) {
        if (mant == 0) e0 = t;
        else if ((long long)mbits + dist <= (long long)nbits - 2) { shift = (int)dist; e0 = t; }
    }
    expo[i] = e0;
    const int ws = shift >> 5, bs = shift & 31;
    const uint64_t lo = mant << bs;
    const uint32_t hi = bs ? (uint32_t)(mant >> (64 - bs)) : 0u;
    uint32_t* row = out + i * (size_t)nw;
    uint32_t borrow = 0;
    for (int k = 0; k < nw; ++k) {
        const uint32_t vk = k == ws ? (uint32_t)lo : (k == ws + 1 ? (uint32_t)(lo >> 32) : (k == ws + 2 ? hi : 0u));
        if (!neg) {
            row[k] = vk;
        } else {
            const uint64_t d = (uint64_t)n_words_ptr[k] - vk - borrow;
            row[k] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
    }
}

// flag 0: mantissa in (-2^63, 2^63) returned; flag 1: anything else (|mantissa| >= 2^63, overflow zone, corrupt
// residue >= n) — the host then runs the exact big-integer path, which also raises the reference's exceptions.
__global__ void __launch_bounds__(256)
k_fp_decode_i64(const uint32_t* __restrict__ res, const uint32_t* __restrict__ n_words_ptr, int nw, int64_t* __restrict__ mant,
                int32_t* __restrict__ flag, size_t N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t* row = res + i * (size_t)nw;
    const uint64_t lo = (uint64_t)row[0] | ((uint64_t)row[1] << 32);
    uint32_t hi_or = 0;
    // d = n - row (multiword); the negative branch needs 0 < d < 2^63
    const uint64_t n_lo = (uint64_t)n_words_ptr[0] | ((uint64_t)n_words_ptr[1] << 32);
    const uint64_t d_lo = n_lo - lo;
    uint32_t borrow = lo > n_lo ? 1u : 0u;
    uint32_t d_hi_or = 0;
    for (int k = 2; k < nw; ++k) {
        const uint32_t r = row[k], w = n_words_ptr[k];
        hi_or |= r;
        const uint64_t t = (uint64_t)w - r - borrow;
        d_hi_or |= (uint32_t)t;
        borrow = (uint32_t)(t >> 63);       // 1 when the subtraction wrapped
    }
    const bool pos = hi_or == 0 && lo < (1ull << 63);
    const bool negv = !pos && borrow == 0 && d_hi_or == 0 && d_lo < (1ull << 63) && d_lo != 0;
    mant[i] = pos ? (int64_t)lo : (negv ? -(int64_t)d_lo : 0);
    flag[i] = (pos || negv) ? 0 : 1;
}

// ---- ChaCha20 block function, RFC 8439 section 2.3 -----------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define PAI_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);

struct ChaChaKey { uint32_t k[8]; uint32_t nonce[3]; uint32_t counter0; };

// word w of the key stream (block = w / 16) goes to out[w]; in every row of r_words words the word with index top_word
// is masked with top_mask and the words above it are zero (r < 2^rbits even when rbits is more than a word short of
// the row width).  The stream is the RFC's: state = constants | key | counter | nonce, 20 rounds, feed-forward.
__global__ void __launch_bounds__(256)
k_draw_r(ChaChaKey K, uint32_t* __restrict__ out, size_t total_words, int r_words, int top_word, uint32_t top_mask) {
    const size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 16 >= total_words) return;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, K.k[0], K.k[1], K.k[2], K.k[3], K.k[4], K.k[5], K.k[6], K.k[7],
                      K.counter0 + (uint32_t)blk, K.nonce[0] + (uint32_t)(blk >> 32), K.nonce[1], K.nonce[2]};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        PAI_CHACHA_QR(x[0], x[4], x[8], x[12]) PAI_CHACHA_QR(x[1], x[5], x[9], x[13])
        PAI_CHACHA_QR(x[2], x[6], x[10], x[14]) PAI_CHACHA_QR(x[3], x[7], x[11], x[15])
        PAI_CHACHA_QR(x[0], x[5], x[10], x[15]) PAI_CHACHA_QR(x[1], x[6], x[11], x[12])
        PAI_CHACHA_QR(x[2], x[7], x[8], x[13]) PAI_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const size_t w = blk * 16 + i;
        if (w < total_words) {
            uint32_t v = x[i] + s[i];
            const int col = (int)(w % (size_t)r_words);
            if (col == top_word) v &= top_mask;
            if (col > top_word) v = 0;
            out[w] = v;
        }
    }
}

// ---- exponent alignment through the digit engine: delta_i -> the exponent 2^max(delta_i, 0) as two words -----------
// (pai_ct_pow2 for large shifts: ct^(2^delta) is ct * pt with a one-bit exponent).  *dmax receives max(delta_i).
__global__ void __launch_bounds__(256)
k_pow2_expo(const int32_t* __restrict__ delta, int bcast, size_t n, uint32_t* __restrict__ e_out, int* __restrict__ dmax,
            int hint, int* __restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int d = 0;
    if (i < n) {
        d = delta[bcast ? 0 : i];
        d = d > 0 ? d : 0;
        const uint64_t e = d < 64 ? (1ull << d) : 0ull;
        e_out[2 * i] = (uint32_t)e;
        e_out[2 * i + 1] = (uint32_t)(e >> 32);
    }
    // wave maximum, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(d, off, 64);
        d = o > d ? o : d;
    }
    if ((threadIdx.x & 63) == 0 && d > 0) {
        atomicMax(dmax, d);
        // a caller's hint (pai_ct_pow2_hint) below a shift of the batch would truncate 2^delta on this path: remember it
        // in the handle's sticky status word (bit 1, pai_pubkey_status) instead of returning wrong ciphertexts silently
        if (hint >= 0 && d > hint) atomicOr(status, 2);
    }
}

}  // namespace pai
