// C ABI of libpaillier_hip.so (see include/paillier_hip.h): key set-up on the host, dispatch of the
// gfx950 kernels.  There is deliberately no CPU implementation of any hot operation in this file.
#include "../../include/paillier_hip.h"

#include <hip/hip_runtime.h>
#include <string.h>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "geo_ops.hpp"
#include "hostbn.hpp"
#include "host_keygen.hpp"
#include "kernels_padic.hpp"
#include "kernels_padic_enc.hpp"
#include "kernels_pair.hpp"
#include "kernels_codec.hpp"
#include "kernels_declat.hpp"

using namespace pai;
using hbn::Limbs;

namespace {

thread_local std::string g_err;

struct PaiError : std::runtime_error {
    int code;
    PaiError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(x)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess)                                                                         \
            throw PaiError(PAI_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                \
    } while (0)

template <class F>
int guarded(F&& f) {
    try {
        f();
        return PAI_OK;
    } catch (const PaiError& e) {
        g_err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_err = e.what();
        return PAI_E_INTERNAL;
    } catch (...) {
        g_err = "unknown error";
        return PAI_E_INTERNAL;
    }
}

__global__ void k_status_or(int* status, const int* flag, int bit) {
    if (*flag) atomicOr(status, bit);
}

void require(bool ok, const char* msg) {
    if (!ok) throw PaiError(PAI_E_INVALID, msg);
}

int words_for_bits(int bits) { return (bits + 31) / 32; }

struct DeviceInfo {
    int ncu = 0;
};

// Properties of every visible device, queried once (hipGetDeviceProperties costs ~ms and the hot calls used to
// pay it every time).
const std::vector<DeviceInfo>& device_table() {
    static const std::vector<DeviceInfo> tbl = [] {
        std::vector<DeviceInfo> t;
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess) cnt = 0;
        for (int d = 0; d < cnt; ++d) {
            hipDeviceProp_t p;
            DeviceInfo di;
            if (hipGetDeviceProperties(&p, d) == hipSuccess) di.ncu = p.multiProcessorCount;
            t.push_back(di);
        }
        return t;
    }();
    return tbl;
}

// Makes `device` current for the calling thread for the lifetime of the object and restores the previous device
// afterwards: in a single-process multi-GPU program the caller's current device (which torch reads through
// hipGetDevice) must not change behind its back.
struct DeviceScope {
    int prev = -1;
    DeviceInfo info;
    explicit DeviceScope(int device) {
        const auto& tbl = device_table();
        if (tbl.empty()) throw PaiError(PAI_E_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
        if (device < 0 || device >= (int)tbl.size()) throw PaiError(PAI_E_INVALID, "device index out of range");
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device && hipSetDevice(device) != hipSuccess) throw PaiError(PAI_E_NODEVICE, "hipSetDevice failed");
        info = tbl[device];
        if (info.ncu <= 0) throw PaiError(PAI_E_NODEVICE, "hipGetDeviceProperties failed");
    }
    ~DeviceScope() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// Orders the operations of one key handle that share its device scratch (window tables, quotient-digit columns)
// when callers issue them on different streams: the next user on another stream waits for the event the previous
// user recorded.  Same-stream users are ordered by the stream itself.  Call begin() and end() under the handle's mutex.
struct ScratchOrder {
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool armed = false;
    void begin(hipStream_t s) {
        if (armed && s != last) HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
    }
    void end(hipStream_t s) {
        if (!ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev, s));
        last = s;
        armed = true;
    }
    void release() {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        armed = false;
    }
};

// begin() now, end() when the scope is left — also by an exception, so that work already queued on the stream stays ordered
// before the next user of the scratch.  done() ends early (before a host synchronisation).
struct OrderScope {
    ScratchOrder& o;
    hipStream_t s;
    bool open = true;
    OrderScope(ScratchOrder& o_, hipStream_t s_) : o(o_), s(s_) { o.begin(s); }
    void done() {
        if (open) { open = false; o.end(s); }
    }
    ~OrderScope() {
        if (!open) return;
        try { o.end(s); } catch (...) {}
    }
    OrderScope(const OrderScope&) = delete;
    OrderScope& operator=(const OrderScope&) = delete;
};

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void ensure(size_t need) {
        if (need <= bytes) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr;
        bytes = 0;
        HIP_CHECK(hipMalloc(&p, need));
        bytes = need;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// a DevBuf that frees itself when its scope ends, exceptions included (temporaries of the table builders)
struct ScopedDevBuf : DevBuf {
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf&) = delete;
    ScopedDevBuf& operator=(const ScopedDevBuf&) = delete;
    ~ScopedDevBuf() { release(); }
};

// NLMAX-padded radix-29 constant on the device
uint32_t* upload_r29(const Limbs& v, int nl) {
    std::vector<uint32_t> h(NLMAX, 0);
    auto r = hbn::to_r29(v, nl);
    std::memcpy(h.data(), r.data(), (size_t)nl * 4);
    uint32_t* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, NLMAX * 4));
    HIP_CHECK(hipMemcpy(d, h.data(), NLMAX * 4, hipMemcpyHostToDevice));
    return d;
}
uint32_t* upload_words(const Limbs& v, int words) {
    std::vector<uint32_t> h(words, 0);
    for (size_t i = 0; i < v.size() && i < (size_t)words; ++i) h[i] = v[i];
    uint32_t* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, (size_t)words * 4));
    HIP_CHECK(hipMemcpy(d, h.data(), (size_t)words * 4, hipMemcpyHostToDevice));
    return d;
}

// One modulus prepared for a geometry: host values + device MontCtx
struct ModSetup {
    Limbs M, R, R2, R3;
    const GeoOps* geo = nullptr;
    MontCtx* d_ctx = nullptr;
    int bits = 0, w32 = 0;
    int nl = 0;       // radix-29 limbs of the Montgomery representation (R = 2^(29 nl))
    int m1_rows = 0;  // minus-one contexts: rows per product, R = 2^(29 m1_rows)
    int rows() const { return m1_rows ? m1_rows : nl; }
    // nl_override != 0: constants for the wide engine's limb count instead of the lane-group geometry's
    // r2_override: the constant MODMUL_FULL multiplies by instead of R^2 (small-batch tagged products: pai_ct_mont_mul)
    void init(const Limbs& mod_, int nl_override = 0, const GeoOps* force_geo = nullptr, const Limbs* r2_override = nullptr) {
        M = mod_;
        require(hbn::is_odd(M), "modulus must be odd");
        bits = hbn::bitlen(M);
        w32 = words_for_bits(bits);
        geo = force_geo ? force_geo : geo_for_bits(bits);
        if (!geo) throw PaiError(PAI_E_UNSUPPORTED, "modulus wider than 8192 bits is not supported");
        nl = nl_override ? nl_override : geo->nl;
        require(nl <= NLMAX, "geometry wider than the constant tables");
        R = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * nl), M);
        R2 = hbn::mulmod(R, R, M);
        R3 = hbn::mulmod(R2, R, M);
        MontCtx h;
        std::memset(&h, 0, sizeof(h));
        auto put = [&](uint32_t* dst, const Limbs& v) {
            auto r = hbn::to_r29(v, nl);
            std::memcpy(dst, r.data(), (size_t)nl * 4);
        };
        put(h.n, M);
        put(h.r2, r2_override ? *r2_override : R2);
        put(h.one, R);
        h.n0inv = hbn::neg_inv32(M[0]) & ((1u << hbn::RB) - 1u);
        h.nl = (uint32_t)nl;
        h.bits = (uint32_t)bits;
        if (geo->u + 3 <= 12) {
            // Barrett constant of the minus-one geometries' way out (kernels_common.hpp: m1_reduce_to_true_modulus)
            const int m = (bits + hbn::RB - 1) / hbn::RB;
            h.mlimbs = (uint32_t)m;
            const Limbs mu = hbn::divq(hbn::shl(Limbs{1u}, hbn::RB * (m + geo->u + 1)), M, nullptr);
            auto r = hbn::to_r29(mu, 12);
            std::memcpy(h.mu, r.data(), sizeof(h.mu));
        }
        HIP_CHECK(hipMalloc((void**)&d_ctx, sizeof(MontCtx)));
        HIP_CHECK(hipMemcpy(d_ctx, &h, sizeof(MontCtx), hipMemcpyHostToDevice));
    }
    // "Minus-one" context of the wide-group (latency) kernels (mont_dev.hpp: Rows::block_m1): the modulus is replaced by
    // M' = M k, k = -M^-1 mod 2^(29 U), so that M' == -1 (mod 2^(29 U)) and the quotient digits of a row block are the
    // limbs the block retires.  Residues modulo M' are residues modulo M; R = 2^(29 rows) with rows = the limbs M' needs
    // (+ 4 bits of head-room: R > 16 M'), not the geometry's capacity.  M, R, R2, R3 of this object then refer to M'.
    void init_m1(const Limbs& mod_, const GeoOps* g, int headroom_bits = 4, int row_multiple = 0) {
        require(hbn::is_odd(mod_), "modulus must be odd");
        geo = g;
        nl = g->nl;
        const int ub = hbn::RB * g->u;
        const Limbs pow = hbn::shl(Limbs{1u}, ub);
        const Limbs k = hbn::sub(pow, hbn::inv_mod_pow2(mod_, ub));           // -M^-1 mod 2^(29 U)
        M = hbn::mul(mod_, k);
        bits = hbn::bitlen(M);
        w32 = words_for_bits(bits);
        int rows = (bits + headroom_bits + hbn::RB - 1) / hbn::RB;
        const int rmul = row_multiple ? row_multiple : g->u;
        rows = (rows + rmul - 1) / rmul * rmul;
        require(rows <= nl, "minus-one modulus does not fit the geometry");
        const Limbs mp1 = hbn::add(M, Limbs{1u});
        require(hbn::is_zero(hbn::low_bits(mp1, ub)), "minus-one modulus: construction failed");
        const Limbs npp = hbn::shr(mp1, ub);
        R = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * rows), M);
        R2 = hbn::mulmod(R, R, M);
        R3 = hbn::mulmod(R2, R, M);
        MontCtx h;
        std::memset(&h, 0, sizeof(h));
        auto put = [&](uint32_t* dst, const Limbs& v) {
            auto r = hbn::to_r29(v, nl);
            std::memcpy(dst, r.data(), (size_t)nl * 4);
        };
        put(h.n, M);
        put(h.npp, npp);
        put(h.r2, R2);
        put(h.one, R);
        h.n0inv = 1;
        h.nl = (uint32_t)nl;
        h.bits = (uint32_t)bits;
        h.rows = (uint32_t)rows;
        m1_rows = rows;
        HIP_CHECK(hipMalloc((void**)&d_ctx, sizeof(MontCtx)));
        HIP_CHECK(hipMemcpy(d_ctx, &h, sizeof(MontCtx), hipMemcpyHostToDevice));
    }
    void release() {
        if (d_ctx) (void)hipFree(d_ctx);
        d_ctx = nullptr;
    }
};

// Optional per-kernel timing with HIP events on the caller's stream (pai_profile_enable): used by
// bench.py to report the dominant kernel's duration next to the rocprofv3 numbers.
bool g_profile = false;
struct KernelTime { std::string name; float ms; };
thread_local std::vector<KernelTime> g_last_times;
struct ScopedKernelTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s;
    const char* name;
    bool on;
    ScopedKernelTimer(const char* n, hipStream_t st) : s(st), name(n), on(g_profile) {
        if (!on) return;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, s));
    }
    void stop() {
        if (!on || !e0) return;
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        g_last_times.push_back({name, ms});
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        e0 = e1 = nullptr;
    }
    ~ScopedKernelTimer() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// ---- run-time knobs ----------------------------------------------------------------------
// Sizing knobs have their own variables (PAI_FB_CACHE_MB, PAI_FB_TABLE_MB, PAI_FB_BIG_KEYS, PAI_FB_SMALL_TABLE_MB,
// PAI_LATENCY_MAX, PAI_LAT_ADD_MAX, PAI_POW2_DIGIT_MIN).  Everything the tests and probes use to steer a call onto a
// particular path lives in two lists, read at every use (tests change them between calls):
//   PAI_DISABLE="padic,pair,..."   engines / forms to leave out: padic (digit-pair engines: lane-group and wide fallbacks
//                                  serve), pair, pair_ctmul, wide, gform (plain fixed-base tables), fb_chain, lat_dense, lat_enc_m1
//   PAI_TUNE="name=value,..."      fb_wbits, fb_digit_wbits, lat_fb_wbits, fb_gform_k, invert_chunk, mexp_wbits, mexp_lanes,
//                                  mexp_by_rows, lat_rl, lat_mul_rl, lat_enc_tree (largest batch of that small-batch form, 0 = off)
static const char* list_find(const char* list, const char* name) {       // -> the character behind `name` in the list, or NULL
    if (!list) return nullptr;
    const size_t n = std::strlen(name);
    for (const char* p = list; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        const char* e = p;
        while (*e && *e != ',') ++e;
        if ((size_t)(e - p) >= n && std::strncmp(p, name, n) == 0 && (p[n] == '=' || p[n] == ',' || p[n] == 0 || p[n] == ' ')) return p + n;
        p = e;
    }
    return nullptr;
}
static bool knob_disabled(const char* name) { return list_find(std::getenv("PAI_DISABLE"), name) != nullptr; }
static bool knob_tune(const char* name, long long* v) {
    const char* q = list_find(std::getenv("PAI_TUNE"), name);
    if (!q || *q != '=') return false;
    *v = std::strtoll(q + 1, nullptr, 10);
    return true;
}

// Batches up to this many elements take the latency paths: every integer spread over 16-64 lanes, (element, prime) pairs
// filling the device instead of lanes.  The switch to the throughput kernels depends on the operation and on the key size —
// measured cross-overs on MI355X (profiles/r05/path_switch_sweep.jsonl: tools/latency_sweep.py, default against both forced
// paths at 1024 / 2048 / 3072 / 4096-bit keys).  PAI_LATENCY_MAX overrides (tests: 0 disables the latency paths, a huge value
// forces them): decryption and DJN encryption switch at twice, ct * pt at four times its value.
enum LatOp { LAT_DEC, LAT_ENC, LAT_MUL };
size_t latency_max_elements(LatOp op, int key_bits) {
    if (const char* env = std::getenv("PAI_LATENCY_MAX")) return (size_t)std::strtoull(env, nullptr, 10) * (op == LAT_MUL ? 4 : 2);
    static const struct { int bits; size_t v[3]; } T[] = {
        {1024, {6400, 8704, 13312}},          // decrypt 3.4 ms at 6144 against 3.6; encrypt 0.45 at 8192 against 0.48; ct * pt 1.13 at 12288 against 1.23
        {2048, {5120, 11776, 18944}},         // 12.1 ms at 4096 against 14.8; 3.4 at 12288 against 3.3; 4.0 at 16384 against 4.6
        {3072, {6144, 4864, 5632}},           // 55.9 at 6144 against 55.9; 3.3 at 4096 against 4.0; 2.1 at 4096 against 3.0 (3.1 at 6144)
        {4096, {7936, 3200, 2944}},           // 98 at 6144 against 127; 4.6 at 3072 against 4.9; 1.8 at 2048 against 2.7 (2.8 at 3072)
    };
    for (const auto& t : T) if (key_bits <= t.bits) return t.v[op];
    return T[3].v[op];
}

// pai_ct_pow2 goes through the digit engine from this batch size on (PAI_POW2_DIGIT_MIN) when the largest shift is
// at least POW2_DIGIT_MIN_SHIFT
constexpr int POW2_DIGIT_MIN_SHIFT = 8;
size_t pow2_digit_min_elements() {
    if (const char* env = std::getenv("PAI_POW2_DIGIT_MIN")) return (size_t)std::strtoull(env, nullptr, 10);
    return (size_t)16384;
}

int grid_for(const GeoOps* g, size_t N, int ncu, int blocks_per_cu = 2) {
    size_t tiles = (N + g->epb - 1) / g->epb;
    size_t cap = (size_t)ncu * blocks_per_cu;
    return (int)std::max<size_t>(1, std::min(tiles, cap));
}

}  // namespace

namespace pai {
const GeoOps* geo_for_bits(int bits) {
    const GeoOps* all[] = {geo_ops_36x1(), geo_ops_36x2(), geo_ops_28x4(), geo_ops_36x4(), geo_ops_28x8(), geo_ops_36x8()};
    for (const GeoOps* g : all)
        if (hbn::RB * g->nl >= bits + 2) return g;
    return nullptr;
}
const GeoOps* geo_latency_for_bits(int bits) {
    const GeoOps* all[] = {geo_ops_3x16(), geo_ops_3x32(), geo_ops_3x64(), geo_ops_9x32()};
    for (const GeoOps* g : all)
        if (hbn::RB * g->nl >= bits + 2) return g;
    return nullptr;
}
}  // namespace pai

// ------------------------------------------------------------------------------------------------
// Pinned host staging for small host operands of asynchronous calls (pai_modexp_fixed's exponent): a ring of slots, each
// reusable once the copy that read it has executed (event); the caller's pageable buffer is free on return.
struct PinnedRing {
    static constexpr int SLOTS = 4;
    void* p[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t bytes[SLOTS] = {0, 0, 0, 0};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int next = 0;
    // copies `n` bytes of h_src to d_dst on stream s through the next slot
    void h2d(void* d_dst, const void* h_src, size_t n, hipStream_t s) {
        const int k = next;
        next = (next + 1) % SLOTS;
        if (!ev[k]) HIP_CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        else HIP_CHECK(hipEventSynchronize(ev[k]));                 // the slot's previous copy has run (normally long ago)
        if (bytes[k] < n) {
            if (p[k]) (void)hipHostFree(p[k]);
            p[k] = nullptr;
            bytes[k] = 0;
            HIP_CHECK(hipHostMalloc(&p[k], n, hipHostMallocDefault));
            bytes[k] = n;
        }
        std::memcpy(p[k], h_src, n);
        HIP_CHECK(hipMemcpyAsync(d_dst, p[k], n, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipEventRecord(ev[k], s));
    }
    void release() {
        for (int k = 0; k < SLOTS; ++k) {
            if (ev[k]) { (void)hipEventSynchronize(ev[k]); (void)hipEventDestroy(ev[k]); ev[k] = nullptr; }
            if (p[k]) { (void)hipHostFree(p[k]); p[k] = nullptr; bytes[k] = 0; }
        }
    }
};

struct pai_modulus {
    int device = 0;
    DeviceInfo dev;
    ModSetup ms;
    DevBuf table, expo;
    PinnedRing pinned;
    ScratchOrder order;
    std::mutex mu;
};

struct pai_pubkey {
    int device = 0;
    DeviceInfo dev;
    int key_bits = 0, n_words = 0, ct_words = 0, r_words = 0, randbits = 0;
    bool djn = false;
    Limbs n, nsq, hs;
    ModSetup msq;                 // n^2
    uint32_t* d_nR = nullptr;     // n * R mod n^2 (radix 29)
    uint32_t* d_fb = nullptr;     // fixed-base table [J][256][NL]
    uint32_t* d_nexp = nullptr;   // n as packed words (exponent of the standard obfuscator)
    int fb_windows = 0, fb_wbits = 8;
    int fbd_windows = 0, fbd_wbits = 8;   // window geometry of the digit-form table (may be wider: see pai_pubkey_create)
    bool fb_gform = false;                // the digit-form table holds g-factored entries (a, t): kernels_padic_enc.hpp
    // digit engine with base n (raw / DJN encryption): modulus n, n - 1, n^2 limbs, digit-form table, scratch
    int penc_nl = 0;
    ModSetup nmod;
    uint32_t* d_nm1 = nullptr;
    uint32_t* d_nsq29 = nullptr;
    uint32_t* d_fb_dig = nullptr;
    uint32_t* d_mscratch = nullptr;
    uint32_t* d_one_dig = nullptr;     // digit pair of R mod n^2 (the element 1 in Montgomery digit form)
    uint32_t* d_ct_kdig = nullptr;     // [ct_nd][2][NL] digit pairs of R^(i+2) mod n^2
    int ct_nd = 0;
    // digit pairs with base n on the lane-group engine (kernels_pair.hpp): DJN obfuscator for n beyond the digit engine
    int pair_nl = 0;
    ModSetup npair;                    // n at pair_nl limbs
    uint32_t* d_pair_nm1 = nullptr;
    uint32_t* d_pair_fb = nullptr;     // [J][2^wb][2][pair_nl]
    uint32_t* d_pair_kdig = nullptr;   // [pair_nd][2][pair_nl] pairs of R^(i+2) mod n^2 (ct * pt on lane-group digit pairs)
    uint32_t* d_pair_one = nullptr;    // pair(R mod n^2)
    int pair_nd = 0;
    mutable DevBuf pair_ct_table;      // per-slot power tables of k_pair_ctmul
    // mid-size ct * pt at keys the one-element-per-lane engine serves (<= 2048 bits): the same lane-group digit-pair exponentiation on
    // 4 lanes x 18 limbs (16 ciphertexts per wavefront); constants built by the first such call
    mutable bool midp_tried = false, midp_ok = false;
    mutable ModSetup midp_n;
    mutable uint32_t* d_midp_nm1 = nullptr;
    mutable uint32_t* d_midp_kdig = nullptr;
    mutable uint32_t* d_midp_one = nullptr;
    mutable int midp_nl = 0, midp_nd = 0, midp_out_words = 0;
    int pair_windows = 0, pair_wbits = 0, pair_out_words = 0;
    mutable DevBuf pair_wv;            // plain digit pairs on their way to k_encrypt mode 5 / 6
    uint16_t* d_pow_ops = nullptr;     // sliding-window schedule of the exponent n (standard scheme)
    int pow_nops = 0;
    mutable DevBuf ctmul_table;        // per-slot window tables of k_ctmul_padic
    mutable DevBuf pow2_expo;          // one-bit exponents of pai_ct_pow2's digit-engine path
    mutable DevBuf mexp_table, mexp_partial;   // power tables and partial products of pai_ct_multiexp
    uint32_t* d_nsq_words = nullptr;   // n^2 as packed words (extended-GCD modulus)
    mutable DevBuf table, tmp;    // standard-scheme scratch
    mutable DevBuf inv_prod, inv_inv, inv_fail;
    // sticky device status word of the asynchronous calls (pai_pubkey_status): bit 0 = pai_ct_invert_async met a
    // ciphertext that is not a unit, bit 1 = a pai_ct_pow2_hint hint was smaller than a shift of its batch
    mutable DevBuf status;
    mutable DevBuf prod_a, prod_b;     // ping-pong levels of pai_ct_prod
    // Product trees run on single Montgomery products (k_modmul MODMUL_MONT); level k of a tree holds true values
    // times R^(1 - 2^k).  tree_c[k] = R^(1 - 2^k) mod n^2 brings a node without a partner to its level's form,
    // tree_fix[L] = R^(2^L) mod n^2 returns the root of an L-level tree to a plain residue (packed rows of ct_words).
    static constexpr int TREE_LEVELS = 40;
    uint32_t* d_tree_c = nullptr;
    uint32_t* d_tree_fix = nullptr;
    mutable uint32_t* d_rpow = nullptr;  // R^m mod n^2 for |m| <= RPOW_SPAN in limb form (k_addn), built by the first pai_ct_addn
    mutable bool fb_ready = false;     // fixed-base tables are built by the first obfuscating call (build_fb_tables)
    mutable size_t fb_bytes = 0;       // device bytes of the built tables (the per-device table cache accounts with it)
    mutable size_t fb_table_budget = 0;  // non-zero: this build takes the small operating point (build_fb_tables)
    // latency path of ct * pt (small batches): n^2 on a wide-group geometry, built by the first small call
    mutable bool lat_ready = false, lat_usable = false;
    mutable ModSetup lat_msq;
    mutable ModSetup lat_msq_m1;       // minus-one context of n^2 for the small-batch ct * pt
    mutable bool lat_m1_tried = false, lat_m1_ok = false;
    // smallest ct * pt batches: the four-wave digit-pair pipeline with base n' = n k (k_ctmul_pp; 3 x 64 geometry only)
    mutable bool lat_pp_tried = false, lat_pp_ok = false;
    mutable ModSetup lat_pp;
    mutable uint32_t* d_lat_pp_kdig = nullptr;
    mutable uint32_t* d_lat_pp_kx = nullptr;
    mutable int lat_pp_nd = 0, lat_pp_nch = 0;
    mutable int lat_pp_chain = 1;
    // small-batch ct + ct: n^2 on the latency geometry; the "tag" context multiplies by R_lat^2 / R instead of R_lat^2, so that
    // MODMUL_FULL there returns a b R^-1 in terms of the throughput geometry's R (the lazy domain tags of the containers)
    mutable ModSetup lat_msq_tag;
    mutable bool lat_tag_tried = false, lat_tag_ok = false;
    mutable DevBuf lat_table;
    // latency path of DJN encryption: n R and a 10-bit fixed-base table in the wide-group geometry (81 MB at 2048-bit keys)
    mutable bool lat_fb_ready = false;
    mutable uint32_t* d_lat_fb_m1 = nullptr;          // the same table in the Montgomery form of lat_msq_m1 (k_encrypt_tree on a minus-one context)
    mutable uint32_t* d_lat_nR_m1 = nullptr;
    mutable uint32_t* d_lat_nR = nullptr;
    mutable uint32_t* d_lat_fb = nullptr;
    mutable int lat_fb_windows = 0, lat_fb_wbits = 10;
    mutable ScratchOrder order;
    mutable std::mutex mu;
    EncParams enc_params() const {
        EncParams P;
        P.nsq = msq.d_ctx;
        P.nR = d_nR;
        P.fb_table = d_fb;
        P.fb_windows = fb_windows;
        P.fb_wbits = fb_wbits;
        P.pt_words = n_words;
        P.ct_words = ct_words;
        P.r_words = r_words;
        return P;
    }
};

struct pai_privkey {
    const pai_pubkey* pk = nullptr;
    Limbs p, q;
    ModSetup sq[2];               // p^2, q^2
    ModSetup pr[2];               // p, q
    ModSetup pdig[2];             // p, q at the digit engine's limb count (stage A on digit pairs)
    uint32_t* d_r3[2] = {nullptr, nullptr};
    uint32_t* d_expo[2] = {nullptr, nullptr};
    int ewords[2] = {0, 0}, ebits[2] = {0, 0};
    uint32_t* d_sinv2[2] = {nullptr, nullptr};
    uint32_t* d_nsinv2[2] = {nullptr, nullptr};
    uint32_t* d_hR[2] = {nullptr, nullptr};
    uint32_t* d_pinvqR = nullptr;
    DevBuf wscratch;
    int u_words = 0;
    int wide_nl = 0;              // != 0: stage A runs on the wide engine with this many limbs
    int padic_nl = 0;             // != 0: stage A runs on the p-adic digit engine (takes precedence)
    uint32_t* d_pm1[2] = {nullptr, nullptr};
    uint32_t* d_kdig[2] = {nullptr, nullptr};
    uint16_t* d_ops[2] = {nullptr, nullptr};
    int nops[2] = {0, 0};
    int padic_nd = 0;
    DevBuf table, ubuf;
    // Mid-size batches: stage A as the lane-group digit-pair exponentiation (k_pair_ctmul: modulus s, exponent s - 1, 4 lanes x 9
    // limbs per chain, both primes in one launch), then w + v s on the s^2 geometry (k_pair_finish); built by the first such call
    struct Mid {
        bool tried = false, ok = false;
        int nl = 0, nd = 0, out_words = 0;
        ModSetup sp[2];                   // s on the pair geometry
        ModSetup s2[2];                   // s^2 on its own lane-group geometry (k_pair_finish)
        uint32_t* d_nm1[2] = {nullptr, nullptr};
        uint32_t* d_kdig[2] = {nullptr, nullptr};
        uint32_t* d_one[2] = {nullptr, nullptr};
        uint32_t* d_sR[2] = {nullptr, nullptr};
        DevBuf table[2], wv[2];
    } mid;
    // Latency path (small batches): stage A and B on the wide-group geometries (an integer spread over 16 / 32 / 64
    // lanes), with their own Montgomery constants; built by the first small call (build_latency_consts).
    Limbs h_host[2], pinvq_host;
    struct Lat {
        bool ready = false, usable = false;
        ModSetup sq[2], pr[2];
        ModSetup sq_true[2];          // s^2 itself (sq[] are minus-one contexts of s^2 k): the last reduction of stage A
        uint16_t* d_ops[2] = {nullptr, nullptr};     // sliding-window schedule of s - 1
        int nops[2] = {0, 0};
        uint32_t* d_r3[2] = {nullptr, nullptr};
        uint32_t* d_sinv2[2] = {nullptr, nullptr};
        uint32_t* d_nsinv2[2] = {nullptr, nullptr};
        uint32_t* d_hR[2] = {nullptr, nullptr};
        uint32_t* d_pinvqR = nullptr;
        DevBuf table;
        // mid-size batches (more than one wave per SIMD at one integer per wavefront): the densest wide-group geometry
        // s^2 k fits (two integers per wavefront at 2048-bit keys) — the same stage A on its own contexts
        bool dense = false;
        ModSetup sq2[2], sq2_true[2];
        uint32_t* d_r3_2[2] = {nullptr, nullptr};
        // smallest batches: stage A on digit pairs with base s' = s k, four waves per (ciphertext, prime) (kernels_declat.hpp)
        bool pp_ok = false;
        ModSetup pp[2];
        uint32_t* d_pp_kdig[2] = {nullptr, nullptr};
        uint32_t* d_pp_kx[2] = {nullptr, nullptr};
        int pp_nd = 0, pp_nch = 0;
        int pp_chain = 1;                 // limbs per lane of the chain's contexts (both primes)
    } lat;
    ScratchOrder order;
    std::mutex mu;
};

struct PubkeyDeleter { void operator()(pai_pubkey* p) const; };
struct PrivkeyDeleter { void operator()(pai_privkey* p) const; };
struct ModulusDeleter { void operator()(pai_modulus* p) const; };

extern "C" {

int pai_version(void) { return 200; }

const char* pai_last_error(void) { return g_err.c_str(); }

int pai_keygen(int key_bits, int djn, const uint64_t* h_seed, uint32_t* h_p, uint32_t* h_q) {
    return guarded([&] {
        require(key_bits >= 128 && key_bits <= 8192 && key_bits % 64 == 0, "pai_keygen: key_bits must be a multiple of 64 in 128..8192");
        require(h_p != nullptr && h_q != nullptr, "pai_keygen: output pointer is NULL");
        const int half = key_bits / 2, L = (half + 63) / 64, words = half / 32;
        uint64_t p[kg::MAXL] = {0}, q[kg::MAXL] = {0};
        kg::generate_primes(key_bits, djn != 0, h_seed, p, q);
        std::memcpy(h_p, p, 4 * (size_t)words);
        std::memcpy(h_q, q, 4 * (size_t)words);
        explicit_bzero(p, sizeof(p));                                 // the primes do not stay on this thread's stack
        explicit_bzero(q, sizeof(q));
        (void)L;
    });
}

int pai_host_modexp(const uint32_t* h_base, const uint32_t* h_exp, int exp_words, const uint32_t* h_mod, int mod_words,
                    uint32_t* h_out) {
    return guarded([&] {
        require(h_base && h_exp && h_mod && h_out, "pai_host_modexp: NULL pointer");
        require(mod_words >= 1 && mod_words <= 2 * kg::MAXL && exp_words >= 1, "pai_host_modexp: bad word count");
        require((h_mod[0] & 1u) != 0, "pai_host_modexp: modulus must be odd");
        const int L = (mod_words + 1) / 2, eL = (exp_words + 1) / 2;
        uint64_t m[kg::MAXL] = {0}, b[kg::MAXL] = {0}, out[kg::MAXL] = {0};
        std::vector<uint64_t> e((size_t)eL, 0);
        std::memcpy(m, h_mod, 4 * (size_t)mod_words);
        std::memcpy(b, h_base, 4 * (size_t)mod_words);
        std::memcpy(e.data(), h_exp, 4 * (size_t)exp_words);
        int Lt = L;
        while (Lt > 1 && m[Lt - 1] == 0) --Lt;                       // Montgomery radix from the modulus' own length
        for (int i = Lt; i < L; ++i) require(b[i] == 0, "pai_host_modexp: base wider than the modulus");
        kg::Mont mt(m, Lt);
        require(kg::cmp(b, mt.m, Lt) < 0, "pai_host_modexp: base must be reduced modulo the modulus");
        mt.pow(out, b, e.data(), eL);
        std::memcpy(h_out, out, 4 * (size_t)mod_words);
    });
}

int pai_device_count(int* count) {
    return guarded([&] {
        require(count != nullptr, "count is NULL");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *count = c;
    });
}

int pai_profile_enable(int on) {
    g_profile = on != 0;
    return PAI_OK;
}
int pai_profile_last(int index, char* name_out, size_t name_cap, float* ms_out) {
    if (index < 0 || (size_t)index >= g_last_times.size()) return PAI_E_INVALID;
    if (name_out && name_cap) {
        std::strncpy(name_out, g_last_times[index].name.c_str(), name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    if (ms_out) *ms_out = g_last_times[index].ms;
    return PAI_OK;
}

int pai_malloc(int device, size_t bytes, void** d_ptr) {
    return guarded([&] {
        require(d_ptr != nullptr, "d_ptr is NULL");
        DeviceScope scope_(device);
        HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 4));
    });
}
int pai_free(int device, void* d_ptr) {
    return guarded([&] {
        DeviceScope scope_(device);
        if (d_ptr) HIP_CHECK(hipFree(d_ptr));
    });
}
int pai_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes, void* stream) {
    return guarded([&] {
        DeviceScope scope_(device);
        HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    });
}
int pai_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes, void* stream) {
    return guarded([&] {
        DeviceScope scope_(device);
        HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    });
}
// Container operations of ipclPlainText / ipclCipherText on device rows (bindings/ipcl_bindings_classes.cpp:224-262,
// 328-366: __getitem__ with an index or a slice, rotate): pure copies, no arithmetic.
int pai_buf_slice(int device, const uint32_t* d_src, int row_words, size_t start, size_t count, size_t step, uint32_t* d_out,
                  void* stream) {
    return guarded([&] {
        require(d_src && d_out && row_words > 0 && step >= 1, "bad arguments");
        if (count == 0) return;
        DeviceScope scope_(device);
        const size_t row_bytes = (size_t)row_words * 4;
        const uint32_t* src = d_src + start * (size_t)row_words;
        if (step == 1) {
            HIP_CHECK(hipMemcpyAsync(d_out, src, count * row_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        } else {
            HIP_CHECK(hipMemcpy2DAsync(d_out, row_bytes, src, step * row_bytes, row_bytes, count, hipMemcpyDeviceToDevice,
                                       (hipStream_t)stream));
        }
    });
}
// out[i] = src[(i + shift) mod N]   (std::rotate to the left by `shift`, as CipherText::rotate)
int pai_buf_rotate(int device, const uint32_t* d_src, int row_words, size_t N, long long shift, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(d_src && d_out && row_words > 0 && d_src != d_out, "bad arguments");
        if (N == 0) return;
        DeviceScope scope_(device);
        const long long n = (long long)N;
        const size_t k = (size_t)(((shift % n) + n) % n);
        const size_t row_bytes = (size_t)row_words * 4;
        HIP_CHECK(hipMemcpyAsync(d_out, d_src + k * (size_t)row_words, (N - k) * row_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (k)
            HIP_CHECK(hipMemcpyAsync(d_out + (N - k) * (size_t)row_words, d_src, k * row_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    });
}
int pai_stream_sync(int device, void* stream) {
    return guarded([&] {
        DeviceScope scope_(device);
        HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    });
}

// ---- small host operands read by kernels straight from pinned memory (include/paillier_hip.h: pai_host_stage) ------------
}  // extern "C"
namespace {
struct HostStageRing {
    static constexpr int SLOTS = 32;
    char* base = nullptr;
    hipEvent_t ev[SLOTS] = {};
    bool recorded[SLOTS] = {};
    int next = 0, pending = -1;
    hipStream_t pending_stream = nullptr;
    ~HostStageRing() {
        // thread exit: the ring's last readers were enqueued long ago; errors of a runtime that is already shutting down are ignored
        for (int k = 0; k < SLOTS; ++k) if (ev[k]) { (void)hipEventSynchronize(ev[k]); (void)hipEventDestroy(ev[k]); }
        if (base) (void)hipHostFree(base);
    }
};
struct HostStageRings { std::vector<std::unique_ptr<HostStageRing>> per_device; };
thread_local HostStageRings tl_stage;
}  // namespace
extern "C" {

int pai_host_stage(int device, int parts, const void* const* h_src, const size_t* bytes, void* stream, void** d_ptrs) {
    return guarded([&] {
        require(parts >= 1 && parts <= 8 && h_src && bytes && d_ptrs, "bad arguments");
        size_t total = 0;
        for (int i = 0; i < parts; ++i) { require(h_src[i] != nullptr || bytes[i] == 0, "NULL part"); total += (bytes[i] + 15) & ~(size_t)15; }
        require(total <= PAI_HOST_STAGE_MAX, "pai_host_stage: more than PAI_HOST_STAGE_MAX bytes");
        DeviceScope scope_(device);
        if ((int)tl_stage.per_device.size() <= device) tl_stage.per_device.resize((size_t)device + 1);
        if (!tl_stage.per_device[device]) tl_stage.per_device[device].reset(new HostStageRing());
        HostStageRing& R = *tl_stage.per_device[device];
        if (!R.base) HIP_CHECK(hipHostMalloc((void**)&R.base, (size_t)HostStageRing::SLOTS * PAI_HOST_STAGE_MAX, hipHostMallocPortable | hipHostMallocMapped));
        // the readers of the previous slot have been enqueued by now (the contract): mark the point behind them
        if (R.pending >= 0) {
            const int k = R.pending;
            if (!R.ev[k]) HIP_CHECK(hipEventCreateWithFlags(&R.ev[k], hipEventDisableTiming));
            if (hipEventRecord(R.ev[k], R.pending_stream) == hipSuccess) {
                R.recorded[k] = true;
            } else {                                         // e.g. the stream no longer exists: everything enqueued has to finish
                (void)hipGetLastError();
                HIP_CHECK(hipDeviceSynchronize());
                R.recorded[k] = false;
            }
            R.pending = -1;
        }
        const int k = R.next;
        R.next = (R.next + 1) % HostStageRing::SLOTS;
        if (R.recorded[k]) HIP_CHECK(hipEventSynchronize(R.ev[k]));      // 31 stagings ago: normally long done
        char* p = R.base + (size_t)k * PAI_HOST_STAGE_MAX;
        for (int i = 0; i < parts; ++i) {
            if (bytes[i]) std::memcpy(p, h_src[i], bytes[i]);
            d_ptrs[i] = p;
            p += (bytes[i] + 15) & ~(size_t)15;
        }
        R.pending = k;
        R.pending_stream = (hipStream_t)stream;
    });
}

// ---- generic modulus ------------------------------------------------------------------------------
int pai_modulus_create(const uint32_t* h_m, int m_words, int device, pai_modulus** out) {
    return guarded([&] {
        require(h_m && out && m_words > 0, "bad arguments");
        std::unique_ptr<pai_modulus, ModulusDeleter> m(new pai_modulus());
        m->device = device;
        DeviceScope scope_(device);
        m->dev = scope_.info;
        m->ms.init(hbn::from_u32(h_m, (size_t)m_words));
        *out = m.release();
    });
}
void pai_modulus_destroy(pai_modulus* m) {
    if (!m) return;
    int prev_ = -1;
    (void)hipGetDevice(&prev_);
    (void)hipSetDevice(m->device);
    m->ms.release();
    m->table.release();
    m->expo.release();
    m->pinned.release();
    m->order.release();
    delete m;
    if (prev_ >= 0) (void)hipSetDevice(prev_);
}
int pai_modmul(pai_modulus* m, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N, uint32_t* d_out,
               void* stream) {
    return guarded([&] {
        require(m && d_a && d_b && d_out, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(m->device);
        const GeoOps* g = m->ms.geo;
        g->modmul((hipStream_t)stream, grid_for(g, N, m->dev.ncu), m->ms.d_ctx, d_a, d_b, d_out, (int)N, m->ms.w32, b_bcast, MODMUL_FULL);
        HIP_CHECK(hipGetLastError());
    });
}
int pai_modexp_fixed(pai_modulus* m, const uint32_t* d_base, const uint32_t* h_e, int e_words, size_t N,
                     uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(m && d_base && h_e && d_out && e_words > 0, "bad arguments");
        if (N == 0) return;
        std::lock_guard<std::mutex> lk(m->mu);
        DeviceScope scope_(m->device);
        const GeoOps* g = m->ms.geo;
        Limbs e = hbn::from_u32(h_e, (size_t)e_words);
        const int ebits = hbn::bitlen(e);
        hipStream_t s = (hipStream_t)stream;
        m->expo.ensure((size_t)e_words * 4);
        m->order.begin(s);
        m->pinned.h2d(m->expo.p, h_e, (size_t)e_words * 4, s);         // h_e is free when this call returns
        const int grid = grid_for(g, N, m->dev.ncu);
        m->table.ensure(g->table_words((size_t)grid) * 4);
        g->modexp_fixed(s, grid, m->ms.d_ctx, d_base, m->ms.w32, m->expo.as<uint32_t>(), e_words, ebits > 0 ? ebits : 1,
                        d_out, m->ms.w32, (int)N, m->table.as<uint32_t>(), 0);
        HIP_CHECK(hipGetLastError());
        m->order.end(s);
    });
}
// window width of the per-element-exponent kernels: table build 2^w - 2 products, then w squarings + 1 product per window
static bool lat_dense_disabled() {                  // PAI_DISABLE=lat_dense: small-batch stage A always on one integer per wavefront
    return knob_disabled("lat_dense");
}
static size_t lat_rl_max(size_t ncu) {              // PAI_TUNE lat_rl: largest batch of the wave-pair small-batch decryption (0 disables)
    long long v;
    return knob_tune("lat_rl", &v) ? (size_t)v : ncu;
}
static size_t lat_pp_max(size_t ncu, int chain_limbs, int key_bits) {      // PAI_TUNE lat_pp: most (ciphertext, prime) chains of the four-wave
    long long v;                                                            // digit-pair decryption (0 disables)
    if (knob_tune("lat_pp", &v)) return (size_t)v;
    // one limb per lane: 29 KB of LDS and < 100 registers per workgroup, five workgroups share a CU and further rounds follow —
    // measured (profiles/r05/lat_pp_range.jsonl, k_dec_a alone, ms): 2048-bit keys 1.56 up to 128 ciphertexts, 1.92 / 2.2 / 2.6 / 3.25 /
    // 3.85 at 256 / 384 / 512 / 640 / 768 against 3.8 (<= 512) and 4.6 of the window kernels, behind at 1 024 (4.9 / 4.6);
    // 3072-bit 3.0 .. 11.9 up to 1 280 against 7.9 .. 15.8; two limbs per lane (4096-bit): 6.7 / 9.4 / 11.4 up to 384 against 12.4 .. 14.1
    if (chain_limbs == 1) return (key_bits <= 2048 ? 6 : 10) * ncu;
    return 3 * ncu;
}
static size_t lat_enc_tree_max(size_t ncu) {        // PAI_TUNE lat_enc_tree: largest batch of the wave-shared small-batch encryption (0 disables)
    long long v;
    (void)ncu;
    return knob_tune("lat_enc_tree", &v) ? (size_t)v : (size_t)1 << 30;                         // measured ahead over the whole latency range (2048-bit keys: 0.29 vs 0.98 ms up to 256
}                                                   // elements, 0.54 vs 1.01 at 1024, 1.63 vs 1.91 at 4096; profiles/r04/lat_enc_tree.jsonl)
static size_t lat_mul_pp_max(size_t ncu) {          // PAI_TUNE lat_mul_pp: largest batch of the four-wave digit-pair ct * pt (0 disables)
    long long v;                                    // 2048-bit keys, 53-bit exponents: 0.22 ms up to 256, 0.31 / 0.43 at 512 / 1 024 against 0.37 / 0.49
    return knob_tune("lat_mul_pp", &v) ? (size_t)v : 4 * ncu;
}
static size_t lat_mul_rl_max(size_t ncu) {          // PAI_TUNE lat_mul_rl: largest batch of the wave-pair small-batch ct * pt (0 disables)
    long long v;
    return knob_tune("lat_mul_rl", &v) ? (size_t)v : 2 * ncu;
}
static bool lat_enc_m1_disabled() {                 // PAI_DISABLE=lat_enc_m1: the wave-shared small-batch encryption on the conventional context
    return knob_disabled("lat_enc_m1");
}
static bool fb_chain_disabled() {                   // PAI_DISABLE=fb_chain: window bases by the table kernel's own squaring chain
    return knob_disabled("fb_chain");
}
static bool pair_ctmul_disabled() {                 // PAI_DISABLE=pair_ctmul: ct * pt above 2048-bit keys as products modulo n^2
    return knob_disabled("pair_ctmul");
}
static int var_window_bits(int ebits_max) { return ebits_max <= 24 ? 2 : (ebits_max <= 80 ? 3 : (ebits_max <= 240 ? 4 : 5)); }

int pai_modexp_var(pai_modulus* m, const uint32_t* d_base, int base_bcast, const uint32_t* d_e, int e_words,
                   int ebits_max, int e_bcast, size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(m && d_base && d_e && d_out && e_words > 0 && ebits_max > 0 && ebits_max <= 32 * e_words, "bad arguments");
        if (N == 0) return;
        DeviceScope scope_(m->device);
        const GeoOps* g = m->ms.geo;
        const int grid = grid_for(g, N, m->dev.ncu);
        if (!base_bcast && ebits_max > 8) {
            // fixed windows sized to the exponent width over a per-slot table (kernels_modexp.hpp: k_modexp_var_win)
            std::lock_guard<std::mutex> lk(m->mu);
            const int wbits = var_window_bits(ebits_max);
            m->table.ensure(((size_t)1 << wbits) * g->nl * (size_t)grid * g->epb * 4);
            m->order.begin((hipStream_t)stream);
            g->modexp_var_win((hipStream_t)stream, grid, m->ms.d_ctx, d_base, m->ms.w32, d_e, e_words, ebits_max, e_bcast, d_out,
                              m->ms.w32, (int)N, m->table.as<uint32_t>(), wbits, nullptr);
            HIP_CHECK(hipGetLastError());
            m->order.end((hipStream_t)stream);
            return;
        }
        g->modexp_var((hipStream_t)stream, grid, m->ms.d_ctx, d_base, m->ms.w32, base_bcast ? 31 : 0,
                      d_e, e_words, ebits_max, e_bcast, d_out, m->ms.w32, (int)N, 0, 0);
        HIP_CHECK(hipGetLastError());
    });
}

// Sliding-window schedule (PADIC_SLIDE_BITS) of a wave-uniform exponent e, most significant bit first: each entry =
// (#squarings | table index << 8), applied as "square nsq times, then multiply by base^(2 idx + 1)"; index 0xFF = no
// multiplication (runs of more than 255 squarings, trailing zeros).  The first entry loads its table element.
static std::vector<uint16_t> compile_sliding_schedule(const Limbs& e) {
    auto bit = [&](int i) { return i >= 0 && ((e[i / 32] >> (i % 32)) & 1u); };
    std::vector<uint16_t> ops;
    int i = hbn::bitlen(e) - 1;
    int pending_sq = 0;
    bool first = true;
    while (i >= 0) {
        if (!bit(i)) { ++pending_sq; --i; continue; }
        int l = std::min(PADIC_SLIDE_BITS, i + 1);
        while (!bit(i - l + 1)) --l;                          // window must end in a 1
        uint32_t val = 0;
        for (int k = 0; k < l; ++k) val = (val << 1) | (bit(i - k) ? 1u : 0u);
        const int idx = (int)(val >> 1);                      // odd value 2 idx + 1
        int nsq = first ? 0 : pending_sq + l;
        while (nsq > 255) { ops.push_back((uint16_t)(255 | (0xFF << 8))); nsq -= 255; }
        ops.push_back((uint16_t)(nsq | (idx << 8)));
        first = false;
        pending_sq = 0;
        i -= l;
    }
    while (pending_sq > 0) { int c = std::min(pending_sq, 255); ops.push_back((uint16_t)(c | (0xFF << 8))); pending_sq -= c; }
    return ops;
}

// ---- public key -----------------------------------------------------------------------------------
}  // extern "C"

namespace {

std::vector<uint32_t> pubkey_digits_of(const pai_pubkey* pk, const Limbs& v) {
    const int pnl = pk->penc_nl;
    Limbs rem;
    Limbs quo = hbn::divq(v, pk->n, &rem);
    std::vector<uint32_t> h(2 * (size_t)pnl, 0);
    auto ra = hbn::to_r29(rem, pnl), rb = hbn::to_r29(quo, pnl);
    std::memcpy(h.data(), ra.data(), (size_t)pnl * 4);
    std::memcpy(h.data() + pnl, rb.data(), (size_t)pnl * 4);
    return h;
}
uint32_t* upload_vec(const std::vector<uint32_t>& h) {
    uint32_t* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, h.size() * 4));
    HIP_CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
}

// Lane-group fixed-base table T[j][d] = hs^(d 2^(wb j)) for the modulus context `ms` (Montgomery form for ITS R, raw
// radix-29 rows of ms.nl limbs).  Two levels when the window width is even: half-width windows
// S[i][e] = hs^(e 2^(h i)) (2 J windows of 2^h entries, binary method, a few thousand entries), then ONE product per
// entry, T[j][hi 2^h + lo] = S[2 j + 1][hi] * S[2 j][lo] (k_fb_expand).  Odd widths (only reachable through
// PAI_TUNE fb_wbits) keep the one-level build.
uint32_t* build_lane_group_fb(const pai_pubkey* pk, const ModSetup& ms, int wb, int J) {
    const int nl = ms.nl;
    const size_t ENT = (size_t)1 << wb;
    hbn::Mont32 mt(pk->nsq);
    const bool two_level = (wb % 2 == 0) && wb >= 8;
    const int h = two_level ? wb / 2 : wb;                     // bits per first-level window
    const int J1 = two_level ? 2 * J : J;
    std::vector<uint32_t> bases((size_t)J1 * pk->ct_words, 0);
    Limbs b = mt.to_mont(pk->hs);
    for (int j = 0; j < J1; ++j) {
        Limbs plain = mt.from_mont(b);
        std::memcpy(&bases[(size_t)j * pk->ct_words], plain.data(), plain.size() * 4);
        for (int s = 0; s < h; ++s) b = mt.mmul(b, b);
    }
    const size_t E1 = (size_t)1 << h, NE1 = (size_t)J1 * E1;
    std::vector<uint32_t> expo(NE1);
    for (size_t i = 0; i < NE1; ++i) expo[i] = (uint32_t)(i & (E1 - 1));
    DevBuf d_bases, d_expo, d_half;
    d_bases.ensure(bases.size() * 4);
    d_expo.ensure(NE1 * 4);
    HIP_CHECK(hipMemcpy(d_bases.p, bases.data(), bases.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_expo.p, expo.data(), NE1 * 4, hipMemcpyHostToDevice));
    const size_t NE = (size_t)J * ENT;
    uint32_t* d_fb = nullptr;
    HIP_CHECK(hipMalloc((void**)&d_fb, NE * (size_t)nl * 4));
    pk->fb_bytes += NE * (size_t)nl * 4;
    const GeoOps* g = ms.geo;
    uint32_t* level1 = d_fb;
    if (two_level) {
        d_half.ensure(NE1 * (size_t)nl * 4);
        level1 = d_half.as<uint32_t>();
    }
    const int g1 = (int)std::max<size_t>(1, std::min<size_t>((NE1 + g->epb - 1) / g->epb, (size_t)pk->dev.ncu * 8));
    g->modexp_var(nullptr, g1, ms.d_ctx, d_bases.as<uint32_t>(), pk->ct_words, h /* base = i >> h */,
                  d_expo.as<uint32_t>(), 1, h, 0, level1, 0, (int)NE1, 1 /*keep_mont*/, 1 /*out_raw*/);
    hipError_t e1 = hipGetLastError();
    if (two_level && e1 == hipSuccess) {
        const int g2 = (int)std::max<size_t>(1, std::min<size_t>((NE + g->epb - 1) / g->epb, (size_t)pk->dev.ncu * 8));
        g->fb_expand(nullptr, g2, ms.d_ctx, level1, d_fb, J, h);
        e1 = hipGetLastError();
    }
    hipError_t e2 = hipDeviceSynchronize();
    d_bases.release();
    d_expo.release();
    d_half.release();
    if (e1 != hipSuccess || e2 != hipSuccess) (void)hipFree(d_fb);
    HIP_CHECK(e1);
    HIP_CHECK(e2);
    return d_fb;
}

static bool ensure_lat_ctx(const pai_pubkey* pk);

// Digit-pair fixed-base table for the lane-group pair kernels: T[j][d] = pair(hs^(d 2^(wb j)) R), R = 2^(29 pair_nl).
// The host supplies pair(hs R) and pair(R); the window bases (squarings), the half-width windows (one sequential chain
// per window) and the full table (one product per entry) are computed on the device.
void build_pair_fb(pai_pubkey* pk, int wb, int J) {
    const int nl = pk->pair_nl;
    const bool two_level = (wb % 2 == 0) && wb >= 8;
    const int h = two_level ? wb / 2 : wb;
    const int J1 = two_level ? 2 * J : J;
    auto pair_of = [&](const Limbs& v, uint32_t* dst) {
        Limbs rem;
        Limbs quo = hbn::divq(v, pk->n, &rem);
        auto ra = hbn::to_r29(rem, nl), rb = hbn::to_r29(quo, nl);
        std::memcpy(dst, ra.data(), (size_t)nl * 4);
        std::memcpy(dst + nl, rb.data(), (size_t)nl * 4);
    };
    const Limbs Rm = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * nl), pk->nsq);
    std::vector<uint32_t> bases(2 * (size_t)nl, 0), one(2 * (size_t)nl, 0);
    pair_of(Rm, one.data());
    pair_of(hbn::mulmod(pk->hs, Rm, pk->nsq), bases.data());        // B_0; the other window bases are squared on the device
    ScopedDevBuf d_bases, d_one, d_half;
    d_bases.ensure(bases.size() * 4);
    d_one.ensure(one.size() * 4);
    HIP_CHECK(hipMemcpy(d_bases.p, bases.data(), bases.size() * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_one.p, one.data(), one.size() * 4, hipMemcpyHostToDevice));
    const size_t ent_words = 2 * (size_t)nl;
    const size_t NE = (size_t)J << wb, NE1 = (size_t)J1 << h;
    HIP_CHECK(hipMalloc((void**)&pk->d_pair_fb, NE * ent_words * 4));
    pk->fb_bytes += NE * ent_words * 4;
    uint32_t* level1 = pk->d_pair_fb;
    if (two_level) {
        d_half.ensure(NE1 * ent_words * 4);
        level1 = d_half.as<uint32_t>();
    }
    const int epb = pair_epb(nl);
    const int g1 = std::max(1, (J1 + epb - 1) / epb);
    // window bases from one chain of squarings on the integer-per-wavefront geometry (k_sq_chain), as for the digit engine
    FbBases fbb;
    ScopedDevBuf d_plain, d_hs_plain;
    if (pk->d_pair_kdig && ensure_lat_ctx(pk) && !fb_chain_disabled()) {
        std::vector<uint32_t> hw((size_t)pk->ct_words, 0);
        std::memcpy(hw.data(), pk->hs.data(), pk->hs.size() * 4);
        d_hs_plain.ensure(hw.size() * 4);
        HIP_CHECK(hipMemcpy(d_hs_plain.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        d_plain.ensure((size_t)J1 * pk->ct_words * 4);
        const GeoOps* gl = pk->lat_msq.geo;
        gl->sq_chain(nullptr, pk->lat_m1_ok ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx, pk->lat_m1_ok ? pk->lat_msq.d_ctx : nullptr,
                     d_hs_plain.as<uint32_t>(), pk->ct_words, d_plain.as<uint32_t>(), h, J1);
        HIP_CHECK(hipGetLastError());
        fbb.bases_plain = d_plain.as<uint32_t>();
        fbb.base_words = pk->ct_words;
        fbb.kdig = pk->d_pair_kdig;
        fbb.nd = pk->pair_nd;
    }
    bool ok = launch_pair_fb_chain(nl, nullptr, g1, pk->npair.d_ctx, pk->d_pair_nm1, d_bases.as<uint32_t>(), d_one.as<uint32_t>(),
                                   level1, J1, h, fbb);
    hipError_t e1 = hipGetLastError();
    if (ok && two_level && e1 == hipSuccess) {
        const int g2 = (int)std::max<size_t>(1, std::min<size_t>((NE + epb - 1) / epb, (size_t)pk->dev.ncu * 2));
        ok = launch_pair_fb_expand(nl, nullptr, g2, pk->npair.d_ctx, pk->d_pair_nm1, level1, pk->d_pair_fb, J, h);
        e1 = hipGetLastError();
    }
    hipError_t e2 = hipDeviceSynchronize();
    d_bases.release();
    d_one.release();
    d_half.release();
    d_plain.release();
    d_hs_plain.release();
    if (!ok || e1 != hipSuccess || e2 != hipSuccess) {
        (void)hipFree(pk->d_pair_fb);
        pk->d_pair_fb = nullptr;
    }
    if (!ok) throw PaiError(PAI_E_INTERNAL, "no digit-pair table kernel for this limb count");
    HIP_CHECK(e1);
    HIP_CHECK(e2);
    pk->pair_windows = J;
    pk->pair_wbits = wb;
}

// Constants of the four-wave digit-pair pipeline (kernels_declat.hpp) for one modulus s: the minus-one context of s' = s k
// (R = 2^(29 r) >= 2^8 s'), the base-s' digits of R^(i+2) mod s'^2 (an integer of in_bits bits into digit form) and
// R^-1 R_sq^(j+2) mod (s^2 k2) (a + b s' into the Montgomery form of sq_m1, the minus-one context of s^2).
// The chain's contexts: k of ONE limb (s' == -1 mod 2^29), one limb per lane where s' fits 60 limbs, else two.
static const GeoOps* pp_chain_geo(int limbs) {
    static const GeoOps g1 = [] { GeoOps o{}; o.nll = 1; o.t = 64; o.u = 1; o.nl = 64; o.epb = 4; return o; }();
    static const GeoOps g2 = [] { GeoOps o{}; o.nll = 2; o.t = 64; o.u = 1; o.nl = 128; o.epb = 4; return o; }();
    return limbs == 1 ? &g1 : &g2;
}
static bool build_pp_consts(const Limbs& smod, const ModSetup& sq_m1, const GeoOps* ga, int in_bits, ModSetup& pp,
                            uint32_t** d_kdig, uint32_t** d_kx, int* nd_out, int* nch_out, int* chain_limbs_out) {
    const Limbs one{1u};
    const int rows = ((hbn::bitlen(smod) + hbn::RB + 8 + hbn::RB - 1) / hbn::RB + 3) / 4 * 4;
    if (rows + 4 > PP_RMAX) return false;                 // (a digit row is read one group of four beyond its end)
    const int chain = rows <= 60 ? 1 : 2;                 // (rows < the limbs of the chain's geometry: the digit rows end in zeros)
    *chain_limbs_out = chain;
    pp.init_m1(smod, pp_chain_geo(chain), 8, 4);           // rows a multiple of the four the row loop takes at a time (its tail costs more than the rows it saves)
    (void)ga;
    const int r = pp.m1_rows;
    const int nd = (in_bits + hbn::RB * r - 1) / (hbn::RB * r);
    const int rows_sq = sq_m1.m1_rows;
    const int nch = (2 * r + 2 + rows_sq - 1) / rows_sq;
    if (r + 4 > PP_RMAX || r >= 64 * chain || nd > PP_MAXND || nch > PP_MAXCH || 2 * r + 2 > PP_YBUF) return false;
    *nd_out = nd;
    *nch_out = nch;
    const Limbs& Mp = pp.M;
    const Limbs Mp2 = hbn::mul(Mp, Mp);
    const Limbs Rm = hbn::mod(hbn::shl(one, hbn::RB * r), Mp2);
    Limbs K = hbn::mulmod(Rm, Rm, Mp2);
    std::vector<uint32_t> host((size_t)nd * 2 * r, 0);
    for (int i = 0; i < nd; ++i) {
        Limbs rem;
        Limbs quo = hbn::divq(K, Mp, &rem);
        auto ra = hbn::to_r29(rem, r), rb = hbn::to_r29(quo, r);
        std::memcpy(&host[(size_t)(2 * i) * r], ra.data(), (size_t)r * 4);
        std::memcpy(&host[(size_t)(2 * i + 1) * r], rb.data(), (size_t)r * 4);
        K = hbn::mulmod(K, Rm, Mp2);
    }
    *d_kdig = upload_vec(host);
    const Limbs& Msq = sq_m1.M;
    hbn::Mont32 mt(Msq);
    const Limbs inv2 = hbn::shr(hbn::add(Msq, one), 1);
    const Limbs rinv = mt.powmod(inv2, hbn::from_u64((uint64_t)hbn::RB * (uint64_t)r));      // R^-1 mod s^2 k2
    const Limbs Rsq = hbn::mod(hbn::shl(one, hbn::RB * rows_sq), Msq);
    Limbs Kx = hbn::mulmod(rinv, hbn::mulmod(Rsq, Rsq, Msq), Msq);
    const int nl = ga->nl;
    std::vector<uint32_t> hx((size_t)nch * nl, 0);
    for (int j = 0; j < nch; ++j) {
        auto rk = hbn::to_r29(Kx, nl);
        std::memcpy(&hx[(size_t)j * nl], rk.data(), (size_t)nl * 4);
        Kx = hbn::mulmod(Kx, Rsq, Msq);
    }
    *d_kx = upload_vec(hx);
    return true;
}

// Contexts of n^2 on the integer-per-wavefront (latency) geometry, built on first need under pk->mu: the conventional one
// (lat_msq) and, where it fits, the minus-one one (lat_msq_m1).  Returns false when no latency geometry is wide enough.
static bool ensure_lat_ctx(const pai_pubkey* pk) {
    if (!pk->lat_ready) {
        pk->lat_ready = true;
        if (const GeoOps* gl = geo_latency_for_bits(hbn::bitlen(pk->nsq))) {
            pk->lat_msq.init(pk->nsq, 0, gl);
            pk->lat_usable = true;
        }
    }
    if (pk->lat_usable && !pk->lat_m1_tried) {
        pk->lat_m1_tried = true;
        const GeoOps* g = pk->lat_msq.geo;
        const int need = hbn::bitlen(pk->nsq) + hbn::RB * g->u + 4;
        if (g->t >= 16 && (need + hbn::RB - 1) / hbn::RB + g->u <= g->nl) {
            pk->lat_msq_m1.init_m1(pk->nsq, g);
            pk->lat_m1_ok = true;
        }
    }
    if (pk->lat_m1_ok && !pk->lat_pp_tried) {
        pk->lat_pp_tried = true;
        if (pk->lat_msq.geo == geo_ops_3x64() && !knob_disabled("lat_pp"))
            pk->lat_pp_ok = build_pp_consts(pk->n, pk->lat_msq_m1, pk->lat_msq.geo, 32 * pk->ct_words, pk->lat_pp, &pk->d_lat_pp_kdig,
                                            &pk->d_lat_pp_kx, &pk->lat_pp_nd, &pk->lat_pp_nch, &pk->lat_pp_chain);
    }
    return pk->lat_usable;
}

// Small batches of ct + ct (one or two Montgomery products per element, all of them latency): n^2 spread over a wavefront per
// ciphertext instead of four lanes.  Returns the context to use on the latency geometry, or nullptr (throughput geometry).
// tagged: the product must come out as a b R^-1 with the THROUGHPUT geometry's R (pai_ct_mont_mul): MODMUL_FULL with the
// constant R_lat^2 / R = 2^(29 (2 nl_lat - nl)) in place of R_lat^2.
static size_t lat_add_max() {                       // PAI_LAT_ADD_MAX: largest ct + ct batch on the latency geometry (0 disables)
    if (const char* env = std::getenv("PAI_LAT_ADD_MAX")) return (size_t)std::strtoull(env, nullptr, 10);
    return (size_t)1024;
}
// Measured at 2048-bit keys (profiles/r04/lat_add_probe.jsonl): wire-form a b 30 against 60 us up to 1024 elements (39 / 65 at
// 2048, level at 4096), the tagged single product 29 against 35 us up to 1024 (level at 2048), aligned additions with shifts
// up to 13: 0.18 against 0.44 ms up to 1024, 0.31 / 0.45 at 4096 — hence the scale factors 2 / 1 / 4 on PAI_LAT_ADD_MAX.
static const ModSetup* lat_add_ctx(const pai_pubkey* pk, size_t N, bool tagged, int scale = 1) {
    if (N > (size_t)scale * lat_add_max()) return nullptr;
    std::lock_guard<std::mutex> lk(pk->mu);
    if (!ensure_lat_ctx(pk)) return nullptr;
    if (!tagged) return &pk->lat_msq;
    if (!pk->lat_tag_tried) {
        pk->lat_tag_tried = true;
        const int nl_lat = pk->lat_msq.nl, nl_thr = pk->msq.nl;
        if (2 * nl_lat >= nl_thr) {
            const Limbs c = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * (2 * nl_lat - nl_thr)), pk->nsq);
            pk->lat_msq_tag.init(pk->nsq, 0, pk->lat_msq.geo, &c);
            pk->lat_tag_ok = true;
        }
    }
    return pk->lat_tag_ok ? &pk->lat_msq_tag : nullptr;
}


// Fixed-base tables of the DJN obfuscator hs^r, built by the FIRST call that obfuscates (pai_encrypt with
// randomness / pai_obfuscate), under pk->mu: a handle that only adds, multiplies or decrypts — every unpickled
// ciphertext or public key on the receiving side of a federated exchange — never pays the multi-GB table.
// g-factoring of the finished digit-form table (kernels_padic_enc.hpp: k_fb_g_prefix / k_fb_g_finish + the wave-parallel
// extended GCD on the chunk totals): entries (a, d) become (a, t = d a^-1 mod n), after which every table product of an
// encryption is the 4 NL^2 rule.  Slabs bound the scratch (one digit per entry).  PAI_DISABLE=gform keeps the plain table;
// any failure (a non-unit would mean a broken key) leaves the table as it was built.
static void gfactor_digit_table(pai_pubkey* pk, size_t NE, int dwb) {
    if (knob_disabled("gform")) return;
    if (!padic_enc_gform_supported()) return;
    const int pnl = pk->penc_nl;
    // chunk length: divides the entries of a window, hence NE; one extended GCD per K entries.  64 measured best (first 2^20
    // encryption of a 2048-bit key 0.188 s; 256-entry chunks measured slower)
    int K = (int)std::min<size_t>(64, (size_t)1 << dwb);
    if (long long v; knob_tune("fb_gform_k", &v)) {                     // a power of two up to the window's entry count
        if (v >= 2 && (v & (v - 1)) == 0 && (size_t)v <= ((size_t)1 << dwb)) K = (int)v;
    }
    const int tw = pk->n_words;
    if ((tw + 63) / 64 > 4) return;                                      // inv_eea instantiations: up to 256 words
    const size_t slab_max = (size_t)1 << 22;                             // entries per slab: 1.2 GB of prefix scratch at 72 limbs
    const size_t slab = std::min(NE, slab_max) / K * K;
    ScopedDevBuf d_pref, d_tot, d_inv, d_fail;
    try {                                                                // no room for the scratch: keep the plain table (nothing was touched yet)
        d_pref.ensure(slab * (size_t)pnl * 4);
        d_tot.ensure(slab / K * (size_t)tw * 4);
        d_inv.ensure(slab / K * (size_t)tw * 4);
        d_fail.ensure(4);
    } catch (const PaiError&) {
        (void)hipGetLastError();
        return;
    }
    HIP_CHECK(hipMemset(d_fail.p, 0, 4));
    const int grid = pk->dev.ncu;                                         // the scratch column is sized for this grid
    const size_t ent_words = 2 * (size_t)pnl;
    // pass 1 and the inversions of every slab first (pass 2 overwrites the second digits: no partial conversion on failure)
    // -> with one slab of scratch the passes must alternate; a failure after some slabs were converted is handled by
    //    rebuilding (fb_ready stays false and the caller's catch frees the table)
    for (size_t e0 = 0; e0 < NE; e0 += slab) {
        const size_t cnt = std::min(slab, NE - e0);
        uint32_t* tbl = pk->d_fb_dig + e0 * ent_words;
        if (!launch_fb_g_prefix_padic(pnl, nullptr, grid, pk->nmod.d_ctx, tbl, cnt, K, d_pref.as<uint32_t>(), d_tot.as<uint32_t>(), tw,
                                      pk->d_mscratch))
            throw PaiError(PAI_E_INTERNAL, "no g-factoring kernel for this limb count");
        HIP_CHECK(hipGetLastError());
        if (!launch_inv_eea(nullptr, tw, pk->d_nexp, d_tot.as<uint32_t>(), d_inv.as<uint32_t>(), (int)(cnt / K), 2 * 32 * tw + 64,
                            d_fail.as<int>()))
            throw PaiError(PAI_E_INTERNAL, "no extended-GCD instantiation for this key size");
        HIP_CHECK(hipGetLastError());
        int fail = 0;
        HIP_CHECK(hipMemcpy(&fail, d_fail.p, 4, hipMemcpyDeviceToHost));
        if (fail) throw PaiError(PAI_E_INTERNAL, "fixed-base table entry without an inverse modulo n");
        launch_fb_g_finish_padic(pnl, nullptr, grid, pk->nmod.d_ctx, tbl, cnt, K, d_pref.as<uint32_t>(), d_inv.as<uint32_t>(), tw,
                                 pk->d_mscratch);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
    pk->fb_gform = true;
}

// the same for the lane-group pair table of keys above 2048 bits (kernels_pair.hpp: k_pair_g_prefix / k_pair_g_finish)
static void gfactor_pair_table(pai_pubkey* pk, size_t NE, int wb) {
    if (knob_disabled("gform")) return;
    const int nl = pk->pair_nl;
    const int K = (int)std::min<size_t>(64, (size_t)1 << wb);
    const int tw = pk->n_words;
    if ((tw + 63) / 64 > 4) return;
    const size_t slab = std::min(NE, (size_t)1 << 21) / K * K;            // 2^21 entries: 1.2 GB of prefix scratch at 144 limbs
    ScopedDevBuf d_pref, d_tot, d_inv, d_fail;
    try {
        d_pref.ensure(slab * (size_t)nl * 4);
        d_tot.ensure(slab / K * (size_t)tw * 4);
        d_inv.ensure(slab / K * (size_t)tw * 4);
        d_fail.ensure(4);
    } catch (const PaiError&) {
        (void)hipGetLastError();
        return;
    }
    HIP_CHECK(hipMemset(d_fail.p, 0, 4));
    const int epb = pair_epb(nl);
    const size_t ent_words = 2 * (size_t)nl;
    for (size_t e0 = 0; e0 < NE; e0 += slab) {
        const size_t cnt = std::min(slab, NE - e0);
        uint32_t* tbl = pk->d_pair_fb + e0 * ent_words;
        const int grid = (int)std::max<size_t>(1, std::min<size_t>((cnt / K + epb - 1) / epb, (size_t)pk->dev.ncu * 2));
        if (!launch_pair_g_prefix(nl, nullptr, grid, pk->npair.d_ctx, tbl, cnt, K, d_pref.as<uint32_t>(), d_tot.as<uint32_t>(), tw))
            throw PaiError(PAI_E_INTERNAL, "no g-factoring kernel for this limb count");
        HIP_CHECK(hipGetLastError());
        if (!launch_inv_eea(nullptr, tw, pk->d_nexp, d_tot.as<uint32_t>(), d_inv.as<uint32_t>(), (int)(cnt / K), 2 * 32 * tw + 64,
                            d_fail.as<int>()))
            throw PaiError(PAI_E_INTERNAL, "no extended-GCD instantiation for this key size");
        HIP_CHECK(hipGetLastError());
        int fail = 0;
        HIP_CHECK(hipMemcpy(&fail, d_fail.p, 4, hipMemcpyDeviceToHost));
        if (fail) throw PaiError(PAI_E_INTERNAL, "fixed-base table entry without an inverse modulo n");
        launch_pair_g_finish(nl, nullptr, grid, pk->npair.d_ctx, tbl, cnt, K, d_pref.as<uint32_t>(), d_inv.as<uint32_t>(), tw);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
    pk->fb_gform = true;
}

// ---- per-device cache of the DJN fixed-base tables (round 4) ---------------------------------------------------
// Every DJN key builds a multi-GB table on its first obfuscating call.  A process that holds many keys (federated
// learning: one key per party or per round) used to need pai_pubkey_trim by hand; now the handles with built tables of a
// device form an LRU list under a byte budget — PAI_FB_CACHE_MB, default half of the device memory — and a build that
// would pass the budget first returns the tables of the least recently used handles (which rebuild on their next
// obfuscating call, bit-identical).  Lock order: own pk->mu, then the registry, then try_lock of a victim (a busy victim
// is skipped, never waited for).
struct FbRegistry {
    std::mutex mu;
    std::vector<pai_pubkey*> lru;      // most recently used last
};
static FbRegistry g_fb;
static size_t fb_cache_budget(size_t mem_total) {
    if (const char* env = std::getenv("PAI_FB_CACHE_MB")) { double v = std::atof(env); if (v >= 1.0) return (size_t)(v * 1048576.0); }
    return mem_total / 2;
}
static void fb_free_tables(pai_pubkey* pk) {          // caller holds pk->mu and has synchronised the device
    if (pk->d_fb) { (void)hipFree(pk->d_fb); pk->d_fb = nullptr; }
    if (pk->d_fb_dig) { (void)hipFree(pk->d_fb_dig); pk->d_fb_dig = nullptr; }
    if (pk->d_pair_fb) { (void)hipFree(pk->d_pair_fb); pk->d_pair_fb = nullptr; }
    pk->fb_ready = false;
    pk->fb_bytes = 0;
}
static void fb_unregister(pai_pubkey* pk) {
    std::lock_guard<std::mutex> g(g_fb.mu);
    g_fb.lru.erase(std::remove(g_fb.lru.begin(), g_fb.lru.end(), pk), g_fb.lru.end());
}
static void fb_touch(pai_pubkey* pk) {                // caller holds pk->mu
    std::lock_guard<std::mutex> g(g_fb.mu);
    auto it = std::find(g_fb.lru.begin(), g_fb.lru.end(), pk);
    if (it != g_fb.lru.end() && it + 1 != g_fb.lru.end()) std::rotate(it, it + 1, g_fb.lru.end());
}
// makes room for `need` more table bytes on pk's device; returns the bytes it freed
static size_t fb_make_room(pai_pubkey* pk, size_t need, size_t mem_total) {
    const size_t budget = fb_cache_budget(mem_total);
    size_t freed = 0;
    std::lock_guard<std::mutex> g(g_fb.mu);
    size_t used = 0;
    for (pai_pubkey* o : g_fb.lru) if (o->device == pk->device) used += o->fb_bytes;
    for (size_t i = 0; i < g_fb.lru.size() && used + need > budget;) {
        pai_pubkey* v = g_fb.lru[i];
        if (v == pk || v->device != pk->device || !v->mu.try_lock()) { ++i; continue; }
        (void)hipDeviceSynchronize();                  // nothing in flight may still read the victim's tables
        used -= std::min(used, v->fb_bytes);
        freed += v->fb_bytes;
        fb_free_tables(v);
        v->mu.unlock();
        g_fb.lru.erase(g_fb.lru.begin() + (long)i);
    }
    return freed;
}

// Table size of a key: the big tables (1/32 of the device memory: 8.6 GB at 2048-bit keys) are for the few keys a process
// works with at a time.  A handle that finds PAI_FB_BIG_KEYS (default 8) built tables on its device already, or whose big
// table would not fit the cache budget beside the resident ones, takes the small operating point instead
// (PAI_FB_SMALL_TABLE_MB, default 256: 12-bit windows, 0.2 GB at 2048-bit keys, ~1.6 x the encryption time) — a server
// holding a hundred parties' keys neither exhausts the device nor evicts and rebuilds a multi-GB table on every call.
static size_t fb_small_table_bytes() {
    if (const char* env = std::getenv("PAI_FB_SMALL_TABLE_MB")) { double v = std::atof(env); if (v >= 1.0) return (size_t)(v * 1048576.0); }
    return (size_t)256 << 20;
}
static int fb_big_keys() {
    if (const char* env = std::getenv("PAI_FB_BIG_KEYS")) { int v = std::atoi(env); if (v >= 0) return v; }
    return 8;
}
static void fb_drop_tables(pai_pubkey* pk) {           // a failed build leaves nothing behind
    if (pk->d_fb_dig) { (void)hipFree(pk->d_fb_dig); pk->d_fb_dig = nullptr; }
    if (pk->d_pair_fb) { (void)hipFree(pk->d_pair_fb); pk->d_pair_fb = nullptr; }
    if (pk->d_fb) { (void)hipFree(pk->d_fb); pk->d_fb = nullptr; }
    pk->fb_ready = false;
    pk->fb_bytes = 0;
}
static void build_fb_tables_body(pai_pubkey* pk);
void build_fb_tables(const pai_pubkey* cpk) {
    pai_pubkey* pk = const_cast<pai_pubkey*>(cpk);
    if (!pk->djn) return;
    if (pk->fb_ready) { fb_touch(pk); return; }
    size_t mem_free_b = 0, mem_total_b = 0;
    HIP_CHECK(hipMemGetInfo(&mem_free_b, &mem_total_b));
    // the largest table the sizing rules below produce is 1/32 of the device memory (PAI_FB_TABLE_MB may ask for more)
    size_t need = mem_total_b / 32;
    bool pinned = false;
    if (const char* env = std::getenv("PAI_FB_TABLE_MB")) { double v = std::atof(env); if (v >= 1.0) { need = (size_t)(v * 1048576.0); pinned = true; } }
    pk->fb_table_budget = 0;
    if (!pinned) {
        size_t used = 0; int resident = 0;
        {
            std::lock_guard<std::mutex> g(g_fb.mu);
            for (pai_pubkey* o : g_fb.lru) if (o->device == pk->device && o != pk) { used += o->fb_bytes; ++resident; }
        }
        if (resident >= fb_big_keys() || used + need > fb_cache_budget(mem_total_b)) {
            need = std::min(need, fb_small_table_bytes());
            pk->fb_table_budget = need;
        }
    }
    fb_make_room(pk, need, mem_total_b);
    for (int attempt = 0;; ++attempt) {
        try {
            pk->fb_bytes = 0;                           // the builders add what they allocate for the tables
            build_fb_tables_body(pk);
            std::lock_guard<std::mutex> g(g_fb.mu);
            g_fb.lru.push_back(pk);
            return;
        } catch (const PaiError& e) {
            // a failed build (out of memory under pressure, a HIP error between the table allocation and fb_ready) must not
            // leave a multi-GB table behind: the next obfuscating call would allocate over the dangling pointer
            fb_drop_tables(pk);
            (void)hipGetLastError();
            // out of memory: return every other handle's tables on this device and try once more
            if (attempt == 0 && e.code == PAI_E_HIP && fb_make_room(pk, (size_t)-1 / 2, mem_total_b) > 0) continue;
            throw;
        } catch (...) {
            fb_drop_tables(pk);
            throw;
        }
    }
}
static void build_fb_tables_body(pai_pubkey* pk) {
    const int nl = pk->msq.nl;
    const int randbits = pk->randbits;
    // Fixed-base window width of the lane-group table (built only when the digit engine does not serve this key
    // size): the widest even width up to 16 bits whose table fits 1/32 of device memory (PAI_FB_TABLE_MB overrides) —
    // every window is one multiplication mod n^2 per ciphertext and the two-level build costs one product per entry
    // (4096-bit keys: 16 bits = 128 windows x 65536 entries x 1152 B = 9.7 GB; 14 bits: 147 windows, 2.8 GB).
    size_t mem_free0 = 0, mem_total0 = 0;
    HIP_CHECK(hipMemGetInfo(&mem_free0, &mem_total0));
    double lg_budget = pk->penc_nl ? 256.0 * 1048576.0
                                   : std::max(256.0 * 1048576.0, std::min((double)mem_total0 / 32.0, (double)mem_free0 / 4.0));
    if (!pk->penc_nl) {
        if (const char* env = std::getenv("PAI_FB_TABLE_MB")) { double v = std::atof(env); if (v >= 1.0) lg_budget = v * 1048576.0; }
        if (pk->fb_table_budget) lg_budget = (double)pk->fb_table_budget;      // the small operating point (build_fb_tables)
    }
    int wb = pk->penc_nl ? 12 : 16;
    while (wb > 4 && (double)((randbits + wb - 1) / wb) * (double)((size_t)1 << wb) * pk->msq.nl * 4.0 > lg_budget) wb -= (wb > 8 ? 2 : 1);
    if (long long v; knob_tune("fb_wbits", &v) && v >= 4 && v <= 16) wb = (int)v;
    pk->fb_wbits = wb;
    const int J = (randbits + wb - 1) / wb;
    const size_t ENT = (size_t)1 << wb;
    pk->fb_windows = J;
    if (pk->pair_nl) {
        build_pair_fb(pk, wb, J);
        pk->fb_gform = false;
        gfactor_pair_table(pk, (size_t)J << wb, wb);
    } else if (!pk->penc_nl) {
        pk->d_fb = build_lane_group_fb(pk, pk->msq, wb, J);
    } else {
        // digit-form fixed-base table for the base-n digit engine
        const int pnl = pk->penc_nl;
        const Limbs Rm = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * pnl), pk->nsq);
        uint32_t* d_one = pk->d_one_dig;
        ScopedDevBuf d_hs, d_half;
        {
            const std::vector<uint32_t> h = pubkey_digits_of(pk, hbn::mulmod(pk->hs, Rm, pk->nsq));
            d_hs.ensure(h.size() * 4);
            HIP_CHECK(hipMemcpy(d_hs.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        // Window width of the digit-form table.  Every window costs one multiplication mod n^2 per
        // ciphertext, and HBM is plentiful: take the widest even width (<= 20 bits) whose table fits the
        // budget — 1/32 of the device memory unless PAI_FB_TABLE_MB says otherwise (MI355X, 288 GB:
        // 9 GB => 2048-bit keys get 18 bits, 57 windows x 262144 entries x 576 B = 8.6 GB; measured
        // k_encrypt per 2^20: 109 / 95 / 84 / 75 / 70 ms at 12 / 14 / 16 / 18 / 20 bits).
        // PAI_TUNE fb_digit_wbits pins the width (<= 12, or an even value up to 20).
        const size_t ent_bytes = 2 * (size_t)pnl * 4;
        auto table_bytes = [&](int w) { return (double)((randbits + w - 1) / w) * (double)((size_t)1 << w) * (double)ent_bytes; };
        size_t mem_free = 0, mem_total = 0;
        HIP_CHECK(hipMemGetInfo(&mem_free, &mem_total));
        double budget = std::min((double)mem_total / 32.0, (double)mem_free / 4.0);   // never more than a quarter of what is free
        if (const char* env = std::getenv("PAI_FB_TABLE_MB")) { double v = std::atof(env); if (v >= 1.0) budget = v * 1048576.0; }
        if (pk->fb_table_budget) budget = (double)pk->fb_table_budget;         // the small operating point (build_fb_tables)
        int dwb = wb;
        for (int cand = 20; cand > 12; cand -= 2)
            if (table_bytes(cand) <= budget) { dwb = cand; break; }
        if (long long v; knob_tune("fb_digit_wbits", &v)) {
            if ((v >= 4 && v <= 12) || (v > 12 && v <= 20 && v % 2 == 0)) dwb = (int)v;
        }
        const int DJ = (randbits + dwb - 1) / dwb;
        pk->fbd_wbits = dwb;
        pk->fbd_windows = DJ;
        HIP_CHECK(hipMalloc((void**)&pk->d_fb_dig, ((size_t)DJ << dwb) * ent_bytes));
        pk->fb_bytes += ((size_t)DJ << dwb) * ent_bytes;
        bool ok = true;
        // window bases hs^(2^(h j)): one chain of squarings on the integer-per-wavefront geometry (k_sq_chain, ~6 us per
        // product) instead of the same chain walked by every lane of the table kernel at 50 us per product
        const int h1 = dwb <= 12 ? dwb : dwb / 2, J1 = dwb <= 12 ? DJ : 2 * DJ;
        FbBases fbb;
        ScopedDevBuf d_bases, d_hs_plain;
        if (pk->d_ct_kdig && ensure_lat_ctx(pk) && !fb_chain_disabled()) {
            const std::vector<uint32_t> hw = [&] { std::vector<uint32_t> v((size_t)pk->ct_words, 0); std::memcpy(v.data(), pk->hs.data(), pk->hs.size() * 4); return v; }();
            d_hs_plain.ensure(hw.size() * 4);
            HIP_CHECK(hipMemcpy(d_hs_plain.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            d_bases.ensure((size_t)J1 * pk->ct_words * 4);
            const GeoOps* gl = pk->lat_msq.geo;
            gl->sq_chain(nullptr, pk->lat_m1_ok ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx, pk->lat_m1_ok ? pk->lat_msq.d_ctx : nullptr,
                         d_hs_plain.as<uint32_t>(), pk->ct_words, d_bases.as<uint32_t>(), h1, J1);
            HIP_CHECK(hipGetLastError());
            fbb.bases_plain = d_bases.as<uint32_t>();
            fbb.base_words = pk->ct_words;
            fbb.kdig = pk->d_ct_kdig;
            fbb.nd = pk->ct_nd;
        }
        if (dwb <= 12) {
            ok = launch_fb_table_padic(pnl, nullptr, pk->nmod.d_ctx, pk->d_nm1, d_hs.as<uint32_t>(), d_one, pk->d_fb_dig, DJ, dwb, fbb);
        } else {
            // two levels: half-width windows at twice the density (sequential chains of 2^h entries), then
            // one parallel pass of DJ * 2^dwb independent products
            const int h = dwb / 2;
            d_half.ensure(((size_t)(2 * DJ) << h) * ent_bytes);
            ok = launch_fb_table_padic(pnl, nullptr, pk->nmod.d_ctx, pk->d_nm1, d_hs.as<uint32_t>(), d_one, d_half.as<uint32_t>(), 2 * DJ, h, fbb) &&
                 launch_fb_expand_padic(pnl, nullptr, pk->dev.ncu, pk->nmod.d_ctx, pk->d_nm1, d_half.as<uint32_t>(), pk->d_fb_dig, DJ, h,
                                        pk->d_mscratch);
        }
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        d_hs.release();
        d_half.release();
        d_bases.release();
        d_hs_plain.release();
        if (!ok) throw PaiError(PAI_E_INTERNAL, "no digit-engine table kernel for this limb count");
        HIP_CHECK(e1);
        HIP_CHECK(e2);
        pk->fb_gform = false;
        gfactor_digit_table(pk, (size_t)DJ << dwb, dwb);
    }
    pk->fb_ready = true;
}

}  // namespace

extern "C" {

int pai_pubkey_create(const uint32_t* h_n, int n_words, int key_bits, const uint32_t* h_hs, int hs_words,
                      int randbits, int device, pai_pubkey** out) {
    return guarded([&] {
        require(h_n && out && n_words > 0 && key_bits > 0, "bad arguments");
        DeviceScope scope_(device);
        std::unique_ptr<pai_pubkey, PubkeyDeleter> pk(new pai_pubkey());     // frees device memory if set-up throws midway
        pk->device = device;
        pk->dev = scope_.info;
        pk->n = hbn::from_u32(h_n, (size_t)n_words);
        require(hbn::is_odd(pk->n) && hbn::bitlen(pk->n) > 16, "n must be an odd integer of more than 16 bits");
        require(hbn::bitlen(pk->n) <= key_bits, "n is wider than key_bits");
        pk->key_bits = key_bits;
        pk->n_words = words_for_bits(key_bits);
        pk->ct_words = 2 * pk->n_words;
        pk->nsq = hbn::mul(pk->n, pk->n);
        pk->msq.init(pk->nsq);
        const int nl = pk->msq.nl;
        pk->d_nR = upload_r29(hbn::mulmod(pk->n, pk->msq.R, pk->nsq), nl);
        pk->d_nexp = upload_words(pk->n, pk->n_words);
        pk->d_nsq_words = upload_words(pk->nsq, pk->ct_words);
        {   // per-level constants of the single-product trees (see pai_pubkey::d_tree_c)
            const int K = pai_pubkey::TREE_LEVELS;
            const size_t W = (size_t)pk->ct_words;
            std::vector<uint32_t> hc((size_t)K * W, 0), hf((size_t)K * W, 0);
            const Limbs one{1u};
            Limbs inv2 = hbn::shr(hbn::add(pk->nsq, one), 1);                 // 2^-1 mod n^2
            hbn::Mont32 mt(pk->nsq);
            Limbs rinv = mt.powmod(inv2, hbn::from_u64((uint64_t)hbn::RB * (uint64_t)nl));
            require(hbn::cmp(hbn::mulmod(rinv, pk->msq.R, pk->nsq), one) == 0, "R^-1 check failed");
            Limbs c = one, f = pk->msq.R;                                     // R^(1 - 2^0) = 1, R^(2^0) = R
            for (int k = 0; k < K; ++k) {
                std::memcpy(&hc[(size_t)k * W], c.data(), c.size() * 4);
                std::memcpy(&hf[(size_t)k * W], f.data(), f.size() * 4);
                c = hbn::mulmod(hbn::mulmod(c, c, pk->nsq), rinv, pk->nsq);   // 1 - 2^(k+1) = 2 (1 - 2^k) - 1
                f = hbn::mulmod(f, f, pk->nsq);
            }
            pk->d_tree_c = upload_vec(hc);
            pk->d_tree_fix = upload_vec(hf);
        }
        // ---- base-n digit engine (kernels_padic_enc.hpp): raw / DJN encryption and ct * pt run on it when n fits 72
        // limbs (PAI_DISABLE=padic falls back to the lane-group kernels, which serve every other key size)
        pk->penc_nl = padic_enc_nl_for_n_bits(hbn::bitlen(pk->n));
        if (knob_disabled("padic")) pk->penc_nl = 0;
        if (pk->penc_nl) {
            const int pnl = pk->penc_nl;
            pk->nmod.init(pk->n, pnl);
            const Limbs one{1u};
            pk->d_nm1 = upload_r29(hbn::sub(pk->n, one), pnl);
            pk->d_nsq29 = upload_r29(pk->nsq, 2 * pnl);
            const Limbs Rm = hbn::mod(hbn::shl(one, hbn::RB * pnl), pk->nsq);
            pk->d_one_dig = upload_vec(pubkey_digits_of(pk.get(), Rm));
            HIP_CHECK(hipMalloc((void**)&pk->d_mscratch, 2 * (size_t)pk->dev.ncu * BLOCK_THREADS * (size_t)pnl * 4));
            // digit pairs of R^(i+2) mod n^2: a ciphertext enters digit form through its base-R digits (as in stage A)
            pk->ct_nd = (32 * pk->ct_words + hbn::RB * pnl - 1) / (hbn::RB * pnl);
            std::vector<uint32_t> kd;
            Limbs K = hbn::mulmod(Rm, Rm, pk->nsq);
            for (int i = 0; i < pk->ct_nd; ++i) {
                auto h = pubkey_digits_of(pk.get(), K);
                kd.insert(kd.end(), h.begin(), h.end());
                K = hbn::mulmod(K, Rm, pk->nsq);
            }
            pk->d_ct_kdig = upload_vec(kd);
        }
        if (!pk->penc_nl) {
            // wider moduli: DJN obfuscation on base-n digit pairs spread over lane groups (PAI_DISABLE=pair: lane-group
            // products modulo n^2 as in round 1)
            pk->pair_nl = pair_nl_for_n_bits(hbn::bitlen(pk->n));
            if (knob_disabled("pair")) pk->pair_nl = 0;
            if (pk->pair_nl) {
                pk->npair.init(pk->n, pk->pair_nl);
                pk->d_pair_nm1 = upload_r29(hbn::sub(pk->n, Limbs{1u}), pk->pair_nl);
                pk->pair_out_words = (hbn::RB * pk->pair_nl + 31) / 32;
                // ct * pt on digit pairs (k_pair_ctmul): a ciphertext enters digit form through its base-R digits D_i,
                // sum_i (D_i, 0) (x) pair(R^(i+2) mod n^2); pair(x) = (x mod n, x div n)
                const int pnl = pk->pair_nl;
                auto pair_of = [&](const Limbs& v, std::vector<uint32_t>& dst) {
                    Limbs rem;
                    Limbs quo = hbn::divq(v, pk->n, &rem);
                    auto ra = hbn::to_r29(rem, pnl), rb = hbn::to_r29(quo, pnl);
                    dst.insert(dst.end(), ra.begin(), ra.end());
                    dst.insert(dst.end(), rb.begin(), rb.end());
                };
                const Limbs Rm = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * pnl), pk->nsq);
                pk->pair_nd = (32 * pk->ct_words + hbn::RB * pnl - 1) / (hbn::RB * pnl);
                std::vector<uint32_t> kd, one;
                Limbs K = hbn::mulmod(Rm, Rm, pk->nsq);
                for (int i = 0; i < pk->pair_nd; ++i) {
                    pair_of(K, kd);
                    K = hbn::mulmod(K, Rm, pk->nsq);
                }
                pair_of(Rm, one);
                pk->d_pair_kdig = upload_vec(kd);
                pk->d_pair_one = upload_vec(one);
            }
        }
        if (h_hs) {
            require(hs_words > 0 && randbits > 0, "DJN key needs hs and randbits");
            pk->djn = true;
            pk->hs = hbn::from_u32(h_hs, (size_t)hs_words);
            require(hbn::cmp(pk->hs, pk->nsq) < 0 && !hbn::is_zero(pk->hs), "hs must lie in (0, n^2)");
            pk->randbits = randbits;
            pk->r_words = words_for_bits(randbits);
            // the fixed-base tables for hs are built by the first obfuscating call (build_fb_tables)
        } else {
            pk->djn = false;
            pk->randbits = 0;
            pk->r_words = pk->n_words;
            if (pk->penc_nl) {        // standard-scheme obfuscator r^n on the base-n digit engine (k_pow_padic)
                const std::vector<uint16_t> ops = compile_sliding_schedule(pk->n);
                pk->pow_nops = (int)ops.size();
                HIP_CHECK(hipMalloc((void**)&pk->d_pow_ops, ops.size() * 2));
                HIP_CHECK(hipMemcpy(pk->d_pow_ops, ops.data(), ops.size() * 2, hipMemcpyHostToDevice));
            }
        }
        *out = pk.release();
    });
}

void pai_pubkey_destroy(pai_pubkey* pk) {
    if (!pk) return;
    fb_unregister(pk);
    int prev_ = -1;
    (void)hipGetDevice(&prev_);
    (void)hipSetDevice(pk->device);
    pk->msq.release();
    if (pk->d_nR) (void)hipFree(pk->d_nR);
    if (pk->d_fb) (void)hipFree(pk->d_fb);
    if (pk->d_nexp) (void)hipFree(pk->d_nexp);
    pk->nmod.release();
    if (pk->d_nm1) (void)hipFree(pk->d_nm1);
    if (pk->d_nsq29) (void)hipFree(pk->d_nsq29);
    if (pk->d_fb_dig) (void)hipFree(pk->d_fb_dig);
    if (pk->d_mscratch) (void)hipFree(pk->d_mscratch);
    if (pk->d_one_dig) (void)hipFree(pk->d_one_dig);
    if (pk->d_ct_kdig) (void)hipFree(pk->d_ct_kdig);
    if (pk->d_pow_ops) (void)hipFree(pk->d_pow_ops);
    pk->npair.release();
    if (pk->d_pair_nm1) (void)hipFree(pk->d_pair_nm1);
    if (pk->d_pair_fb) (void)hipFree(pk->d_pair_fb);
    if (pk->d_pair_kdig) (void)hipFree(pk->d_pair_kdig);
    if (pk->d_pair_one) (void)hipFree(pk->d_pair_one);
    pk->pair_ct_table.release();
    pk->pair_wv.release();
    pk->midp_n.release();
    if (pk->d_midp_nm1) (void)hipFree(pk->d_midp_nm1);
    if (pk->d_midp_kdig) (void)hipFree(pk->d_midp_kdig);
    if (pk->d_midp_one) (void)hipFree(pk->d_midp_one);
    pk->ctmul_table.release();
    pk->pow2_expo.release();
    pk->mexp_table.release();
    pk->mexp_partial.release();
    if (pk->d_nsq_words) (void)hipFree(pk->d_nsq_words);
    if (pk->d_tree_c) (void)hipFree(pk->d_tree_c);
    if (pk->d_tree_fix) (void)hipFree(pk->d_tree_fix);
    if (pk->d_rpow) (void)hipFree(pk->d_rpow);
    pk->prod_a.release();
    pk->prod_b.release();
    pk->lat_msq.release();
    pk->lat_msq_m1.release();
    pk->lat_pp.release();
    if (pk->d_lat_pp_kdig) (void)hipFree(pk->d_lat_pp_kdig);
    if (pk->d_lat_pp_kx) (void)hipFree(pk->d_lat_pp_kx);
    pk->lat_msq_tag.release();
    pk->lat_table.release();
    if (pk->d_lat_nR) (void)hipFree(pk->d_lat_nR);
    if (pk->d_lat_fb) (void)hipFree(pk->d_lat_fb);
    if (pk->d_lat_fb_m1) (void)hipFree(pk->d_lat_fb_m1);
    if (pk->d_lat_nR_m1) (void)hipFree(pk->d_lat_nR_m1);
    pk->order.release();
    pk->inv_prod.release();
    pk->inv_inv.release();
    pk->inv_fail.release();
    pk->status.release();
    pk->table.release();
    pk->tmp.release();
    delete pk;
    if (prev_ >= 0) (void)hipSetDevice(prev_);
}

int pai_pubkey_trim(pai_pubkey* pk, size_t* freed_bytes) {
    return guarded([&] {
        require(pk != nullptr, "pk is NULL");
        std::lock_guard<std::mutex> lk(pk->mu);
        DeviceScope scope_(pk->device);
        HIP_CHECK(hipDeviceSynchronize());                     // nothing in flight may still read the tables / scratch
        size_t before = 0, after = 0, total = 0;
        HIP_CHECK(hipMemGetInfo(&before, &total));
        // the DJN fixed-base tables (rebuilt by the next obfuscating call) ...
        fb_unregister(pk);
        fb_free_tables(pk);
        if (pk->d_lat_fb) { (void)hipFree(pk->d_lat_fb); pk->d_lat_fb = nullptr; }
        if (pk->d_lat_fb_m1) { (void)hipFree(pk->d_lat_fb_m1); pk->d_lat_fb_m1 = nullptr; }
        if (pk->d_lat_nR) { (void)hipFree(pk->d_lat_nR); pk->d_lat_nR = nullptr; }
        pk->lat_fb_ready = false;
        // ... and the grow-only scratch of the batch operations (re-grown on demand)
        pk->table.release();
        pk->tmp.release();
        pk->lat_table.release();
        pk->ctmul_table.release();
        pk->pair_ct_table.release();
        pk->pair_wv.release();
        pk->pow2_expo.release();
        pk->mexp_table.release();
        pk->mexp_partial.release();
        pk->inv_prod.release();
        pk->inv_inv.release();
        pk->prod_a.release();
        pk->prod_b.release();
        HIP_CHECK(hipMemGetInfo(&after, &total));
        if (freed_bytes) *freed_bytes = after > before ? after - before : 0;
    });
}

int pai_pubkey_info(const pai_pubkey* pk, int* key_bits, int* n_words, int* ct_words, int* r_words, int* randbits,
                    int* is_djn, int* device) {
    return guarded([&] {
        require(pk != nullptr, "pk is NULL");
        if (key_bits) *key_bits = pk->key_bits;
        if (n_words) *n_words = pk->n_words;
        if (ct_words) *ct_words = pk->ct_words;
        if (r_words) *r_words = pk->r_words;
        if (randbits) *randbits = pk->randbits;
        if (is_djn) *is_djn = pk->djn ? 1 : 0;
        if (device) *device = pk->device;
    });
}

int pai_pubkey_table_info(const pai_pubkey* pk, size_t* table_bytes, int* window_bits, int* windows) {
    return guarded([&] {
        require(pk != nullptr, "pk is NULL");
        std::lock_guard<std::mutex> lk(pk->mu);
        const bool digit = pk->fb_ready && pk->d_fb_dig != nullptr;
        if (table_bytes) *table_bytes = pk->fb_ready ? pk->fb_bytes : 0;
        if (window_bits) *window_bits = !pk->fb_ready ? 0 : (digit ? pk->fbd_wbits : pk->fb_wbits);
        if (windows) *windows = !pk->fb_ready ? 0 : (digit ? pk->fbd_windows : pk->fb_windows);
    });
}

static bool ensure_midp(const pai_pubkey* pk);
// PAI_TUNE enc_mid_min / enc_mid_max: batch range of the lane-group digit-pair DJN encryption at keys the one-element-per-lane engine
// serves (max 0 disables).  Measured (profiles/r05/enc_mid.jsonl): 2048-bit keys 1.1 - 1.26 ms flat up to 16 384 elements, 2.3 ms at
// 32 768, against 1.3 / 2.4 ms of the small-batch kernel at 4 096 / 8 192 and 3.34 ms of the one-element-per-lane engine up to 65 536
// (level at ~3 500 and ~49 000); 1024-bit 0.27 - 0.31 / 0.40 ms against 0.26 - 0.50 / 0.50
static size_t enc_mid_min(size_t ncu) {
    long long v;
    return knob_tune("enc_mid_min", &v) ? (size_t)v : 16 * ncu;
}
static size_t enc_mid_max(size_t ncu, int n_bits) {
    long long v;
    if (knob_tune("enc_mid_max", &v)) return (size_t)v;
    // (the caller also needs the pair geometry's limb count to equal the digit engine's: 1024-class keys and 1537 .. 2048-bit keys)
    return n_bits > 900 && n_bits <= 2048 ? 160 * ncu : 0;
}
static void encrypt_common(const pai_pubkey* pk, const uint32_t* d_m, const uint32_t* d_r, const uint32_t* d_ct_in,
                           uint32_t* d_ct_out, size_t N, void* stream, bool from_plain) {
    DeviceScope scope_(pk->device);
    const GeoOps* g = pk->msq.geo;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(g, N, pk->dev.ncu);
    g_last_times.clear();
    // every path below shares per-key device scratch (quotient-digit columns, window tables): one at a time per
    // handle, ordered across streams by pk->order
    std::lock_guard<std::mutex> lk(pk->mu);
    if (d_r && pk->djn && pk->penc_nl && N >= enc_mid_min((size_t)pk->dev.ncu) && N <= enc_mid_max((size_t)pk->dev.ncu, pk->key_bits) &&
        ensure_midp(pk) && pk->midp_nl == pk->penc_nl) {
        // mid-size DJN batch at a key the one-element-per-lane engine serves: the same fixed-base table (raw [window][digit][2][NL]
        // digit pairs, g-factored or not: the two engines share the layout and R = 2^(29 NL)) read by the lane-group digit-pair
        // kernel with 4 lanes per element (16 elements per wavefront), then w + v n on the n^2 geometry (k_pair_finish)
        build_fb_tables(pk);
        if (pk->d_fb_dig) {
            EncParams P = pk->enc_params();
            PairParams Q;
            Q.nctx = pk->midp_n.d_ctx;
            Q.nm1 = pk->d_midp_nm1;
            Q.fb_table = pk->d_fb_dig;
            Q.fb_windows = pk->fbd_windows;
            Q.fb_wbits = pk->fbd_wbits;
            Q.pt_words = pk->n_words;
            Q.r_words = pk->r_words;
            Q.out_words = pk->midp_out_words;
            Q.fb_gform = pk->fb_gform ? 1 : 0;
            pk->pair_wv.ensure(N * 2 * (size_t)pk->midp_out_words * 4);
            const int epb = pair_epb(pk->midp_nl);
            const size_t tiles = (N + epb - 1) / epb;
            const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu * 8));
            pk->order.begin(s);
            ScopedKernelTimer t(from_plain ? "k_encrypt(djn)" : "k_encrypt(obfuscate)", s);
            if (!launch_pair_fixed_base(pk->midp_nl, s, pgrid, Q, d_m, d_r, pk->pair_wv.as<uint32_t>(), (int)N, from_plain ? 1 : 0))
                throw PaiError(PAI_E_INTERNAL, "no digit-pair kernel for this limb count");
            g->pair_finish(s, grid, P, pk->pair_wv.as<uint32_t>(), pk->midp_out_words, d_ct_in, d_ct_out, (int)N, from_plain ? 0 : 1);
            t.stop();
            HIP_CHECK(hipGetLastError());
            pk->order.end(s);
            return;
        }
    }
    if (d_r && pk->djn && N <= latency_max_elements(LAT_ENC, pk->key_bits)) {
        // small DJN batch: n^2 spread over a wavefront per ciphertext, 10-bit fixed-base windows in that geometry
        if (!pk->lat_ready) {
            pk->lat_ready = true;
            if (const GeoOps* gl = geo_latency_for_bits(hbn::bitlen(pk->nsq))) {
                pk->lat_msq.init(pk->nsq, 0, gl);
                pk->lat_usable = true;
            }
        }
        if (pk->lat_usable) {
            const GeoOps* gl = pk->lat_msq.geo;
            if (!pk->lat_fb_ready) {
                // window width of the small-batch table: every window is one sequential product (~11 us at 2048-bit keys, 6 us on
                // a minus-one context) of the call's latency; 12 bits = 86 windows x 4096 entries (0.2 GB at 2048-bit keys; 10
                // bits: 103 windows, 60 MB; 14 bits: 74 windows, 0.7 GB).  PAI_TUNE lat_fb_wbits pins it (4..16).
                int lw = 12;
                if (long long v; knob_tune("lat_fb_wbits", &v) && v >= 4 && v <= 16) lw = (int)v;
                pk->lat_fb_wbits = lw;
                pk->lat_fb_windows = (pk->randbits + pk->lat_fb_wbits - 1) / pk->lat_fb_wbits;
                pk->lat_fb_ready = true;
            }
            // the four waves of a workgroup share one wave's integers (k_encrypt_tree: a quarter of the windows each, two
            // levels of combining products); PAI_TUNE lat_enc_tree=0 keeps one chain per integer
            const bool tree = gl->t >= 16 && gl->t <= 64 && N <= lat_enc_tree_max((size_t)pk->dev.ncu);
            // ... and on a minus-one context of n^2 where one fits (ensure_lat_ctx): the table is converted once into that
            // context's Montgomery form and the conventional copy is dropped (rebuilt only if PAI_DISABLE=lat_enc_m1 / PAI_TUNE lat_enc_tree ask for it)
            ensure_lat_ctx(pk);
            const bool m1 = tree && pk->lat_m1_ok && !lat_enc_m1_disabled();
            if ((m1 && !pk->d_lat_fb_m1) || (!m1 && !pk->d_lat_fb)) {
                if (!pk->d_lat_fb) pk->d_lat_fb = build_lane_group_fb(pk, pk->lat_msq, pk->lat_fb_wbits, pk->lat_fb_windows);
                if (m1) {
                    const ModSetup& M1 = pk->lat_msq_m1;
                    // c == R'^2 / R_c (mod n^2), R' = 2^(29 rows), R_c = 2^(29 nl): a power of two, negative exponents by halving
                    const int e = hbn::RB * (2 * (int)M1.rows() - pk->lat_msq.nl);
                    Limbs c;
                    if (e >= 0) c = hbn::mod(hbn::shl(Limbs{1u}, e), pk->nsq);
                    else {
                        c = Limbs{1u};
                        for (int i = 0; i < -e; ++i) { if (hbn::is_odd(c)) c = hbn::add(c, pk->nsq); c = hbn::shr(c, 1); }
                    }
                    uint32_t* d_c = upload_r29(c, M1.nl);
                    const size_t NE = (size_t)pk->lat_fb_windows << pk->lat_fb_wbits;
                    hipError_t e0 = hipMalloc((void**)&pk->d_lat_fb_m1, NE * (size_t)M1.nl * 4);
                    if (e0 == hipSuccess) {
                        EncParams PC;
                        PC.nsq = M1.d_ctx;
                        const int gconv = (int)std::max<size_t>(1, std::min<size_t>((NE + gl->epb - 1) / gl->epb, (size_t)pk->dev.ncu * 8));
                        gl->encrypt(nullptr, gconv, PC, pk->d_lat_fb, d_c, nullptr, pk->d_lat_fb_m1, (int)NE, 7);
                        e0 = hipGetLastError();
                        const hipError_t e1 = hipDeviceSynchronize();
                        if (e0 == hipSuccess) e0 = e1;
                    }
                    (void)hipFree(d_c);
                    if (e0 != hipSuccess) {
                        if (pk->d_lat_fb_m1) { (void)hipFree(pk->d_lat_fb_m1); pk->d_lat_fb_m1 = nullptr; }
                        HIP_CHECK(e0);
                    }
                    (void)hipFree(pk->d_lat_fb);
                    pk->d_lat_fb = nullptr;
                    if (!pk->d_lat_nR_m1) pk->d_lat_nR_m1 = upload_r29(hbn::mulmod(pk->n, M1.R, M1.M), M1.nl);
                }
            }
            if (!pk->d_lat_nR) pk->d_lat_nR = upload_r29(hbn::mulmod(pk->n, pk->lat_msq.R, pk->nsq), pk->lat_msq.nl);
            EncParams PL;
            PL.nsq = m1 ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx;
            PL.nR = m1 ? pk->d_lat_nR_m1 : pk->d_lat_nR;
            PL.fb_table = m1 ? pk->d_lat_fb_m1 : pk->d_lat_fb;
            PL.fin = m1 ? pk->lat_msq.d_ctx : nullptr;
            PL.fb_windows = pk->lat_fb_windows;
            PL.fb_wbits = pk->lat_fb_wbits;
            PL.pt_words = pk->n_words;
            PL.ct_words = pk->ct_words;
            PL.r_words = pk->r_words;
            pk->order.begin(s);
            ScopedKernelTimer t(from_plain ? "k_encrypt(djn)" : "k_encrypt(obfuscate)", s);
            const int per_wg = tree ? 64 / gl->t : gl->epb;
            gl->encrypt(s, (int)((N + per_wg - 1) / per_wg), PL, d_m, d_r, d_ct_in, d_ct_out, (int)N, (from_plain ? 1 : 2) + (tree ? 4 : 0));
            t.stop();
            HIP_CHECK(hipGetLastError());
            pk->order.end(s);
            return;
        }
    }
    if (d_r == nullptr && from_plain && N <= 2 * lat_add_max() && ensure_lat_ctx(pk)) {
        // small raw encryptions (the plaintext side of ct + pt): 1 + m n as ONE product with n^2 spread over a wavefront
        // (k_encrypt mode 0 on the latency geometry): 25 against 60 us of kernel time
        if (!pk->d_lat_nR) pk->d_lat_nR = upload_r29(hbn::mulmod(pk->n, pk->lat_msq.R, pk->nsq), pk->lat_msq.nl);
        const GeoOps* gl = pk->lat_msq.geo;
        EncParams PL;
        PL.nsq = pk->lat_msq.d_ctx;
        PL.nR = pk->d_lat_nR;
        PL.fb_table = nullptr;
        PL.fb_windows = 0;
        PL.fb_wbits = 0;
        PL.pt_words = pk->n_words;
        PL.ct_words = pk->ct_words;
        PL.r_words = pk->r_words;
        ScopedKernelTimer t("k_encrypt(raw)", s);
        gl->encrypt(s, (int)((N + gl->epb - 1) / gl->epb), PL, d_m, nullptr, nullptr, d_ct_out, (int)N, 0);
        t.stop();
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (d_r && pk->djn) build_fb_tables(pk);
    EncParams P = pk->enc_params();
    pk->order.begin(s);
    if (pk->penc_nl && ((from_plain && (d_r == nullptr || pk->djn)) || (!from_plain && d_r && pk->djn))) {
        // raw / DJN encryption on the base-n digit engine: one workgroup per CU
        EncPadicParams Q;
        Q.nctx = pk->nmod.d_ctx;
        Q.nm1 = pk->d_nm1;
        Q.nsq = pk->d_nsq29;
        Q.fb_table = reinterpret_cast<const uint4*>(pk->d_fb_dig);
        Q.mscratch = reinterpret_cast<uint4*>(pk->d_mscratch);
        Q.kdig = pk->d_ct_kdig;
        Q.nd = pk->ct_nd;
        Q.fb_windows = pk->fbd_windows;
        Q.fb_wbits = pk->fbd_wbits;
        Q.fb_gform = pk->fb_gform ? 1 : 0;
        Q.pt_words = pk->n_words;
        Q.ct_words = pk->ct_words;
        Q.r_words = pk->r_words;
        const size_t tiles = (N + BLOCK_THREADS - 1) / BLOCK_THREADS;
        const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu));
        ScopedKernelTimer t(!from_plain ? "k_encrypt(obfuscate)" : (d_r ? "k_encrypt(djn)" : "k_encrypt(raw)"), s);
        if (!launch_encrypt_padic(pk->penc_nl, s, pgrid, Q, d_m, d_r, d_ct_in, d_ct_out, (int)N, !from_plain ? 2 : (d_r ? 1 : 0)))
            throw PaiError(PAI_E_INTERNAL, "no digit-engine encrypt kernel for this limb count");
        t.stop();
    } else if (pk->pair_nl && d_r && pk->djn) {
        // DJN encryption / obfuscation on lane-group digit pairs: (w, v) = plain pair of hs^r (1 + m n) [or hs^r], then
        // ct = w + v n [or ct_in (w + v n)] as one product on the n^2 geometry (k_pair_finish)
        PairParams Q;
        Q.nctx = pk->npair.d_ctx;
        Q.nm1 = pk->d_pair_nm1;
        Q.fb_table = pk->d_pair_fb;
        Q.fb_windows = pk->pair_windows;
        Q.fb_wbits = pk->pair_wbits;
        Q.pt_words = pk->n_words;
        Q.r_words = pk->r_words;
        Q.out_words = pk->pair_out_words;
        Q.fb_gform = pk->fb_gform ? 1 : 0;
        pk->pair_wv.ensure(N * 2 * (size_t)pk->pair_out_words * 4);
        const int epb = pair_epb(pk->pair_nl);
        const size_t tiles = (N + epb - 1) / epb;
        const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu * 2));
        ScopedKernelTimer t(from_plain ? "k_encrypt(djn)" : "k_encrypt(obfuscate)", s);
        if (!launch_pair_fixed_base(pk->pair_nl, s, pgrid, Q, d_m, d_r, pk->pair_wv.as<uint32_t>(), (int)N, from_plain ? 1 : 0))
            throw PaiError(PAI_E_INTERNAL, "no digit-pair kernel for this limb count");
        g->pair_finish(s, grid, P, pk->pair_wv.as<uint32_t>(), pk->pair_out_words, d_ct_in, d_ct_out, (int)N, from_plain ? 0 : 1);
        t.stop();
    } else if (d_r == nullptr) {
        require(from_plain, "obfuscation needs randomness");
        ScopedKernelTimer t("k_encrypt(raw)", s);
        g->encrypt(s, grid, P, d_m, nullptr, nullptr, d_ct_out, (int)N, 0);
        t.stop();
    } else if (pk->djn) {
        ScopedKernelTimer t("k_encrypt(djn)", s);
        g->encrypt(s, grid, P, d_m, d_r, d_ct_in, d_ct_out, (int)N, from_plain ? 1 : 2);
        t.stop();
    } else {
        // standard scheme: obf_i = r_i^n mod n^2 (uniform exponent), then one fused multiply
        pk->tmp.ensure(N * (size_t)pk->ct_words * 4);
        if (pk->penc_nl && pk->d_pow_ops) {
            const size_t tiles = (N + BLOCK_THREADS - 1) / BLOCK_THREADS;
            const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu));
            pk->table.ensure(padic_table_words(pk->penc_nl, (size_t)pgrid) * 4);
            PowPadicParams Q;
            Q.nctx = pk->nmod.d_ctx;
            Q.nm1 = pk->d_nm1;
            Q.nsq = pk->d_nsq29;
            Q.kdig = pk->d_ct_kdig;
            Q.ops = pk->d_pow_ops;
            Q.nops = pk->pow_nops;
            Q.tbl_entries = PADIC_TBL_ENTRIES;
            Q.mscratch = reinterpret_cast<uint4*>(pk->d_mscratch);
            Q.table = pk->table.as<uint4>();
            Q.in_words = pk->n_words;
            Q.ct_words = pk->ct_words;
            ScopedKernelTimer t("k_pow(r^n)", s);
            if (!launch_pow_padic(pk->penc_nl, s, pgrid, Q, d_r, pk->tmp.as<uint32_t>(), (int)N))
                throw PaiError(PAI_E_INTERNAL, "no digit-engine power kernel for this limb count");
            t.stop();
        } else {
            pk->table.ensure(g->table_words((size_t)grid) * 4);
            g->modexp_fixed(s, grid, pk->msq.d_ctx, d_r, pk->n_words, pk->d_nexp, pk->n_words, hbn::bitlen(pk->n),
                            pk->tmp.as<uint32_t>(), pk->ct_words, (int)N, pk->table.as<uint32_t>(), 0);
        }
        g->encrypt(s, grid, P, d_m, pk->tmp.as<uint32_t>(), d_ct_in, d_ct_out, (int)N, from_plain ? 3 : 4);
    }
    HIP_CHECK(hipGetLastError());
    pk->order.end(s);
}

// ---- data formats either side of the path (kernels_codec.hpp) ---------------------------------------------
int pai_fp_encode_f64(const pai_pubkey* pk, const double* d_x, size_t N, uint32_t* d_m, int32_t* d_expo, void* stream) {
    return guarded([&] {
        require(pk && d_x && d_m && d_expo, "NULL argument");
        require(hbn::bitlen(pk->n) > 66, "device encode needs a modulus of more than 66 bits");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        hipLaunchKernelGGL(k_fp_encode_f64, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, pk->d_nexp,
                           pk->n_words, d_m, d_expo, N);
        HIP_CHECK(hipGetLastError());
    });
}

int pai_fp_encode_i64(const pai_pubkey* pk, const int64_t* d_x, size_t N, uint32_t* d_m, int32_t* d_expo, void* stream) {
    return guarded([&] {
        require(pk && d_x && d_m && d_expo, "NULL argument");
        require(hbn::bitlen(pk->n) > 66, "device encode needs a modulus of more than 66 bits");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        hipLaunchKernelGGL(k_fp_encode_i64, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, pk->d_nexp,
                           pk->n_words, d_m, d_expo, N);
        HIP_CHECK(hipGetLastError());
    });
}

int pai_fp_encode_at(const pai_pubkey* pk, const void* d_x, int is_f64, size_t N, const int32_t* d_target, int target_bcast,
                     uint32_t* d_m, int32_t* d_expo, void* stream) {
    return guarded([&] {
        require(pk && d_x && d_m && d_expo && d_target, "NULL argument");
        require(hbn::bitlen(pk->n) > 66, "device encode needs a modulus of more than 66 bits");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        const int nbits = hbn::bitlen(pk->n);
        if (is_f64)
            hipLaunchKernelGGL(k_fp_encode_at<true>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, pk->d_nexp,
                               pk->n_words, nbits, d_target, target_bcast, d_m, d_expo, N);
        else
            hipLaunchKernelGGL(k_fp_encode_at<false>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, pk->d_nexp,
                               pk->n_words, nbits, d_target, target_bcast, d_m, d_expo, N);
        HIP_CHECK(hipGetLastError());
    });
}

int pai_fp_decode_i64(const pai_pubkey* pk, const uint32_t* d_m, size_t N, int64_t* d_mant, int32_t* d_flag, void* stream) {
    return guarded([&] {
        require(pk && d_m && d_mant && d_flag, "NULL argument");
        require(hbn::bitlen(pk->n) > 66, "device decode needs a modulus of more than 66 bits");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        hipLaunchKernelGGL(k_fp_decode_i64, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_m, pk->d_nexp,
                           pk->n_words, d_mant, d_flag, N);
        HIP_CHECK(hipGetLastError());
    });
}

int pai_draw_r(const pai_pubkey* pk, const uint32_t* h_key8, const uint32_t* h_nonce3, uint32_t counter0, size_t N,
               uint32_t* d_r, void* stream) {
    return guarded([&] {
        require(pk && h_key8 && h_nonce3 && d_r, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        ChaChaKey K;
        std::memcpy(K.k, h_key8, 32);
        std::memcpy(K.nonce, h_nonce3, 12);
        K.counter0 = counter0;
        const size_t total = N * (size_t)pk->r_words;
        const size_t blocks = (total + 15) / 16;
        // DJN keys: r < 2^randbits.  Standard keys: candidates of bits(n) bits — the caller keeps those in [1, n)
        // (rejection sampling, bindings.py) so that r is uniform there.
        const int rbits = pk->djn ? pk->randbits : hbn::bitlen(pk->n);
        require(rbits >= 1 && rbits <= 32 * pk->r_words, "randomness width does not fit the r rows");
        const int top_word = (rbits - 1) / 32;                  // the live top word; n may be words shorter than key_bits
        const int top = rbits - 32 * top_word;                  // 1 ... 32
        const uint32_t mask = top >= 32 ? 0xFFFFFFFFu : ((1u << top) - 1u);
        hipLaunchKernelGGL(k_draw_r, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, d_r, total,
                           pk->r_words, top_word, mask);
        HIP_CHECK(hipGetLastError());
    });
}

int pai_raw_encrypt(const pai_pubkey* pk, const uint32_t* d_m, size_t N, uint32_t* d_ct, void* stream) {
    return guarded([&] {
        require(pk && d_m && d_ct, "NULL argument");
        if (N == 0) return;
        encrypt_common(pk, d_m, nullptr, nullptr, d_ct, N, stream, true);
    });
}
int pai_encrypt(const pai_pubkey* pk, const uint32_t* d_m, const uint32_t* d_r, size_t N, uint32_t* d_ct,
                void* stream) {
    return guarded([&] {
        require(pk && d_m && d_ct, "NULL argument");
        if (N == 0) return;
        encrypt_common(pk, d_m, d_r, nullptr, d_ct, N, stream, true);
    });
}
int pai_obfuscate(const pai_pubkey* pk, uint32_t* d_ct, const uint32_t* d_r, size_t N, void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_r, "NULL argument");
        if (N == 0) return;
        encrypt_common(pk, nullptr, d_r, d_ct, d_ct, N, stream, false);
    });
}

int pai_ct_add(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N,
               uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_a && d_b && d_out, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        g_last_times.clear();
        const ModSetup* L = lat_add_ctx(pk, N, false, 2);
        const GeoOps* g = L ? L->geo : pk->msq.geo;
        ScopedKernelTimer t("k_modmul", (hipStream_t)stream);
        g->modmul((hipStream_t)stream, L ? (int)((N + g->epb - 1) / g->epb) : grid_for(g, N, pk->dev.ncu), L ? L->d_ctx : pk->msq.d_ctx,
                  d_a, d_b, d_out, (int)N, pk->ct_words, b_bcast,
                  MODMUL_FULL);
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}

// ct^e on the base-n digit engine (k_ctmul_padic); the caller holds pk->mu
static void ctmul_padic_locked(const pai_pubkey* pk, hipStream_t s, const uint32_t* d_ct, const uint32_t* d_e, int e_words,
                               int ebits_max, int e_bcast, size_t N, uint32_t* d_out, int wbits, const char* timer_name) {
    const size_t tiles = (N + BLOCK_THREADS - 1) / BLOCK_THREADS;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu));
    pk->ctmul_table.ensure(ctmul_padic_table_words(pk->penc_nl, wbits, (size_t)grid) * 4);
    CtMulPadicParams Q;
    Q.nctx = pk->nmod.d_ctx;
    Q.nm1 = pk->d_nm1;
    Q.nsq = pk->d_nsq29;
    Q.kdig = pk->d_ct_kdig;
    Q.one_dig = pk->d_one_dig;
    Q.mscratch = reinterpret_cast<uint4*>(pk->d_mscratch);
    Q.table = pk->ctmul_table.as<uint4>();
    Q.nd = pk->ct_nd;
    Q.wbits = wbits;
    Q.ct_words = pk->ct_words;
    Q.e_words = e_words;
    Q.ebits_max = ebits_max;
    Q.e_bcast = e_bcast;
    pk->order.begin(s);
    ScopedKernelTimer t(timer_name, s);
    if (!launch_ctmul_padic(pk->penc_nl, s, grid, Q, d_ct, d_e, d_out, (int)N))
        throw PaiError(PAI_E_INTERNAL, "no digit-engine ct*pt kernel for this limb count");
    t.stop();
    HIP_CHECK(hipGetLastError());
    pk->order.end(s);
}

// ct^e on lane-group digit pairs (k_pair_ctmul, then w + v n on the n^2 geometry: k_pair_finish); the caller holds pk->mu
static void ctmul_pair_locked(const pai_pubkey* pk, hipStream_t s, int nl, const MontCtx* nctx, const uint32_t* nm1, const uint32_t* kdig,
                              const uint32_t* one, int nd, int out_words, const uint32_t* d_ct, const uint32_t* d_e, int e_words,
                              int ebits_max, int e_bcast, size_t N, uint32_t* d_out) {
    const GeoOps* g = pk->msq.geo;
    const int grid = grid_for(g, N, pk->dev.ncu);
    const int wbits = var_window_bits(ebits_max);
    const int epb = pair_epb(nl);
    const size_t tiles = (N + epb - 1) / epb;
    const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu * (nl <= 72 ? 8 : 2)));
    pk->pair_ct_table.ensure(((size_t)pgrid * epb << wbits) * 2 * (size_t)nl * 4);
    pk->pair_wv.ensure(N * 2 * (size_t)out_words * 4);
    PairCtMulParams Q;
    Q.nctx = nctx;
    Q.nm1 = nm1;
    Q.kdig = kdig;
    Q.one_pair = one;
    Q.table = pk->pair_ct_table.as<uint32_t>();
    Q.nd = nd;
    Q.wbits = wbits;
    Q.ct_words = pk->ct_words;
    Q.e_words = e_words;
    Q.ebits_max = ebits_max;
    Q.e_bcast = e_bcast;
    Q.out_words = out_words;
    EncParams P;
    P.nsq = pk->msq.d_ctx;
    P.nR = pk->d_nR;
    P.fb_table = nullptr;
    P.fb_windows = 0;
    P.fb_wbits = 0;
    P.pt_words = pk->n_words;
    P.ct_words = pk->ct_words;
    P.r_words = pk->r_words;
    OrderScope order_(pk->order, s);
    ScopedKernelTimer t("k_ctmul", s);
    if (!launch_pair_ctmul(nl, s, pgrid, Q, d_ct, d_e, pk->pair_wv.as<uint32_t>(), (int)N))
        throw PaiError(PAI_E_INTERNAL, "no digit-pair ct * pt kernel for this limb count");
    g->pair_finish(s, grid, P, pk->pair_wv.as<uint32_t>(), out_words, nullptr, d_out, (int)N, 0);
    t.stop();
    HIP_CHECK(hipGetLastError());
}
// constants of the same for n of a key the one-element-per-lane engine serves (mid-size batches); the caller holds pk->mu
static bool ensure_midp(const pai_pubkey* pk) {
    if (pk->midp_tried) return pk->midp_ok;
    pk->midp_tried = true;
    const int nbits = hbn::bitlen(pk->n);
    const int nl = pair_nl_for_prime_bits(nbits);           // (the 4-lane geometries of the primes serve an n of the same size)
    if (!nl || knob_disabled("pair") || !pk->d_nR) return false;
    pk->midp_nl = nl;
    pk->midp_n.init(pk->n, nl);
    pk->d_midp_nm1 = upload_r29(hbn::sub(pk->n, Limbs{1u}), nl);
    pk->midp_out_words = (hbn::RB * nl + 31) / 32;
    pk->midp_nd = (32 * pk->ct_words + hbn::RB * nl - 1) / (hbn::RB * nl);
    auto pair_of = [&](const Limbs& v, std::vector<uint32_t>& dst) {
        Limbs rem;
        Limbs quo = hbn::divq(v, pk->n, &rem);
        auto ra = hbn::to_r29(rem, nl), rb = hbn::to_r29(quo, nl);
        dst.insert(dst.end(), ra.begin(), ra.end());
        dst.insert(dst.end(), rb.begin(), rb.end());
    };
    const Limbs Rm = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * nl), pk->nsq);
    std::vector<uint32_t> kd, one;
    Limbs K = hbn::mulmod(Rm, Rm, pk->nsq);
    for (int i = 0; i < pk->midp_nd; ++i) {
        pair_of(K, kd);
        K = hbn::mulmod(K, Rm, pk->nsq);
    }
    pair_of(Rm, one);
    pk->d_midp_kdig = upload_vec(kd);
    pk->d_midp_one = upload_vec(one);
    pk->midp_ok = true;
    return true;
}
// PAI_TUNE ctmul_mid_min / ctmul_mid_max: batch range of it (max 0 disables).  Measured with 53-bit exponents
// (profiles/r05/ctmul_mid.jsonl): 2048-bit keys 1.4 - 1.5 ms flat up to 16 384 ciphertexts (one wave of 16 per SIMD), 2.9 ms at 32 768,
// against 2.2 / 4.1 ms of the small-batch kernels at 8 192 / 16 384 and 4.7 ms of the one-element-per-lane engine up to 65 536
// (behind below ~5 000 and from ~55 000); 1024-bit keys 0.55 - 0.63 / 0.9 ms against 0.84 - 1.25 / 1.27
static bool mid_band(int n_bits) {                   // the key sizes the 4-lane geometries are cut for (measured); others take the next wider one
    return (n_bits > 900 && n_bits <= 1024) || (n_bits > 1400 && n_bits <= 1536) || (n_bits > 1900 && n_bits <= 2048);
}
static size_t ctmul_mid_min(size_t ncu, int n_bits) {
    long long v;
    if (knob_tune("ctmul_mid_min", &v)) return (size_t)v;
    return (mid_band(n_bits) ? 20 : 28) * ncu;       // (1280- / 1792-bit keys: level near 7 000 / 5 500 ciphertexts)
}
static size_t ctmul_mid_max(size_t ncu, int n_bits) {
    long long v;
    if (knob_tune("ctmul_mid_max", &v)) return (size_t)v;
    return n_bits > 900 && n_bits <= 2048 ? 192 * ncu : 0;
}

int pai_ct_mul(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_e, int e_words, int ebits_max,
               int e_bcast, size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_e && d_out, "NULL argument");
        require(e_words > 0 && ebits_max > 0 && ebits_max <= 32 * e_words, "bad exponent shape");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        hipStream_t s = (hipStream_t)stream;
        g_last_times.clear();
        if (!pk->pair_nl && ebits_max > 8 && N >= ctmul_mid_min((size_t)pk->dev.ncu, pk->key_bits) &&
            N <= ctmul_mid_max((size_t)pk->dev.ncu, pk->key_bits)) {
            std::lock_guard<std::mutex> lk(pk->mu);
            if (ensure_midp(pk)) {
                ctmul_pair_locked(pk, s, pk->midp_nl, pk->midp_n.d_ctx, pk->d_midp_nm1, pk->d_midp_kdig, pk->d_midp_one, pk->midp_nd,
                                  pk->midp_out_words, d_ct, d_e, e_words, ebits_max, e_bcast, N, d_out);
                return;
            }
        }
        if (N <= latency_max_elements(LAT_MUL, pk->key_bits) && ebits_max > 8) {
            // small batch: windowed exponentiation with n^2 spread over a whole wavefront per ciphertext
            std::lock_guard<std::mutex> lk(pk->mu);
            if (ensure_lat_ctx(pk) && pk->lat_pp_ok && e_words <= PP_EWORDS && N <= lat_mul_pp_max((size_t)pk->dev.ncu)) {
                // smallest batches: digit pairs with base n' = n k, the chain pipelined over the four waves of a workgroup per
                // ciphertext (kernels_declat.hpp)
                DecPPParams Q{};
                Q.pp[0] = pk->lat_pp.d_ctx;
                Q.kdig[0] = pk->d_lat_pp_kdig;
                Q.kx[0] = pk->d_lat_pp_kx;
                Q.sq[0] = pk->lat_msq_m1.d_ctx;
                Q.fin[0] = pk->lat_msq.d_ctx;
                Q.expo[0] = d_e;
                Q.ebits[0] = ebits_max;
                Q.nd = pk->lat_pp_nd;
                Q.nch = pk->lat_pp_nch;
                Q.ct_words = pk->ct_words;
                Q.u_words = pk->ct_words;
                Q.e_words = e_words;
                Q.e_bcast = e_bcast;
                pk->order.begin(s);
                ScopedKernelTimer t("k_ctmul", s);
                launch_ctmul_pp(s, (int)N, Q, d_ct, d_out, pk->lat_pp_chain);
                t.stop();
                HIP_CHECK(hipGetLastError());
                pk->order.end(s);
                return;
            }
            if (ensure_lat_ctx(pk)) {
                const GeoOps* g = pk->lat_msq.geo;
                // right to left on wave pairs (k_modexp_rl: squarings on one wave, products on another, no table) for the
                // smallest batches; needs the minus-one context
                const bool rl = pk->lat_m1_ok && g->epb >= 2 && N <= lat_mul_rl_max((size_t)pk->dev.ncu);
                const int per_wg = rl ? g->epb / 2 : g->epb;
                const int grid = (int)((N + per_wg - 1) / per_wg);
                const int wbits = rl ? 0 : var_window_bits(ebits_max);
                if (!rl) pk->lat_table.ensure(((size_t)1 << wbits) * g->nl * (size_t)grid * g->epb * 4);
                pk->order.begin(s);
                ScopedKernelTimer t("k_ctmul", s);
                g->modexp_var_win(s, grid, pk->lat_m1_ok ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx, d_ct, pk->ct_words, d_e, e_words,
                                  ebits_max, e_bcast, d_out, pk->ct_words, (int)N, pk->lat_table.as<uint32_t>(), wbits,
                                  pk->lat_m1_ok ? pk->lat_msq.d_ctx : nullptr);
                t.stop();
                HIP_CHECK(hipGetLastError());
                pk->order.end(s);
                return;
            }
        }
        if (pk->penc_nl) {
            // base-n digit engine: fixed windows sized to the exponent width (table build 2^w - 2 products, then
            // w squarings + 1 product per window)
            std::lock_guard<std::mutex> lk(pk->mu);
            ctmul_padic_locked(pk, s, d_ct, d_e, e_words, ebits_max, e_bcast, N, d_out, var_window_bits(ebits_max), "k_ctmul");
            return;
        }
        const GeoOps* g = pk->msq.geo;
        const int grid = grid_for(g, N, pk->dev.ncu);
        if (pk->pair_nl && pk->d_pair_kdig && ebits_max > 8 && !pair_ctmul_disabled()) {
            // n of 2049 .. 4156 bits: squarings at 4 NL^2 and multiplications at 5 NL^2 limb products on lane-group digit
            // pairs (k_pair_ctmul) instead of 8 NL^2 per Montgomery product modulo n^2, then w + v n (k_pair_finish)
            std::lock_guard<std::mutex> lk(pk->mu);
            ctmul_pair_locked(pk, s, pk->pair_nl, pk->npair.d_ctx, pk->d_pair_nm1, pk->d_pair_kdig, pk->d_pair_one, pk->pair_nd,
                              pk->pair_out_words, d_ct, d_e, e_words, ebits_max, e_bcast, N, d_out);
            return;
        }
        if (ebits_max > 8) {
            std::lock_guard<std::mutex> lk(pk->mu);
            const int wbits = var_window_bits(ebits_max);
            pk->ctmul_table.ensure(((size_t)1 << wbits) * g->nl * (size_t)grid * g->epb * 4);
            pk->order.begin(s);
            ScopedKernelTimer t("k_ctmul", s);
            g->modexp_var_win(s, grid, pk->msq.d_ctx, d_ct, pk->ct_words, d_e, e_words, ebits_max, e_bcast, d_out,
                              pk->ct_words, (int)N, pk->ctmul_table.as<uint32_t>(), wbits, nullptr);
            t.stop();
            HIP_CHECK(hipGetLastError());
            pk->order.end(s);
            return;
        }
        g->modexp_var(s, grid, pk->msq.d_ctx, d_ct, pk->ct_words, 0, d_e, e_words,
                      ebits_max, e_bcast, d_out, pk->ct_words, (int)N, 0, 0);
        HIP_CHECK(hipGetLastError());
    });
}

static int* status_word(const pai_pubkey* pk, hipStream_t s) {      // under pk->mu
    if (!pk->status.p) {
        pk->status.ensure(4);
        HIP_CHECK(hipMemsetAsync(pk->status.p, 0, 4, s));
        HIP_CHECK(hipStreamSynchronize(s));                          // once per handle: other streams may use it next
    }
    return pk->status.as<int>();
}

static int ct_pow2_impl(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N, int dmax_hint,
                        void* stream);

int pai_ct_pow2(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N,
                void* stream) {
    return ct_pow2_impl(pk, d_ct, d_delta, delta_bcast, N, -1, stream);
}

int pai_ct_pow2_hint(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N, int max_delta,
                     void* stream) {
    return ct_pow2_impl(pk, d_ct, d_delta, delta_bcast, N, max_delta < 0 ? 0 : max_delta, stream);
}

static int ct_pow2_impl(const pai_pubkey* pk, uint32_t* d_ct, const int32_t* d_delta, int delta_bcast, size_t N, int dmax_hint,
                        void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_delta, "NULL argument");
        if (N == 0) return;
        if (dmax_hint == 0) return;                                       // the caller knows that nothing is to be raised
        DeviceScope scope_(pk->device);
        const GeoOps* g = pk->msq.geo;
        g_last_times.clear();
        if (pk->penc_nl && N >= pow2_digit_min_elements()) {
            // Large batches on keys the digit engine serves: ct^(2^delta) is ct * pt with the one-bit exponent 2^delta —
            // delta squarings at 4 NL^2 limb products on base-n digit pairs (+ ~4 products of conversions) against
            // delta + 2 products of 8 NL^2 on the lane-group engine.  Worth it from shifts of ~8 on (ct - ct aligns by
            // up to 52: 88 -> ~55 ms per 2^20); the largest shift decides — the caller's hint (pai_ct_pow2_hint: fully
            // asynchronous) or, without one, a 4-byte read-back that synchronises the stream; smaller shifts keep the
            // lane-group kernel.
            hipStream_t s = (hipStream_t)stream;
            std::unique_lock<std::mutex> lk(pk->mu);
            pk->pow2_expo.ensure(N * 8 + 16);
            int* d_max = reinterpret_cast<int*>(pk->pow2_expo.as<uint32_t>() + 2 * N);
            pk->order.begin(s);
            HIP_CHECK(hipMemsetAsync(d_max, 0, sizeof(int), s));
            // an under-estimated hint only matters where the digit path will run on it (it would truncate 2^delta): with a
            // hint outside that range the lane-group kernel below serves any shift correctly and nothing is flagged
            const bool hint_digit = dmax_hint >= POW2_DIGIT_MIN_SHIFT && dmax_hint <= 62;
            hipLaunchKernelGGL(k_pow2_expo, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_delta, delta_bcast, N,
                               pk->pow2_expo.as<uint32_t>(), d_max, hint_digit ? dmax_hint : -1, hint_digit ? status_word(pk, s) : nullptr);
            HIP_CHECK(hipGetLastError());
            int dmax = dmax_hint;
            if (dmax_hint < 0) {                                          // no hint: read the largest shift back (synchronises)
                HIP_CHECK(hipMemcpyAsync(&dmax, d_max, sizeof(int), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
            }
            pk->order.end(s);
            if (dmax >= POW2_DIGIT_MIN_SHIFT && dmax <= 62) {
                ctmul_padic_locked(pk, s, d_ct, pk->pow2_expo.as<uint32_t>(), 2, dmax + 1, 0, N, d_ct, 1, "k_pow2");
                return;
            }
            if (dmax == 0) return;
        }
        ScopedKernelTimer t("k_pow2", (hipStream_t)stream);
        if (const ModSetup* L = lat_add_ctx(pk, N, false, 4)) {           // small batches: an integer per wavefront (as the aligned additions)
            const GeoOps* gl = L->geo;
            gl->pow2((hipStream_t)stream, (int)((N + gl->epb - 1) / gl->epb), L->d_ctx, d_ct, d_delta, delta_bcast, (int)N, pk->ct_words);
            t.stop();
            HIP_CHECK(hipGetLastError());
            return;
        }
        g->pow2((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->msq.d_ctx, d_ct, d_delta, delta_bcast, (int)N, pk->ct_words);
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}

static void add_aligned_common(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                               size_t N, uint32_t* d_out, const uint32_t* d_entry, void* stream) {
    require(pk && d_a && d_b && d_delta && d_out, "NULL argument");
    if (N == 0) return;
    DeviceScope scope_(pk->device);
    // wire-form operands (no entry constant): small batches on the latency geometry — the kernel enters and leaves the
    // Montgomery domain itself, so the geometry's R does not show in the result
    const ModSetup* L = d_entry == nullptr ? lat_add_ctx(pk, N, false, 4) : nullptr;
    const GeoOps* g = L ? L->geo : pk->msq.geo;
    g_last_times.clear();
    ScopedKernelTimer t("k_add_aligned", (hipStream_t)stream);
    g->add_aligned((hipStream_t)stream, L ? (int)((N + g->epb - 1) / g->epb) : grid_for(g, N, pk->dev.ncu), L ? L->d_ctx : pk->msq.d_ctx,
                   d_a, d_b, b_bcast, d_delta, d_out, (int)N,
                   pk->ct_words, d_entry);
    t.stop();
    HIP_CHECK(hipGetLastError());
}

int pai_ct_add_aligned(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                       size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] { add_aligned_common(pk, d_a, d_b, b_bcast, d_delta, N, d_out, nullptr, stream); });
}

int pai_ct_add_aligned_dom(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                           size_t N, uint32_t* d_out, const uint32_t* d_entry, void* stream) {
    return guarded([&] {
        require(d_entry != nullptr, "NULL entry constant");
        add_aligned_common(pk, d_a, d_b, b_bcast, d_delta, N, d_out, d_entry, stream);
    });
}

int pai_ct_mont_mul(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N, uint32_t* d_out,
                    void* stream) {
    return guarded([&] {
        require(pk && d_a && d_b && d_out, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        g_last_times.clear();
        if (const ModSetup* L = lat_add_ctx(pk, N, true)) {
            const GeoOps* gl = L->geo;
            ScopedKernelTimer t("k_modmul", (hipStream_t)stream);
            gl->modmul((hipStream_t)stream, (int)((N + gl->epb - 1) / gl->epb), L->d_ctx, d_a, d_b, d_out, (int)N, pk->ct_words, b_bcast,
                       MODMUL_FULL);
            t.stop();
            HIP_CHECK(hipGetLastError());
            return;
        }
        const GeoOps* g = pk->msq.geo;
        ScopedKernelTimer t("k_modmul", (hipStream_t)stream);
        g->modmul((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->msq.d_ctx, d_a, d_b, d_out, (int)N, pk->ct_words, b_bcast,
                  MODMUL_MONT);
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}

// table of R^m mod n^2, |m| <= RPOW_SPAN, in limb form (caller holds pk->mu)
static const uint32_t* rpow_table(const pai_pubkey* pk) {
    if (pk->d_rpow) return pk->d_rpow;
    const int nl = pk->msq.nl;
    const Limbs one{1u};
    hbn::Mont32 mt(pk->nsq);
    const Limbs inv2 = hbn::shr(hbn::add(pk->nsq, one), 1);
    const Limbs rinv = mt.powmod(inv2, hbn::from_u64((uint64_t)hbn::RB * (uint64_t)nl));
    require(hbn::cmp(hbn::mulmod(rinv, pk->msq.R, pk->nsq), one) == 0, "R^-1 check failed");
    std::vector<uint32_t> h((size_t)(2 * RPOW_SPAN + 1) * nl, 0);
    auto put = [&](int m, const Limbs& v) {
        const std::vector<uint32_t> r = hbn::to_r29(v, nl);
        std::memcpy(&h[(size_t)(RPOW_SPAN + m) * nl], r.data(), (size_t)nl * 4);
    };
    Limbs up = one, dn = one;
    put(0, one);
    for (int m = 1; m <= RPOW_SPAN; ++m) {
        up = hbn::mulmod(up, pk->msq.R, pk->nsq);
        dn = hbn::mulmod(dn, rinv, pk->nsq);
        put(m, up);
        put(-m, dn);
    }
    pk->d_rpow = upload_vec(h);
    return pk->d_rpow;
}

int pai_ct_addn(const pai_pubkey* pk, const uint32_t* const* h_ops, const int32_t* const* h_raise, int k, int tag0, int tag,
                int dom_out, size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && h_ops && d_out, "NULL argument");
        require(k >= 2 && k <= ADDN_MAX, "pai_ct_addn: between 2 and 16 operands per call");
        for (int j = 0; j < k; ++j) require(h_ops[j] != nullptr, "pai_ct_addn: NULL operand");
        require(!(h_raise && h_raise[0]) || tag0 == tag, "pai_ct_addn: a raised first operand must share the others' domain tag");
        // every domain tag a tile can pass through must have its fix-up constant R^(1 + dom_out - c) in the table
        const int c_lo = std::min(tag0, 1) + (k - 1) * std::min(tag - 1, 0), c_hi = std::max(tag0, 1) + (k - 1) * std::max(tag - 1, 0);
        require(std::abs(2 - tag) <= RPOW_SPAN && std::abs(1 + dom_out - c_lo) <= RPOW_SPAN && std::abs(1 + dom_out - c_hi) <= RPOW_SPAN,
                "pai_ct_addn: domain tags out of range");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        const uint32_t* rpow;
        {
            std::lock_guard<std::mutex> lk(pk->mu);
            rpow = rpow_table(pk);
        }
        AddnArgs A{};
        for (int j = 0; j < k; ++j) { A.op[j] = h_ops[j]; A.raise[j] = h_raise ? h_raise[j] : nullptr; }
        A.k = k; A.tag0 = tag0; A.tag = tag; A.dom_out = dom_out;
        const GeoOps* g = pk->msq.geo;
        g_last_times.clear();
        ScopedKernelTimer t("k_addn", (hipStream_t)stream);
        g->addn((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->msq.d_ctx, A, d_out, (int)N, pk->ct_words, rpow);
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}

int pai_pubkey_mont_bits(const pai_pubkey* pk, int* bits) {
    return guarded([&] {
        require(pk && bits, "NULL argument");
        *bits = RB * pk->msq.geo->nl;
    });
}

// one level of a product tree on single Montgomery products: out[i] = a[i] * b[i] * R^-1 mod n^2
static void tree_mul(const pai_pubkey* pk, hipStream_t s, const uint32_t* a, const uint32_t* b, int b_bcast, uint32_t* out, size_t n) {
    if (n == 0) return;
    const GeoOps* g = pk->msq.geo;
    g->modmul(s, grid_for(g, n, pk->dev.ncu), pk->msq.d_ctx, a, b, out, (int)n, pk->ct_words, b_bcast, MODMUL_MONT);
    HIP_CHECK(hipGetLastError());
}

// the caller holds pk->mu and has selected the device
static void ct_prod_locked(const pai_pubkey* pk, hipStream_t s, const uint32_t* d_ct, size_t count, size_t groups, uint32_t* d_out,
                           bool clear_times = true) {
    {
        const size_t W = (size_t)pk->ct_words, ROW = W * 4;
        size_t members = count / groups;
        if (members == 1) {
            if (d_out != d_ct) HIP_CHECK(hipMemcpyAsync(d_out, d_ct, count * ROW, hipMemcpyDeviceToDevice, s));
            return;
        }
        // Product over halves, member-major rows [member][group]: level k + 1 has h = ceil(members / 2) members,
        // P[i] = X[i] * X[i + h] for i < members - h (one k_modmul launch over (members - h) * groups contiguous rows);
        // a member without a partner is multiplied by tree_c[k] so that the whole level shares the form R^(1 - 2^(k+1)).
        const size_t h0 = (members + 1) / 2;
        pk->prod_a.ensure(h0 * groups * ROW);
        pk->prod_b.ensure(((h0 + 1) / 2) * groups * ROW);
        pk->order.begin(s);
        if (clear_times) g_last_times.clear();
        ScopedKernelTimer t("k_modmul(tree)", s);
        const uint32_t* src = d_ct;
        uint32_t* bufs[2] = {pk->prod_a.as<uint32_t>(), pk->prod_b.as<uint32_t>()};
        int level = 0;
        while (members > 1) {
            require(level < pai_pubkey::TREE_LEVELS, "ct_prod: too many levels");
            const size_t h = (members + 1) / 2, lo = members - h;
            uint32_t* dst = bufs[level & 1];
            tree_mul(pk, s, src, src + h * groups * W, 0, dst, lo * groups);
            if (lo < h) tree_mul(pk, s, src + lo * groups * W, pk->d_tree_c + (size_t)level * W, 1, dst + lo * groups * W, groups);
            src = dst;
            members = h;
            ++level;
        }
        tree_mul(pk, s, src, pk->d_tree_fix + (size_t)level * W, 1, d_out, groups);     // R^(1 - 2^L) * R^(2^L) * R^-1 = 1
        t.stop();
        pk->order.end(s);
    }
}

int pai_ct_prod(const pai_pubkey* pk, const uint32_t* d_ct, size_t count, size_t groups, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_out, "NULL argument");
        require(groups > 0 && count >= groups && count % groups == 0, "count must be a positive multiple of groups");
        std::lock_guard<std::mutex> lk(pk->mu);
        DeviceScope scope_(pk->device);
        ct_prod_locked(pk, (hipStream_t)stream, d_ct, count, groups, d_out);
    });
}

// Multi-exponentiation behind the matrix products (kernels_padic_enc.hpp: k_mexp_table_padic, k_mexp_padic)
int pai_ct_multiexp(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_ct_inv, size_t R, size_t K, size_t M,
                    const uint32_t* d_e, int e_words, int ebits_max, const uint8_t* d_sign, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_e && d_out, "NULL argument");
        require(R > 0 && K > 0 && M > 0 && e_words > 0 && ebits_max > 0 && ebits_max <= 32 * e_words, "bad shape");
        require((d_sign == nullptr) == (d_ct_inv == nullptr), "signs and inverses come together");
        const size_t G = R * M, bases = R * K;
        if (G * K >= ((size_t)1 << 31) || bases >= ((size_t)1 << 28)) throw PaiError(PAI_E_UNSUPPORTED, "matrix product too large for one call");
        std::lock_guard<std::mutex> lk(pk->mu);
        DeviceScope scope_(pk->device);
        hipStream_t s = (hipStream_t)stream;
        const GeoOps* lg = pk->msq.geo;                   // lane-group engine when the digit engine does not serve the key
        const bool digit = pk->penc_nl != 0;
        const int nsigns = d_sign ? 2 : 1;
        // words of one table entry: a digit pair of 2 pnl limbs, or a Montgomery residue of nl limbs
        const int pnl = digit ? pk->penc_nl : (lg->nl + 1) / 2;
        const int lanes_per_wg = digit ? BLOCK_THREADS : lg->epb;
        // members per lane: enough lanes to fill the device (one workgroup of 256 lanes per CU, several rounds), the
        // rest of the sharing goes into longer chunks (the squarings are shared by a chunk)
        size_t want_lanes = (size_t)pk->dev.ncu * lanes_per_wg * (digit ? 2 : 4);
        if (long long v; knob_tune("mexp_lanes", &v) && v > 0) want_lanes = (size_t)v;
        size_t chunk = std::max<size_t>(1, (G * K + want_lanes - 1) / want_lanes);
        chunk = std::min(chunk, K);
        const size_t chunks = (K + chunk - 1) / chunk;
        const size_t nlanes = chunks * G;
        size_t mem_free = 0, mem_total = 0;
        HIP_CHECK(hipMemGetInfo(&mem_free, &mem_total));
        // window width: a term costs ebits / w table products and every base (2^w - 2) products per sign for its table,
        // which M output columns share; the widest tables must fit 1/16 of the device memory
        int wbits = 2;
        {
            double best = 1e300;
            for (int w = 2; w <= 7; ++w) {
                const double tb = (double)bases * nsigns * (double)((size_t)1 << w) * 2.0 * pnl * 4.0;
                if (w > 2 && tb > (double)mem_total / 16.0) break;
                const double cost = (double)((ebits_max + w - 1) / w) + (double)nsigns * (double)(((size_t)1 << w) - 2) / (double)M;
                if (cost < best) { best = cost; wbits = w; }
            }
            if (long long v; knob_tune("mexp_wbits", &v) && v >= 1 && v <= 8) wbits = (int)v;
        }
        const size_t table_bytes = bases * nsigns * ((size_t)1 << wbits) * 2 * (size_t)pnl * 4;
        if (table_bytes > mem_total / 8 || table_bytes + nlanes * (size_t)pk->ct_words * 4 > mem_free + pk->mexp_table.bytes + pk->mexp_partial.bytes)
            throw PaiError(PAI_E_UNSUPPORTED, "power tables of this matrix product do not fit the device");
        pk->mexp_table.ensure(table_bytes);
        pk->mexp_partial.ensure(nlanes * (size_t)pk->ct_words * 4);
        g_last_times.clear();
        OrderScope order_(pk->order, s);
        if (digit) {
            MexpPadicParams Q;
            Q.nctx = pk->nmod.d_ctx;
            Q.nm1 = pk->d_nm1;
            Q.nsq = pk->d_nsq29;
            Q.kdig = pk->d_ct_kdig;
            Q.one_dig = pk->d_one_dig;
            Q.mscratch = reinterpret_cast<uint4*>(pk->d_mscratch);
            Q.table = pk->mexp_table.as<uint4>();
            Q.nd = pk->ct_nd;
            Q.ct_words = pk->ct_words;
            Q.R = (int)R; Q.K = (int)K; Q.M = (int)M; Q.chunk = (int)chunk; Q.nsigns = nsigns;
            Q.e_words = e_words;
            Q.ebits_max = ebits_max;
            Q.by_rows = 0;
            Q.wbits = wbits;
            if (long long v; knob_tune("mexp_by_rows", &v)) Q.by_rows = v != 0;
            {
                const size_t tl = bases * nsigns, tiles = (tl + BLOCK_THREADS - 1) / BLOCK_THREADS;
                const int grid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu));
                ScopedKernelTimer t("k_mexp_table", s);
                if (!launch_mexp_table_padic(pnl, s, grid, Q, d_ct, d_ct_inv, (int)tl))
                    throw PaiError(PAI_E_INTERNAL, "no multi-exponentiation kernel for this limb count");
                t.stop();
                HIP_CHECK(hipGetLastError());
            }
            {
                const size_t tiles = (nlanes + BLOCK_THREADS - 1) / BLOCK_THREADS;
                const int grid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)pk->dev.ncu));
                ScopedKernelTimer t("k_mexp", s);
                if (!launch_mexp_padic(pnl, s, grid, Q, d_e, d_sign, pk->mexp_partial.as<uint32_t>(), (int)nlanes))
                    throw PaiError(PAI_E_INTERNAL, "no multi-exponentiation kernel for this limb count");
                t.stop();
                HIP_CHECK(hipGetLastError());
            }
        } else {
            // lane-group engine (keys above 2048 bits): Montgomery residues modulo n^2 as table entries
            MexpParams P;
            P.R = (int)R; P.K = (int)K; P.M = (int)M; P.chunk = (int)chunk; P.nsigns = nsigns;
            P.e_words = e_words; P.ebits_max = ebits_max; P.wbits = wbits; P.w32 = pk->ct_words;
            {
                const size_t tl = bases * nsigns;
                ScopedKernelTimer t("k_mexp_table", s);
                lg->mexp_table(s, grid_for(lg, tl, pk->dev.ncu), pk->msq.d_ctx, d_ct, d_ct_inv, pk->ct_words, pk->mexp_table.as<uint32_t>(),
                               (int)tl, nsigns, wbits);
                t.stop();
                HIP_CHECK(hipGetLastError());
            }
            {
                ScopedKernelTimer t("k_mexp", s);
                lg->mexp(s, grid_for(lg, nlanes, pk->dev.ncu), pk->msq.d_ctx, P, pk->mexp_table.as<uint32_t>(), d_e, d_sign,
                         pk->mexp_partial.as<uint32_t>(), (int)nlanes);
                t.stop();
                HIP_CHECK(hipGetLastError());
            }
        }
        order_.done();
        ct_prod_locked(pk, s, pk->mexp_partial.as<uint32_t>(), nlanes, G, d_out, false);
    });
}

static int ct_invert_impl(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream, bool sync, int* d_flag);
int pai_ct_invert(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream) {
    return ct_invert_impl(pk, d_ct, N, d_out, stream, true, nullptr);
}
int pai_ct_invert_async(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream) {
    return ct_invert_impl(pk, d_ct, N, d_out, stream, false, nullptr);
}
int pai_ct_invert_flag(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, int32_t* d_flag, void* stream) {
    if (!d_flag) return guarded([&] { require(false, "NULL argument"); });
    return ct_invert_impl(pk, d_ct, N, d_out, stream, false, d_flag);
}
int pai_pubkey_status(const pai_pubkey* pk, int* status_out, int clear, void* stream) {
    return guarded([&] {
        require(pk && status_out, "NULL argument");
        std::lock_guard<std::mutex> lk(pk->mu);
        DeviceScope scope_(pk->device);
        hipStream_t s = (hipStream_t)stream;
        int* w = status_word(pk, s);
        int v = 0;
        HIP_CHECK(hipMemcpyAsync(&v, w, 4, hipMemcpyDeviceToHost, s));
        if (clear) HIP_CHECK(hipMemsetAsync(w, 0, 4, s));
        HIP_CHECK(hipStreamSynchronize(s));
        *status_out = v;
    });
}

static int ct_invert_impl(const pai_pubkey* pk, const uint32_t* d_ct, size_t N, uint32_t* d_out, void* stream, bool sync, int* d_flag) {
    return guarded([&] {
        require(pk && d_ct && d_out, "NULL argument");
        if (N == 0) return;
        std::lock_guard<std::mutex> lk(pk->mu);
        DeviceScope scope_(pk->device);
        hipStream_t s = (hipStream_t)stream;
        const size_t W = (size_t)pk->ct_words, ROW = W * 4;
        // product tree over halves (kernels_invert.hpp): level k + 1 has ceil(count_k / 2) products; the tree stops
        // at <= `top` values, each inverted by one wave's extended GCD.  Every tree product is ONE Montgomery
        // product: level k holds (true value) * R^(1 - 2^k) on the way up (a value without a partner is brought to
        // its level's form by the constant tree_c[k]); the extended GCD inverts the stored top values, and on the
        // way down level k holds (true inverse) * R^(2^k - 1) — the powers of R telescope, so the leaves come out as
        // plain canonical inverses.
        size_t top = 64;
        if (long long v; knob_tune("invert_chunk", &v) && v >= 1 && v <= 65536) top = (size_t)v;     // test hook: where the tree stops
        std::vector<size_t> cnt{N};
        while (cnt.back() > top) cnt.push_back((cnt.back() + 1) / 2);
        const int L = (int)cnt.size() - 1;
        require(L < pai_pubkey::TREE_LEVELS, "ct_invert: too many levels");
        size_t upper = 0;                                              // rows of all levels above the leaves
        std::vector<size_t> off(L + 1, 0);
        for (int k = 1; k <= L; ++k) { off[k] = upper; upper += cnt[k]; }
        const bool alias = (d_out == d_ct);
        pk->inv_prod.ensure(std::max<size_t>(1, upper + (alias ? N : 0)) * ROW);
        pk->inv_inv.ensure(std::max<size_t>(1, upper + (L == 0 ? N : 0)) * ROW);
        pk->inv_fail.ensure(4);
        OrderScope order_(pk->order, s);
        HIP_CHECK(hipMemsetAsync(pk->inv_fail.p, 0, 4, s));
        uint32_t* prod = pk->inv_prod.as<uint32_t>();
        uint32_t* inv = pk->inv_inv.as<uint32_t>();
        const uint32_t* leaves = d_ct;
        if (alias) {                                                    // the way down reads both halves after writing one
            uint32_t* copy = prod + upper * W;
            HIP_CHECK(hipMemcpyAsync(copy, d_ct, N * ROW, hipMemcpyDeviceToDevice, s));
            leaves = copy;
        }
        auto level = [&](int k) -> const uint32_t* { return k == 0 ? leaves : prod + off[k] * W; };
        auto tree_c = [&](int k) -> const uint32_t* { return pk->d_tree_c + (size_t)k * W; };
        g_last_times.clear();
        ScopedKernelTimer t("k_invert", s);
        for (int k = 0; k < L; ++k) {                                   // up
            const size_t h = cnt[k + 1], lo = cnt[k] - h;
            const uint32_t* src = level(k);
            uint32_t* dst = prod + off[k + 1] * W;
            tree_mul(pk, s, src, src + h * W, 0, dst, lo);
            if (lo < h) tree_mul(pk, s, src + lo * W, tree_c(k), 1, dst + lo * W, 1);
        }
        uint32_t* top_out = (L == 0) ? d_out : inv + off[L] * W;
        if (L == 0 && alias) top_out = inv;                             // in-place single level: stage, then copy back
        if (!launch_inv_eea(s, (int)W, pk->d_nsq_words, level(L), top_out, (int)cnt[L], 2 * 32 * (int)W + 64, pk->inv_fail.as<int>()))
            throw PaiError(PAI_E_UNSUPPORTED, "ct_invert: key size without an extended-GCD instantiation");
        HIP_CHECK(hipGetLastError());
        if (L == 0 && alias) HIP_CHECK(hipMemcpyAsync(d_out, inv, N * ROW, hipMemcpyDeviceToDevice, s));
        for (int k = L - 1; k >= 0; --k) {                              // down
            const size_t h = cnt[k + 1], lo = cnt[k] - h;
            const uint32_t* src = level(k);
            const uint32_t* pinv = inv + off[k + 1] * W;
            uint32_t* dst = (k == 0) ? d_out : inv + off[k] * W;
            tree_mul(pk, s, pinv, src + h * W, 0, dst, lo);             // a[i]^-1     = P[i]^-1 a[i + h]
            tree_mul(pk, s, pinv, src, 0, dst + h * W, lo);             // a[i + h]^-1 = P[i]^-1 a[i]
            if (lo < h) tree_mul(pk, s, pinv + lo * W, tree_c(k), 1, dst + lo * W, 1);
        }
        t.stop();
        if (!sync) {
            // asynchronous forms: a non-unit is remembered in the caller's flag word (pai_ct_invert_flag: the outcome travels
            // with the result) or in the handle's sticky status word (pai_ct_invert_async + pai_pubkey_status)
            hipLaunchKernelGGL(k_status_or, dim3(1), dim3(1), 0, s, d_flag ? d_flag : status_word(pk, s), pk->inv_fail.as<int>(), 1);
            HIP_CHECK(hipGetLastError());
            order_.done();
            return;
        }
        int fail = 0;
        HIP_CHECK(hipMemcpyAsync(&fail, pk->inv_fail.p, 4, hipMemcpyDeviceToHost, s));
        order_.done();
        HIP_CHECK(hipStreamSynchronize(s));
        if (fail) throw PaiError(PAI_E_INVALID, "ct_invert: a ciphertext is not invertible modulo n^2");
    });
}

// ---- private key ----------------------------------------------------------------------------------
int pai_privkey_create(const pai_pubkey* pk, const uint32_t* h_p, int p_words, const uint32_t* h_q, int q_words,
                       pai_privkey** out) {
    return guarded([&] {
        require(pk && h_p && h_q && out && p_words > 0 && q_words > 0, "bad arguments");
        std::unique_ptr<pai_privkey, PrivkeyDeleter> sk(new pai_privkey());
        DeviceScope scope_(pk->device);
        sk->pk = pk;
        Limbs p = hbn::from_u32(h_p, (size_t)p_words), q = hbn::from_u32(h_q, (size_t)q_words);
        if (hbn::cmp(p, q) > 0) std::swap(p, q);            // upstream keeps p < q
        require(hbn::cmp(p, q) != 0, "p and q must differ");
        require(hbn::cmp(hbn::mul(p, q), pk->n) == 0, "p*q does not match the public key");
        require(hbn::is_odd(p) && hbn::is_odd(q), "p and q must be odd primes");
        sk->p = p;
        sk->q = q;
        const Limbs one{1u};
        const Limbs g = hbn::add(pk->n, one);
        const Limbs prime[2] = {p, q};
        // both primes share the geometry of the wider one
        Limbs q2 = hbn::mul(q, q);
        sk->wide_nl = wide_nl_for_bits(hbn::bitlen(q2));       // 0: fall back to the lane-group kernel
        if (knob_disabled("wide")) sk->wide_nl = 0;
        for (int w = 0; w < 2; ++w) {
            const Limbs& s = prime[w];
            Limbs s2 = hbn::mul(s, s);
            sk->sq[w].init(s2, sk->wide_nl);
            sk->pr[w].init(s);
        }
        require(sk->sq[0].geo == sk->sq[1].geo && sk->pr[0].geo == sk->pr[1].geo,
                "p and q must have (nearly) the same bit length");
        sk->u_words = std::max(sk->sq[0].w32, sk->sq[1].w32);
        for (int w = 0; w < 2; ++w) {
            const Limbs& s = prime[w];
            const Limbs& s2 = sk->sq[w].M;
            sk->d_r3[w] = upload_r29(sk->sq[w].R3, sk->sq[w].nl);
            Limbs e = hbn::sub(s, one);
            sk->ebits[w] = hbn::bitlen(e);
            sk->ewords[w] = words_for_bits(sk->ebits[w]);
            sk->d_expo[w] = upload_words(e, sk->ewords[w]);
            // h_s = (L_s(g^(s-1) mod s^2))^-1 mod s
            hbn::Mont32 m2(s2);
            Limbs gs = m2.powmod(hbn::mod(g, s2), e);
            Limbs rem;
            Limbs L = hbn::divq(hbn::sub(gs, one), s, &rem);
            require(hbn::is_zero(rem), "L function not exact: p/q are not the factors of n");
            Limbs h = hbn::inv_mod_prime(hbn::mod(L, s), s);
            require(hbn::cmp(hbn::mulmod(h, L, s), one) == 0, "p or q is not prime (inverse check failed)");
            sk->h_host[w] = h;
            const int nl = sk->pr[w].geo->nl;
            sk->d_hR[w] = upload_r29(hbn::mulmod(h, sk->pr[w].R, s), nl);
            const int k = hbn::RB * nl;
            Limbs sinv2 = hbn::inv_mod_pow2(s, k);
            require(hbn::cmp(hbn::low_bits(hbn::mul(sinv2, s), k), one) == 0, "2-adic inverse check failed");
            sk->d_sinv2[w] = upload_r29(sinv2, nl);
            sk->d_nsinv2[w] = upload_r29(hbn::sub(hbn::shl(one, k), sinv2), nl);
        }
        // p-adic digit engine: digit pairs of R^(i+2) mod s^2 and s - 1 as limbs
        sk->padic_nl = padic_nl_for_prime_bits(std::max(hbn::bitlen(p), hbn::bitlen(q)));
        if (knob_disabled("padic")) sk->padic_nl = 0;
        if (sk->padic_nl) {
            const int nl = sk->padic_nl;
            sk->padic_nd = (32 * pk->ct_words + hbn::RB * nl - 1) / (hbn::RB * nl);
            for (int w = 0; w < 2; ++w) {
                const Limbs& s = prime[w];
                const Limbs& s2 = sk->sq[w].M;
                sk->pdig[w].init(s, nl);                                  // modulus context at the digit engine's limb count
                sk->d_pm1[w] = upload_r29(hbn::sub(s, one), nl);
                Limbs Rm = hbn::mod(hbn::shl(one, hbn::RB * nl), s2);
                Limbs K = hbn::mulmod(Rm, Rm, s2);                       // R^2
                std::vector<uint32_t> host((size_t)sk->padic_nd * 2 * nl, 0);
                for (int i = 0; i < sk->padic_nd; ++i) {
                    Limbs rem;
                    Limbs quo = hbn::divq(K, s, &rem);
                    auto ra = hbn::to_r29(rem, nl), rb = hbn::to_r29(quo, nl);
                    std::memcpy(&host[(size_t)(2 * i) * nl], ra.data(), (size_t)nl * 4);
                    std::memcpy(&host[(size_t)(2 * i + 1) * nl], rb.data(), (size_t)nl * 4);
                    K = hbn::mulmod(K, Rm, s2);
                }
                HIP_CHECK(hipMalloc((void**)&sk->d_kdig[w], host.size() * 4));
                HIP_CHECK(hipMemcpy(sk->d_kdig[w], host.data(), host.size() * 4, hipMemcpyHostToDevice));
                {
                    const std::vector<uint16_t> ops = compile_sliding_schedule(hbn::sub(s, one));
                    sk->nops[w] = (int)ops.size();
                    HIP_CHECK(hipMalloc((void**)&sk->d_ops[w], ops.size() * 2));
                    HIP_CHECK(hipMemcpy(sk->d_ops[w], ops.data(), ops.size() * 2, hipMemcpyHostToDevice));
                }
            }
        }
        Limbs pinvq = hbn::inv_mod_prime(hbn::mod(p, q), q);
        require(hbn::cmp(hbn::mulmod(pinvq, p, q), one) == 0, "q is not prime (inverse check failed)");
        sk->pinvq_host = pinvq;
        sk->d_pinvqR = upload_r29(hbn::mulmod(pinvq, sk->pr[1].R, q), sk->pr[1].geo->nl);
        *out = sk.release();
    });
}

void pai_privkey_destroy(pai_privkey* sk) {
    if (!sk) return;
    int prev_ = -1;
    (void)hipGetDevice(&prev_);
    (void)hipSetDevice(sk->pk ? sk->pk->device : 0);
    for (int w = 0; w < 2; ++w) {
        sk->sq[w].release();
        sk->pr[w].release();
        sk->pdig[w].release();
        if (sk->d_r3[w]) (void)hipFree(sk->d_r3[w]);
        if (sk->d_expo[w]) (void)hipFree(sk->d_expo[w]);
        if (sk->d_sinv2[w]) (void)hipFree(sk->d_sinv2[w]);
        if (sk->d_nsinv2[w]) (void)hipFree(sk->d_nsinv2[w]);
        if (sk->d_hR[w]) (void)hipFree(sk->d_hR[w]);
        if (sk->d_pm1[w]) (void)hipFree(sk->d_pm1[w]);
        if (sk->d_kdig[w]) (void)hipFree(sk->d_kdig[w]);
        if (sk->d_ops[w]) (void)hipFree(sk->d_ops[w]);
    }
    if (sk->d_pinvqR) (void)hipFree(sk->d_pinvqR);
    for (int w = 0; w < 2; ++w) {
        sk->mid.sp[w].release();
        sk->mid.s2[w].release();
        if (sk->mid.d_nm1[w]) (void)hipFree(sk->mid.d_nm1[w]);
        if (sk->mid.d_kdig[w]) (void)hipFree(sk->mid.d_kdig[w]);
        if (sk->mid.d_one[w]) (void)hipFree(sk->mid.d_one[w]);
        if (sk->mid.d_sR[w]) (void)hipFree(sk->mid.d_sR[w]);
        sk->mid.table[w].release();
        sk->mid.wv[w].release();
    }
    for (int w = 0; w < 2; ++w) {
        sk->lat.sq[w].release();
        sk->lat.pp[w].release();
        if (sk->lat.d_pp_kdig[w]) (void)hipFree(sk->lat.d_pp_kdig[w]);
        if (sk->lat.d_pp_kx[w]) (void)hipFree(sk->lat.d_pp_kx[w]);
        sk->lat.sq_true[w].release();
        sk->lat.sq2[w].release();
        sk->lat.sq2_true[w].release();
        if (sk->lat.d_r3_2[w]) (void)hipFree(sk->lat.d_r3_2[w]);
        sk->lat.pr[w].release();
        if (sk->lat.d_r3[w]) (void)hipFree(sk->lat.d_r3[w]);
        if (sk->lat.d_ops[w]) (void)hipFree(sk->lat.d_ops[w]);
        if (sk->lat.d_sinv2[w]) (void)hipFree(sk->lat.d_sinv2[w]);
        if (sk->lat.d_nsinv2[w]) (void)hipFree(sk->lat.d_nsinv2[w]);
        if (sk->lat.d_hR[w]) (void)hipFree(sk->lat.d_hR[w]);
    }
    if (sk->lat.d_pinvqR) (void)hipFree(sk->lat.d_pinvqR);
    sk->lat.table.release();
    sk->table.release();
    sk->wscratch.release();
    sk->ubuf.release();
    sk->order.release();
    delete sk;
    if (prev_ >= 0) (void)hipSetDevice(prev_);
}


static void build_latency_consts(pai_privkey* sk) {
    pai_privkey::Lat& L = sk->lat;
    if (L.ready) return;
    L.ready = true;
    const Limbs prime[2] = {sk->p, sk->q};
    // stage A: one integer per wavefront (3 x 64) whenever s^2 k fits — the quotient digits then travel through
    // v_readfirstlane into an SGPR operand (one instruction per digit; 32-lane groups need five), and a product runs
    // over the limbs the modulus needs, not the geometry's capacity, so the idle lanes cost nothing
    const int sq_bits = hbn::bitlen(hbn::mul(sk->q, sk->q));
    const GeoOps* ga = geo_ops_3x64();       // (2 limbs per lane, 2 x 64, measured slower: 4.46 vs 3.77 ms at 2048-bit keys)
    if (sq_bits + hbn::RB * ga->u + 8 > hbn::RB * ga->nl) ga = geo_latency_for_bits(sq_bits + hbn::RB * 3 + 8);
    const GeoOps* gb = geo_latency_for_bits(hbn::bitlen(sk->q));
    if (!ga || !gb) return;                                  // key too wide for the latency geometries: throughput path only
    const Limbs one{1u};
    for (int w = 0; w < 2; ++w) {
        const Limbs& s = prime[w];
        L.sq[w].init_m1(hbn::mul(s, s), ga);
        L.sq_true[w].init(hbn::mul(s, s), 0, ga);
        L.pr[w].init(s, 0, gb);
        L.d_r3[w] = upload_r29(L.sq[w].R3, L.sq[w].nl);
        {
            const std::vector<uint16_t> ops = compile_sliding_schedule(hbn::sub(s, one));
            L.nops[w] = (int)ops.size();
            HIP_CHECK(hipMalloc((void**)&L.d_ops[w], ops.size() * 2));
            HIP_CHECK(hipMemcpy(L.d_ops[w], ops.data(), ops.size() * 2, hipMemcpyHostToDevice));
        }
        const int nl = gb->nl, k = hbn::RB * nl;
        L.d_hR[w] = upload_r29(hbn::mulmod(sk->h_host[w], L.pr[w].R, s), nl);
        Limbs sinv2 = hbn::inv_mod_pow2(s, k);
        L.d_sinv2[w] = upload_r29(sinv2, nl);
        L.d_nsinv2[w] = upload_r29(hbn::sub(hbn::shl(one, k), sinv2), nl);
    }
    L.d_pinvqR = upload_r29(hbn::mulmod(sk->pinvq_host, L.pr[1].R, sk->q), gb->nl);
    L.usable = true;
    if (ga == geo_ops_3x64() && !knob_disabled("lat_pp")) {
        // digit pairs with base s' = s k (minus-one context of s itself, R = 2^(29 r) >= 2^8 s'): the digits of R^(i+2) mod
        // s'^2 take a ciphertext into digit form; R^-1 R_sq^(j+2) mod (s^2 k2) take a + b s' into L.sq's Montgomery form
        bool ok = true;
        const int ct_bits = 32 * sk->pk->ct_words;
        for (int w = 0; w < 2 && ok; ++w) {
            int nd = 0, nch = 0;
            int chain = 1;
            ok = build_pp_consts(prime[w], L.sq[w], ga, ct_bits, L.pp[w], &L.d_pp_kdig[w], &L.d_pp_kx[w], &nd, &nch, &chain);
            if (ok && w == 1 && (nd != L.pp_nd || nch != L.pp_nch || chain != L.pp_chain)) ok = false;
            L.pp_chain = chain;
            L.pp_nd = nd;
            L.pp_nch = nch;
        }
        L.pp_ok = ok;
    }
    const GeoOps* gd = geo_latency_for_bits(sq_bits + hbn::RB * 3 + 8);
    if (gd && gd != ga && gd->t >= 16 && gd->t < ga->t) {
        for (int w = 0; w < 2; ++w) {
            const Limbs s2 = hbn::mul(prime[w], prime[w]);
            L.sq2[w].init_m1(s2, gd);
            L.sq2_true[w].init(s2, 0, gd);
            L.d_r3_2[w] = upload_r29(L.sq2[w].R3, L.sq2[w].nl);
        }
        L.dense = true;
    }
}

// Mid-size decryption (between the four-wave pipeline and the one-element-per-lane engine): constants of the lane-group digit
// pairs with base s = p, q
static bool ensure_mid(pai_privkey* sk) {
    pai_privkey::Mid& M = sk->mid;
    if (M.tried) return M.ok;
    M.tried = true;
    const Limbs prime[2] = {sk->p, sk->q};
    const int nl = pair_nl_for_prime_bits(std::max(hbn::bitlen(sk->p), hbn::bitlen(sk->q)));
    if (!nl || knob_disabled("pair")) return false;
    M.nl = nl;
    M.out_words = (hbn::RB * nl + 31) / 32;
    M.nd = (32 * sk->pk->ct_words + hbn::RB * nl - 1) / (hbn::RB * nl);
    for (int w = 0; w < 2; ++w) {
        const Limbs& sp = prime[w];
        const Limbs s2 = hbn::mul(sp, sp);
        M.sp[w].init(sp, nl);
        M.d_nm1[w] = upload_r29(hbn::sub(sp, Limbs{1u}), nl);
        auto pair_of = [&](const Limbs& v, std::vector<uint32_t>& dst) {
            Limbs rem;
            Limbs quo = hbn::divq(v, sp, &rem);
            auto ra = hbn::to_r29(rem, nl), rb = hbn::to_r29(quo, nl);
            dst.insert(dst.end(), ra.begin(), ra.end());
            dst.insert(dst.end(), rb.begin(), rb.end());
        };
        const Limbs Rm = hbn::mod(hbn::shl(Limbs{1u}, hbn::RB * nl), s2);
        std::vector<uint32_t> kd, one;
        Limbs K = hbn::mulmod(Rm, Rm, s2);
        for (int i = 0; i < M.nd; ++i) {
            pair_of(K, kd);
            K = hbn::mulmod(K, Rm, s2);
        }
        pair_of(Rm, one);
        M.d_kdig[w] = upload_vec(kd);
        M.d_one[w] = upload_vec(one);
        M.s2[w].init(s2);
        M.d_sR[w] = upload_r29(hbn::mulmod(sp, M.s2[w].R, s2), M.s2[w].nl);          // s R mod s^2: v s as one Montgomery product (k_pair_finish)
    }
    M.ok = true;
    return true;
}
// PAI_TUNE dec_mid_min / dec_mid_max: batch range of the lane-group digit-pair stage A (max 0 disables).  Measured at 2048-bit
// keys (profiles/r05/dec_mid.jsonl): 7.3 ms up to 8 192 ciphertexts (one wave of 16 chains per SIMD), 12.0 / 12.3 ms at 12 288 /
// 16 384 — against 9.4 / 12.1 ms of the window kernels at 3 072 / 4 096 and 14.8 ms of the one-element-per-lane engine from 6 144 on
// (level at 2 048: 7.3 / 6.8, behind from ~20 000: 17.7 / 15.0 at 24 576)
static size_t dec_mid_min(size_t ncu, int prime_bits) {
    long long v;
    if (knob_tune("dec_mid_min", &v)) return (size_t)v;
    if ((prime_bits > 900 && prime_bits <= 1024) || (prime_bits > 1400 && prime_bits <= 1536)) return 8 * ncu + 1;      // measured cross-overs at the
    if (prime_bits > 1900 && prime_bits <= 2048) return 9 * ncu + 1;                                                    // 2048 / 3072 / 4096-bit keys
    return 12 * ncu + 1;                  // sizes in between (1536- / 2560- / 3584-bit keys measured: level near 3 000 / 3 600 / 2 700 ciphertexts)
}
static size_t dec_mid_max(size_t ncu, int prime_bits) {
    long long v;
    if (knob_tune("dec_mid_max", &v)) return (size_t)v;
    // measured (profiles/r05/dec_mid.jsonl): 3072-bit keys 19.8 ms flat up to 8 192 against 38.2 at 4 096 and 55.9 beyond, 35 / 51 ms at
    // 16 384 / 24 576; 4096-bit 40 ms up to 8 192 against 67 / 127, 80 / 120 at 16 384 / 24 576; keys in between on the next wider geometry:
    // 1536-bit 5.6 ms against 11.4 at 8 192, 2560-bit 16.7 against 46.8, 3584-bit 35 against 112
    if (prime_bits < 700 || prime_bits > 2048) return 0;
    return prime_bits <= 1024 ? 72 * ncu : 96 * ncu;
}

int pai_decrypt(pai_privkey* sk, const uint32_t* d_ct, size_t N, uint32_t* d_m, void* stream) {
    return guarded([&] {
        require(sk && d_ct && d_m, "NULL argument");
        if (N == 0) return;
        std::lock_guard<std::mutex> lk(sk->mu);
        const pai_pubkey* pk = sk->pk;
        DeviceScope scope_(pk->device);
        DeviceInfo dev = scope_.info;
        hipStream_t s = (hipStream_t)stream;
        const int prime_bits = std::max(hbn::bitlen(sk->p), hbn::bitlen(sk->q));
        if (N >= dec_mid_min((size_t)dev.ncu, prime_bits) && N <= dec_mid_max((size_t)dev.ncu, prime_bits) && sk->u_words && ensure_mid(sk)) {
            pai_privkey::Mid& M = sk->mid;
            const GeoOps* ga = M.s2[0].geo;
            const GeoOps* gb = sk->pr[0].geo;
            const int wbits = var_window_bits(std::max(sk->ebits[0], sk->ebits[1]));
            const int epb = pair_epb(M.nl);
            const size_t tiles = (N + epb - 1) / epb;
            const int pgrid = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)dev.ncu * 6));     // (x 2 primes)
            for (int w = 0; w < 2; ++w) {
                M.table[w].ensure(((size_t)pgrid * epb << wbits) * 2 * (size_t)M.nl * 4);
                M.wv[w].ensure(N * 2 * (size_t)M.out_words * 4);
            }
            sk->ubuf.ensure(2 * N * (size_t)sk->u_words * 4);
            sk->order.begin(s);
            g_last_times.clear();
            PairCtMulParams Q;
            Q.nctx = M.sp[0].d_ctx; Q.nm1 = M.d_nm1[0]; Q.kdig = M.d_kdig[0]; Q.one_pair = M.d_one[0]; Q.table = M.table[0].as<uint32_t>();
            Q.nctx1 = M.sp[1].d_ctx; Q.nm11 = M.d_nm1[1]; Q.kdig1 = M.d_kdig[1]; Q.one_pair1 = M.d_one[1]; Q.table1 = M.table[1].as<uint32_t>();
            Q.nd = M.nd;
            Q.wbits = wbits;
            Q.ct_words = pk->ct_words;
            Q.e_words = sk->ewords[0]; Q.ebits_max = sk->ebits[0];
            Q.e_words1 = sk->ewords[1]; Q.ebits_max1 = sk->ebits[1];
            Q.e1 = sk->d_expo[1];
            Q.wv1 = M.wv[1].as<uint32_t>();
            Q.e_bcast = 1;
            Q.out_words = M.out_words;
            {
                ScopedKernelTimer t("k_dec_a", s);
                if (!launch_pair_ctmul(M.nl, s, pgrid, Q, d_ct, sk->d_expo[0], M.wv[0].as<uint32_t>(), (int)N))
                    throw PaiError(PAI_E_INTERNAL, "no digit-pair kernel for this prime size");
                for (int w = 0; w < 2; ++w) {
                    EncParams P{};
                    P.nsq = M.s2[w].d_ctx;
                    P.nR = M.d_sR[w];
                    P.pt_words = pk->n_words;
                    P.ct_words = sk->u_words;
                    ga->pair_finish(s, grid_for(ga, N, dev.ncu), P, M.wv[w].as<uint32_t>(), M.out_words, nullptr,
                                    sk->ubuf.as<uint32_t>() + (size_t)w * N * sk->u_words, (int)N, 0);
                }
                t.stop();
            }
            HIP_CHECK(hipGetLastError());
            DecBParams B;
            for (int w = 0; w < 2; ++w) {
                B.pr[w] = sk->pr[w].d_ctx;
                B.sinv2[w] = sk->d_sinv2[w];
                B.nsinv2[w] = sk->d_nsinv2[w];
                B.hR[w] = sk->d_hR[w];
            }
            B.pinvqR = sk->d_pinvqR;
            B.u_words = sk->u_words;
            B.pt_words = pk->n_words;
            B.u_is_L = 0;
            {
                ScopedKernelTimer t("k_dec_b", s);
                gb->dec_b(s, grid_for(gb, N, dev.ncu), B, sk->ubuf.as<uint32_t>(), d_m, (int)N);
                t.stop();
            }
            HIP_CHECK(hipGetLastError());
            sk->order.end(s);
            return;
        }
        if (N <= latency_max_elements(LAT_DEC, pk->key_bits)) {
            build_latency_consts(sk);
            if (sk->lat.usable) {
                // small batch: every integer is spread over 16-64 lanes, one product takes microseconds instead of
                // tens of microseconds; (element, prime) pairs fill the device instead of lanes
                pai_privkey::Lat& L = sk->lat;
                // one integer per wavefront while that leaves at most one wave per SIMD (2 N <= 4 x CUs), the denser
                // geometry (two integers per wavefront at 2048-bit keys) beyond
                const bool dense = L.dense && 2 * N > 4 * (size_t)dev.ncu && !lat_dense_disabled();
                const ModSetup* SQ = dense ? L.sq2 : L.sq;
                const ModSetup* SQT = dense ? L.sq2_true : L.sq_true;
                uint32_t* const* R3 = dense ? L.d_r3_2 : L.d_r3;
                const GeoOps* ga = SQ[0].geo;
                const GeoOps* gb = L.pr[0].geo;
                const int u_words = std::max(L.sq_true[0].w32, L.sq_true[1].w32);
                // right-to-left stage A on wave pairs (kernels_paillier.hpp: k_dec_a_rl) while two waves per (element, prime)
                // leave at most one wave per SIMD: 4 N <= 4 x CUs
                const bool rl = N <= lat_rl_max((size_t)dev.ncu);
                const int epb_a = rl ? ga->epb / 2 : ga->epb;
                const int gridx = (int)((N + epb_a - 1) / epb_a);
                if (!rl) L.table.ensure(ga->table_words((size_t)gridx * 2) * 4 / 32 * (PADIC_TBL_ENTRIES + 2));    // odd powers + base^2
                sk->ubuf.ensure(2 * N * (size_t)u_words * 4);
                sk->order.begin(s);
                DecAParams A;
                DecBParams B;
                for (int w = 0; w < 2; ++w) {
                    A.sq[w] = SQ[w].d_ctx;
                    A.fin[w] = SQT[w].d_ctx;
                    A.ops[w] = L.d_ops[w];
                    A.nops[w] = L.nops[w];
                    A.r3[w] = R3[w];
                    A.expo[w] = sk->d_expo[w];
                    A.ewords[w] = sk->ewords[w];
                    A.ebits[w] = sk->ebits[w];
                    B.pr[w] = L.pr[w].d_ctx;
                    B.sinv2[w] = L.d_sinv2[w];
                    B.nsinv2[w] = L.d_nsinv2[w];
                    B.hR[w] = L.d_hR[w];
                }
                A.ct_words = pk->ct_words;
                A.u_words = u_words;
                A.tbl_entries = PADIC_TBL_ENTRIES;
                A.rl = rl ? 1 : 0;
                B.pinvqR = L.d_pinvqR;
                B.u_words = u_words;
                B.pt_words = pk->n_words;
                B.u_is_L = 0;
                g_last_times.clear();
                // the smallest batches (a workgroup per (ciphertext, prime), at most two per CU): digit pairs on four waves
                if (L.pp_ok && 2 * N <= lat_pp_max((size_t)dev.ncu, L.pp_chain, pk->key_bits)) {
                    DecPPParams Q;
                    for (int w = 0; w < 2; ++w) {
                        Q.pp[w] = L.pp[w].d_ctx;
                        Q.kdig[w] = L.d_pp_kdig[w];
                        Q.kx[w] = L.d_pp_kx[w];
                        Q.sq[w] = L.sq[w].d_ctx;
                        Q.fin[w] = L.sq_true[w].d_ctx;
                        Q.expo[w] = sk->d_expo[w];
                        Q.ebits[w] = sk->ebits[w];
                    }
                    Q.nd = L.pp_nd;
                    Q.nch = L.pp_nch;
                    Q.ct_words = pk->ct_words;
                    Q.u_words = u_words;
                    ScopedKernelTimer t("k_dec_a", s);
                    launch_dec_a_pp(s, (int)N, Q, d_ct, sk->ubuf.as<uint32_t>(), L.pp_chain);
                    t.stop();
                } else {
                    ScopedKernelTimer t("k_dec_a", s);
                    ga->dec_a(s, gridx, A, d_ct, sk->ubuf.as<uint32_t>(), (int)N, L.table.as<uint32_t>());
                    t.stop();
                }
                HIP_CHECK(hipGetLastError());
                {
                    ScopedKernelTimer t("k_dec_b", s);
                    gb->dec_b(s, (int)((N + gb->epb - 1) / gb->epb), B, sk->ubuf.as<uint32_t>(), d_m, (int)N);
                    t.stop();
                }
                HIP_CHECK(hipGetLastError());
                sk->order.end(s);
                return;
            }
        }
        const GeoOps* ga = sk->sq[0].geo;
        const GeoOps* gb = sk->pr[0].geo;
        int gridx = grid_for(ga, N, dev.ncu, 1);          // x2 primes => 2 workgroups per CU
        if (sk->padic_nl) {
            const size_t tiles = (N + BLOCK_THREADS - 1) / BLOCK_THREADS;
            const size_t per_prime = (size_t)dev.ncu * (size_t)padic_blocks_per_cu(sk->padic_nl) / 2;   // x2 primes => one (or two) workgroups per CU
            gridx = (int)std::max<size_t>(1, std::min<size_t>(tiles, per_prime));
            sk->table.ensure(padic_table_words(sk->padic_nl, (size_t)gridx * 2) * 4);
            if (const size_t sw = padic_scratch_words(sk->padic_nl, (size_t)gridx * 2)) sk->wscratch.ensure(sw * 4);
        } else if (sk->wide_nl) {
            const size_t tiles = (N + BLOCK_THREADS - 1) / BLOCK_THREADS;
            gridx = (int)std::max<size_t>(1, std::min<size_t>(tiles, (size_t)dev.ncu / 2));   // x2 primes => one workgroup per CU
            sk->table.ensure(wide_table_words(sk->wide_nl, (size_t)gridx * 2) * 4);
        } else {
            sk->table.ensure(ga->table_words((size_t)gridx * 2) * 4);
        }
        sk->ubuf.ensure(2 * N * (size_t)sk->u_words * 4);
        sk->order.begin(s);
        DecAParams A;
        for (int w = 0; w < 2; ++w) {
            A.sq[w] = sk->sq[w].d_ctx;
            A.fin[w] = nullptr;
            A.ops[w] = nullptr;
            A.nops[w] = 0;
            A.r3[w] = sk->d_r3[w];
            A.expo[w] = sk->d_expo[w];
            A.ewords[w] = sk->ewords[w];
            A.ebits[w] = sk->ebits[w];
        }
        A.ct_words = pk->ct_words;
        A.u_words = sk->u_words;
        g_last_times.clear();
        {
            ScopedKernelTimer t("k_dec_a", s);
            if (sk->padic_nl) {
                DecPadicParams Q;
                for (int w = 0; w < 2; ++w) {
                    Q.pr[w] = sk->pdig[w].d_ctx;
                    Q.pm1[w] = sk->d_pm1[w];
                    Q.kdig[w] = sk->d_kdig[w];
                    Q.ops[w] = sk->d_ops[w];
                    Q.nops[w] = sk->nops[w];
                }
                Q.tbl_entries = PADIC_TBL_ENTRIES;
                Q.nd = sk->padic_nd;
                Q.wscratch = sk->wscratch.as<uint4>();
                Q.ct_words = pk->ct_words;
                Q.u_words = sk->u_words;
                if (!launch_dec_a_padic(sk->padic_nl, s, gridx, Q, d_ct, sk->ubuf.as<uint32_t>(), (int)N, sk->table.as<uint32_t>()))
                    throw PaiError(PAI_E_INTERNAL, "no p-adic kernel for this limb count");
            } else if (sk->wide_nl) {
                if (!launch_dec_a_wide(sk->wide_nl, s, gridx, A, d_ct, sk->ubuf.as<uint32_t>(), (int)N, sk->table.as<uint32_t>()))
                    throw PaiError(PAI_E_INTERNAL, "no wide kernel for this limb count");
            } else {
                ga->dec_a(s, gridx, A, d_ct, sk->ubuf.as<uint32_t>(), (int)N, sk->table.as<uint32_t>());
            }
            t.stop();
        }
        HIP_CHECK(hipGetLastError());
        DecBParams B;
        for (int w = 0; w < 2; ++w) {
            B.pr[w] = sk->pr[w].d_ctx;
            B.sinv2[w] = sk->d_sinv2[w];
            B.nsinv2[w] = sk->d_nsinv2[w];
            B.hR[w] = sk->d_hR[w];
        }
        B.pinvqR = sk->d_pinvqR;
        B.u_words = sk->u_words;
        B.pt_words = pk->n_words;
        B.u_is_L = sk->padic_nl ? 1 : 0;
        {
            ScopedKernelTimer t("k_dec_b", s);
            gb->dec_b(s, grid_for(gb, N, dev.ncu), B, sk->ubuf.as<uint32_t>(), d_m, (int)N);
            t.stop();
        }
        HIP_CHECK(hipGetLastError());
        sk->order.end(s);                       // table / u scratch are reused by the next call: ordered by stream or event
    });
}

// ---- multi-GPU helpers (one node) -------------------------------------------------------------------
int pai_shard_plan(size_t N, int nshards, int shard, size_t* begin, size_t* count) {
    return guarded([&] {
        require(nshards > 0 && shard >= 0 && shard < nshards && begin && count, "bad arguments");
        const size_t per = (N + (size_t)nshards - 1) / (size_t)nshards;
        const size_t b = std::min(N, (size_t)shard * per), e = std::min(N, ((size_t)shard + 1) * per);
        *begin = b;
        *count = e - b;
    });
}

static void peer_copy_all(int nshards, const int* devices, void* const* d_shards, const size_t* rows, int row_words, int hub_device,
                          void* d_hub, bool to_hub) {
    require(nshards > 0 && devices && d_shards && rows && d_hub && row_words > 0, "bad arguments");
    const size_t ROW = (size_t)row_words * 4;
    size_t off = 0;
    for (int i = 0; i < nshards; ++i) {           // one copy per shard on the null stream of the shard's device: they overlap
        const size_t bytes = rows[i] * ROW;
        if (bytes) {
            require(d_shards[i] != nullptr, "NULL shard");
            DeviceScope scope_(devices[i]);
            char* hub = static_cast<char*>(d_hub) + off;
            if (devices[i] == hub_device) {
                HIP_CHECK(hipMemcpyAsync(to_hub ? (void*)hub : d_shards[i], to_hub ? d_shards[i] : (void*)hub, bytes, hipMemcpyDeviceToDevice, nullptr));
            } else if (to_hub) {
                HIP_CHECK(hipMemcpyPeerAsync(hub, hub_device, d_shards[i], devices[i], bytes, nullptr));
            } else {
                HIP_CHECK(hipMemcpyPeerAsync(d_shards[i], devices[i], hub, hub_device, bytes, nullptr));
            }
        }
        off += bytes;
    }
    for (int i = 0; i < nshards; ++i) {
        DeviceScope scope_(devices[i]);
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
}

int pai_gather(int nshards, const int* devices, const void* const* d_shards, const size_t* rows, int row_words,
               int dst_device, void* d_out) {
    return guarded([&] { peer_copy_all(nshards, devices, const_cast<void* const*>(d_shards), rows, row_words, dst_device, d_out, true); });
}

int pai_scatter(int nshards, const int* devices, void* const* d_shards, const size_t* rows, int row_words,
                int src_device, const void* d_in) {
    return guarded([&] { peer_copy_all(nshards, devices, d_shards, rows, row_words, src_device, const_cast<void*>(d_in), false); });
}

}  // extern "C"

void PubkeyDeleter::operator()(pai_pubkey* p) const { pai_pubkey_destroy(p); }
void PrivkeyDeleter::operator()(pai_privkey* p) const { pai_privkey_destroy(p); }
void ModulusDeleter::operator()(pai_modulus* p) const { pai_modulus_destroy(p); }
