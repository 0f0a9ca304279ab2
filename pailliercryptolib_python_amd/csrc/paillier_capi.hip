// C ABI of libpaillier_hip.so (see include/paillier_hip.h): key set-up on the host, dispatch of the
// gfx950 kernels.  There is deliberately no CPU implementation of any hot operation in this file.
#include "../../include/paillier_hip.h"

#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
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

// Constants of the most-significant-limb-first product (mont_msb.hpp) for modulus M on geometry g and rows of w32 words, or
// nullptr where its conditions do not hold: a lane-group geometry with the estimate's four cells in one lane, M's top limb at
// least one limb below the geometry's (off >= 1: the products a b_i then stay below 4 Mt for ANY word pattern of the row), 3 .. 26
// bits of M in its top limb, rows no wider than the limbs the multiplier is read from.
static MsbCtx* build_msb_ctx(const Limbs& M, const GeoOps* g, int w32) {
    if (!g || !g->modmul_msb || g->t > 8 || g->nll < 4) return nullptr;
    const int NL = g->nl, bits = hbn::bitlen(M);
    const int mtop = (bits - 1) / hbn::RB, off = NL - 1 - mtop;
    if (off < 1 || off > MSB_OFF_MAX) return nullptr;
    const int tb = bits - hbn::RB * mtop;
    if (tb < 3 || tb > 26) return nullptr;
    if (32 * w32 > hbn::RB * (mtop + 1) || 32 * w32 > bits + 2) return nullptr;
    const Limbs one{1u};
    const Limbs Mt = hbn::shl(M, hbn::RB * off);
    const int P = hbn::bitlen(Mt);
    const Limbs W = hbn::sub(hbn::shl(one, hbn::RB * NL), Mt);
    const Limbs mu = hbn::divq(hbn::shl(one, P + 31), Mt, nullptr);
    require(hbn::bitlen(mu) <= 32 && P == hbn::RB * (NL - 1) + tb, "msb context: construction failed");
    MsbCtx h;
    std::memset(&h, 0, sizeof(h));
    const auto w = hbn::to_r29(W, NL), mt = hbn::to_r29(Mt, NL);
    std::memcpy(h.w, w.data(), (size_t)NL * 4);
    std::memcpy(h.mt, mt.data(), (size_t)NL * 4);
    h.mu = mu.empty() ? 0u : mu[0];
    h.tb = (uint32_t)tb;
    h.off = (uint32_t)off;
    h.nl = (uint32_t)NL;
    h.m2 = 1u << (32 - tb);
    h.eight = 8u;
    MsbCtx* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, sizeof(MsbCtx)));
    HIP_CHECK(hipMemcpy(d, &h, sizeof(MsbCtx), hipMemcpyHostToDevice));
    return d;
}

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
//                                  serve), pair, pair_ctmul, wide, gform (plain fixed-base tables), fb_chain, lat_dense, lat_enc_m1, lat_add_m1
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

#include "path_ranges.hpp"

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
    MsbCtx* d_msb = nullptr;      // wire-form ct + ct by one most-significant-limb-first product (mont_msb.hpp), where the key allows it
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
    mutable size_t fb_registered = 0;  // fb_bytes as entered in the per-device LRU: written and read under g_fb.mu only (other handles sum it)
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

}  // extern "C"
namespace {
// the body of pai_host_stage; also behind the calls that take a host operand themselves (pai_ct_add_aligned_host, pai_ct_mul_host)
void host_stage_parts(int device, int parts, const void* const* h_src, const size_t* bytes, void* stream, void** d_ptrs) {
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
}
}  // namespace
extern "C" {

int pai_host_stage(int device, int parts, const void* const* h_src, const size_t* bytes, void* stream, void** d_ptrs) {
    return guarded([&] { host_stage_parts(device, parts, h_src, bytes, stream, d_ptrs); });
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
        g->modmul((hipStream_t)stream, grid_for(g, N, m->dev.ncu), m->ms.d_ctx, d_a, d_b, d_out, (int)N, m->ms.w32, b_bcast, MODMUL_FULL, nullptr);
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
        OrderScope order_0(m->order, s);
        m->pinned.h2d(m->expo.p, h_e, (size_t)e_words * 4, s);         // h_e is free when this call returns
        const int grid = grid_for(g, N, m->dev.ncu);
        m->table.ensure(g->table_words((size_t)grid) * 4);
        g->modexp_fixed(s, grid, m->ms.d_ctx, d_base, m->ms.w32, m->expo.as<uint32_t>(), e_words, ebits > 0 ? ebits : 1,
                        d_out, m->ms.w32, (int)N, m->table.as<uint32_t>(), 0);
        HIP_CHECK(hipGetLastError());
        order_0.done();
    });
}
// window width of the per-element-exponent kernels: table build 2^w - 2 products, then w squarings + 1 product per window
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
            OrderScope order_1(m->order, (hipStream_t)stream);
            g->modexp_var_win((hipStream_t)stream, grid, m->ms.d_ctx, d_base, m->ms.w32, d_e, e_words, ebits_max, e_bcast, d_out,
                              m->ms.w32, (int)N, m->table.as<uint32_t>(), wbits, nullptr);
            HIP_CHECK(hipGetLastError());
            order_1.done();
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

#include "capi_pubkey_tables.hpp"


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
        if (!knob_disabled("add_msb")) pk->d_msb = build_msb_ctx(pk->nsq, pk->msq.geo, pk->ct_words);
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
    if (pk->d_msb) (void)hipFree(pk->d_msb);
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

int pai_path_edges(const pai_pubkey* pk, int op, size_t* edges, int cap, int* count) {
    return guarded([&] {
        require(pk && count && op >= 0 && op <= 3 && (edges || cap == 0), "bad arguments");
        const std::vector<size_t> e = path_edges(op, pk->key_bits, (size_t)pk->dev.ncu);
        *count = (int)e.size();
        for (int i = 0; i < (int)e.size() && i < cap; ++i) edges[i] = e[(size_t)i];
    });
}

#include "dispatch_encrypt.hpp"
#include "dispatch_ctmul.hpp"
#include "dispatch_add.hpp"
#include "dispatch_reduce.hpp"
#include "dispatch_decrypt.hpp"

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
