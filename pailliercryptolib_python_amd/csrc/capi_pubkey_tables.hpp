// Public-key internals of the C API: small-batch contexts (minus-one Montgomery contexts, four-wave pipeline constants), DJN
// fixed-base tables (two-level build, g-factoring) and their per-device LRU cache.
// (Part of the C-API translation unit: included by paillier_capi.hip; not a stand-alone header.)
#pragma once
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
// Measured at 2048-bit keys (profiles/r04/lat_add_probe.jsonl): wire-form a b 30 against 60 us up to 1024 elements (39 / 65 at
// 2048, level at 4096), the tagged single product 29 against 35 us up to 1024 (level at 2048), aligned additions with shifts
// up to 13: 0.18 against 0.44 ms up to 1024, 0.31 / 0.45 at 4096 — hence the scale factors 1 / 4 on PAI_LAT_ADD_MAX; the wire form's
// factor is per key size since round 6 (path_ranges.hpp: lat_add_wire_scale, measured with the minus-one contexts on this side and
// the most-significant-limb-first product on the other).
static const ModSetup* lat_add_ctx(const pai_pubkey* pk, size_t N, bool tagged, int scale = 1) {
    if (N > (size_t)scale * lat_add_max((size_t)pk->dev.ncu)) return nullptr;
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
    pk->fb_registered = 0;
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
    for (pai_pubkey* o : g_fb.lru) if (o->device == pk->device) used += o->fb_registered;
    for (size_t i = 0; i < g_fb.lru.size() && used + need > budget;) {
        pai_pubkey* v = g_fb.lru[i];
        if (v == pk || v->device != pk->device || !v->mu.try_lock()) { ++i; continue; }
        (void)hipDeviceSynchronize();                  // nothing in flight may still read the victim's tables
        used -= std::min(used, v->fb_registered);
        freed += v->fb_registered;
        fb_free_tables(v);
        v->fb_registered = 0;
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
            // (fb_registered, not fb_bytes: another key's build writes its fb_bytes under its own mutex only)
            for (pai_pubkey* o : g_fb.lru) if (o->device == pk->device && o != pk) { used += o->fb_registered; ++resident; }
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
            pk->fb_registered = pk->fb_bytes;
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
