// pai_ct_mul / pai_ct_pow2[_hint]: ciphertext x plaintext and the exponent alignment ct^(2^delta) (ranges: path_ranges.hpp,
// section ct x pt).  Replaces CipherText::operator*(PlainText) behind bindings/ipcl_bindings_classes.cpp:324-325 and the
// alignment loops of ipcl_python.py:570-741.
// (Part of the C-API translation unit: included by paillier_capi.hip inside extern "C"; not a stand-alone header.)
#pragma once
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
    OrderScope order_5(pk->order, s);
    ScopedKernelTimer t(timer_name, s);
    if (!launch_ctmul_padic(pk->penc_nl, s, grid, Q, d_ct, d_e, d_out, (int)N))
        throw PaiError(PAI_E_INTERNAL, "no digit-engine ct*pt kernel for this limb count");
    t.stop();
    HIP_CHECK(hipGetLastError());
    order_5.done();
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
        if (N <= latency_max_elements(LAT_MUL, pk->key_bits, (size_t)pk->dev.ncu) && ebits_max > 8) {
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
                OrderScope order_6(pk->order, s);
                ScopedKernelTimer t("k_ctmul", s);
                launch_ctmul_pp(s, (int)N, Q, d_ct, d_out, pk->lat_pp_chain);
                t.stop();
                HIP_CHECK(hipGetLastError());
                order_6.done();
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
                OrderScope order_7(pk->order, s);
                ScopedKernelTimer t("k_ctmul", s);
                g->modexp_var_win(s, grid, pk->lat_m1_ok ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx, d_ct, pk->ct_words, d_e, e_words,
                                  ebits_max, e_bcast, d_out, pk->ct_words, (int)N, pk->lat_table.as<uint32_t>(), wbits,
                                  pk->lat_m1_ok ? pk->lat_msq.d_ctx : nullptr);
                t.stop();
                HIP_CHECK(hipGetLastError());
                order_7.done();
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
            OrderScope order_8(pk->order, s);
            ScopedKernelTimer t("k_ctmul", s);
            g->modexp_var_win(s, grid, pk->msq.d_ctx, d_ct, pk->ct_words, d_e, e_words, ebits_max, e_bcast, d_out,
                              pk->ct_words, (int)N, pk->ctmul_table.as<uint32_t>(), wbits, nullptr);
            t.stop();
            HIP_CHECK(hipGetLastError());
            order_8.done();
            return;
        }
        g->modexp_var(s, grid, pk->msq.d_ctx, d_ct, pk->ct_words, 0, d_e, e_words,
                      ebits_max, e_bcast, d_out, pk->ct_words, (int)N, 0, 0);
        HIP_CHECK(hipGetLastError());
    });
}

// pai_ct_mul with the exponents still on the host (<= PAI_HOST_STAGE_MAX bytes): staged and read by the kernel in place
int pai_ct_mul_host(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* h_e, int e_words, int ebits_max, int e_bcast, size_t N,
                    uint32_t* d_out, void* stream) {
    const void* dp0 = nullptr;
    const int rc = guarded([&] {
        require(pk && h_e && e_words > 0, "NULL argument");
        const void* src[1] = {h_e};
        const size_t len[1] = {(e_bcast ? 1 : N) * (size_t)e_words * 4};
        void* dp[1] = {nullptr};
        host_stage_parts(pk->device, 1, src, len, stream, dp);
        dp0 = dp[0];
    });
    if (rc != PAI_OK) return rc;
    return pai_ct_mul(pk, d_ct, static_cast<const uint32_t*>(dp0), e_words, ebits_max, e_bcast, N, d_out, stream);
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
        if (pk->penc_nl && N >= pow2_digit_min_elements((size_t)pk->dev.ncu)) {
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
            OrderScope order_9(pk->order, s);
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
            order_9.done();
            if (dmax >= POW2_DIGIT_MIN_SHIFT && dmax <= 62) {
                ctmul_padic_locked(pk, s, d_ct, pk->pow2_expo.as<uint32_t>(), 2, dmax + 1, 0, N, d_ct, 1, "k_pow2");
                return;
            }
            if (dmax == 0) return;
        }
        ScopedKernelTimer t("k_pow2", (hipStream_t)stream);
        if (const ModSetup* L = lat_add_ctx(pk, N, false, 4)) {           // small batches: an integer per wavefront (as the aligned additions)
            const GeoOps* gl = L->geo;
            const bool m1 = pk->lat_m1_ok && gl->t >= 16 && !knob_disabled("lat_add_m1");       // ... on the minus-one context of n^2
            gl->pow2((hipStream_t)stream, (int)((N + gl->epb - 1) / gl->epb), m1 ? pk->lat_msq_m1.d_ctx : L->d_ctx, d_ct, d_delta, delta_bcast,
                     (int)N, pk->ct_words, m1 ? pk->lat_msq.d_ctx : nullptr);
            t.stop();
            HIP_CHECK(hipGetLastError());
            return;
        }
        g->pow2((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->msq.d_ctx, d_ct, d_delta, delta_bcast, (int)N, pk->ct_words, nullptr);
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}
