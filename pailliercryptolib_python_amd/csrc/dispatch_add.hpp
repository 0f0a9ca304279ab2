// pai_ct_add / pai_ct_add_aligned[_dom] / pai_ct_mont_mul / pai_ct_addn: ciphertext + ciphertext (ranges: path_ranges.hpp,
// section ct + ct).  Replaces CipherText::operator+ behind bindings/ipcl_bindings_classes.cpp:318-321 and __raw_add
// (ipcl_python.py:490-526).
// (Part of the C-API translation unit: included by paillier_capi.hip inside extern "C"; not a stand-alone header.)
#pragma once
int pai_ct_add(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, size_t N,
               uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_a && d_b && d_out, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        g_last_times.clear();
        const ModSetup* L = lat_add_ctx(pk, N, false, lat_add_wire_scale(pk->key_bits));
        // beyond the small-batch range, two wire-form operands: ONE most-significant-limb-first product (mont_msb.hpp) where the key
        // has the context (PAI_DISABLE=add_msb: the two Montgomery products below).  (The division kernel of round 6 — base-n digits and
        // Barrett division on the one-element-per-lane engine, 3.1 ms per 2^20 at 2048 bits against 2.8 here — is in the history: DESIGN.md section 8.)
        const bool msb = L == nullptr && pk->d_msb != nullptr && !b_bcast && !knob_disabled("add_msb");
        if (msb) {
            const GeoOps* g = pk->msq.geo;
            ScopedKernelTimer t("k_modmul_msb", (hipStream_t)stream);
            g->modmul_msb((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->d_msb, d_a, d_b, d_out, (int)N, pk->ct_words);
            t.stop();
            HIP_CHECK(hipGetLastError());
            return;
        }
        const GeoOps* g = L ? L->geo : pk->msq.geo;
        // small batches without a broadcast addend: the two products on the minus-one context of n^2 (PAI_DISABLE=lat_add_m1)
        const bool m1 = L != nullptr && !b_bcast && pk->lat_m1_ok && g->t >= 16 && !knob_disabled("lat_add_m1");
        ScopedKernelTimer t("k_modmul", (hipStream_t)stream);
        g->modmul((hipStream_t)stream, L ? (int)((N + g->epb - 1) / g->epb) : grid_for(g, N, pk->dev.ncu),
                  m1 ? pk->lat_msq_m1.d_ctx : (L ? L->d_ctx : pk->msq.d_ctx), d_a, d_b, d_out, (int)N, pk->ct_words, b_bcast,
                  MODMUL_FULL, m1 ? pk->lat_msq.d_ctx : nullptr);
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
    // ... on the minus-one context of n^2 where the key has one (PAI_DISABLE=lat_add_m1: the conventional context)
    const bool m1 = L != nullptr && pk->lat_m1_ok && g->t >= 16 && !knob_disabled("lat_add_m1");
    g_last_times.clear();
    ScopedKernelTimer t("k_add_aligned", (hipStream_t)stream);
    g->add_aligned((hipStream_t)stream, L ? (int)((N + g->epb - 1) / g->epb) : grid_for(g, N, pk->dev.ncu),
                   m1 ? pk->lat_msq_m1.d_ctx : (L ? L->d_ctx : pk->msq.d_ctx), d_a, d_b, b_bcast, d_delta, d_out, (int)N,
                   pk->ct_words, d_entry, m1 ? pk->lat_msq.d_ctx : nullptr);
    t.stop();
    HIP_CHECK(hipGetLastError());
}

int pai_ct_add_aligned(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* d_delta,
                       size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] { add_aligned_common(pk, d_a, d_b, b_bcast, d_delta, N, d_out, nullptr, stream); });
}

// pai_ct_add_aligned with the shifts still on the host (<= PAI_HOST_STAGE_MAX bytes): staged and read by the kernel in place — one call
// where pai_host_stage + pai_ct_add_aligned are two (the reference's BM_Add_CTCT: ~4 us of its 57)
int pai_ct_add_aligned_host(const pai_pubkey* pk, const uint32_t* d_a, const uint32_t* d_b, int b_bcast, const int32_t* h_delta,
                            size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && h_delta, "NULL argument");
        const void* src[1] = {h_delta};
        const size_t len[1] = {N * sizeof(int32_t)};
        void* dp[1] = {nullptr};
        host_stage_parts(pk->device, 1, src, len, stream, dp);
        add_aligned_common(pk, d_a, d_b, b_bcast, static_cast<const int32_t*>(dp[0]), N, d_out, nullptr, stream);
    });
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
                       MODMUL_FULL, nullptr);
            t.stop();
            HIP_CHECK(hipGetLastError());
            return;
        }
        const GeoOps* g = pk->msq.geo;
        ScopedKernelTimer t("k_modmul", (hipStream_t)stream);
        g->modmul((hipStream_t)stream, grid_for(g, N, pk->dev.ncu), pk->msq.d_ctx, d_a, d_b, d_out, (int)N, pk->ct_words, b_bcast,
                  MODMUL_MONT, nullptr);
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
