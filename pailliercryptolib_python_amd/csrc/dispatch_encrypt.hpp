// pai_raw_encrypt / pai_encrypt / pai_obfuscate: which kernel family serves a batch (ranges: path_ranges.hpp, section encrypt).
// Replaces ipcl::PublicKey::encrypt / apply_obfuscator behind bindings/ipcl_bindings_classes.cpp:53-60,71-83.
// (Part of the C-API translation unit: included by paillier_capi.hip inside extern "C"; not a stand-alone header.)
#pragma once
static bool ensure_midp(const pai_pubkey* pk);
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
            OrderScope order_2(pk->order, s);
            ScopedKernelTimer t(from_plain ? "k_encrypt(djn)" : "k_encrypt(obfuscate)", s);
            if (!launch_pair_fixed_base(pk->midp_nl, s, pgrid, Q, d_m, d_r, pk->pair_wv.as<uint32_t>(), (int)N, from_plain ? 1 : 0))
                throw PaiError(PAI_E_INTERNAL, "no digit-pair kernel for this limb count");
            g->pair_finish(s, grid, P, pk->pair_wv.as<uint32_t>(), pk->midp_out_words, d_ct_in, d_ct_out, (int)N, from_plain ? 0 : 1);
            t.stop();
            HIP_CHECK(hipGetLastError());
            order_2.done();
            return;
        }
    }
    if (d_r && pk->djn && N <= latency_max_elements(LAT_ENC, pk->key_bits, (size_t)pk->dev.ncu)) {
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
            OrderScope order_3(pk->order, s);
            ScopedKernelTimer t(from_plain ? "k_encrypt(djn)" : "k_encrypt(obfuscate)", s);
            const int per_wg = tree ? 64 / gl->t : gl->epb;
            gl->encrypt(s, (int)((N + per_wg - 1) / per_wg), PL, d_m, d_r, d_ct_in, d_ct_out, (int)N, (from_plain ? 1 : 2) + (tree ? 4 : 0));
            t.stop();
            HIP_CHECK(hipGetLastError());
            order_3.done();
            return;
        }
    }
    if (d_r == nullptr && from_plain && N <= 2 * lat_add_max((size_t)pk->dev.ncu) && ensure_lat_ctx(pk)) {
        // small raw encryptions (the plaintext side of ct + pt): 1 + m n as ONE product with n^2 spread over a wavefront
        // (k_encrypt mode 0 on the latency geometry): 25 against 60 us of kernel time
        const GeoOps* gl = pk->lat_msq.geo;
        // ... on the minus-one context of n^2 where the key has one (PAI_DISABLE=lat_add_m1: the conventional context)
        const bool m1 = pk->lat_m1_ok && gl->t >= 16 && gl->t <= 64 && !knob_disabled("lat_add_m1");
        if (m1 && !pk->d_lat_nR_m1) pk->d_lat_nR_m1 = upload_r29(hbn::mulmod(pk->n, pk->lat_msq_m1.R, pk->lat_msq_m1.M), pk->lat_msq_m1.nl);
        if (!m1 && !pk->d_lat_nR) pk->d_lat_nR = upload_r29(hbn::mulmod(pk->n, pk->lat_msq.R, pk->nsq), pk->lat_msq.nl);
        EncParams PL;
        PL.nsq = m1 ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx;
        PL.nR = m1 ? pk->d_lat_nR_m1 : pk->d_lat_nR;
        PL.fin = m1 ? pk->lat_msq.d_ctx : nullptr;
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
    OrderScope order_4(pk->order, s);
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
    order_4.done();
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

// ct + plaintext in one pass: k_encrypt mode 3 with the ciphertext in the obfuscator's place ((1 + m n) * ct)
int pai_ct_add_plain(const pai_pubkey* pk, const uint32_t* d_ct, const uint32_t* d_m, size_t N, uint32_t* d_out, void* stream) {
    return guarded([&] {
        require(pk && d_ct && d_m && d_out, "NULL argument");
        if (N == 0) return;
        DeviceScope scope_(pk->device);
        hipStream_t s = (hipStream_t)stream;
        g_last_times.clear();
        std::lock_guard<std::mutex> lk(pk->mu);
        ScopedKernelTimer t("k_encrypt(add_plain)", s);
        if (N <= 2 * lat_add_max((size_t)pk->dev.ncu) && ensure_lat_ctx(pk)) {
            const GeoOps* gl = pk->lat_msq.geo;
            // ... on the minus-one context of n^2 where the key has one (as the small aligned additions; PAI_DISABLE=lat_add_m1)
            const bool m1 = pk->lat_m1_ok && gl->t >= 16 && gl->t <= 64 && !knob_disabled("lat_add_m1");
            if (m1 && !pk->d_lat_nR_m1) pk->d_lat_nR_m1 = upload_r29(hbn::mulmod(pk->n, pk->lat_msq_m1.R, pk->lat_msq_m1.M), pk->lat_msq_m1.nl);
            if (!m1 && !pk->d_lat_nR) pk->d_lat_nR = upload_r29(hbn::mulmod(pk->n, pk->lat_msq.R, pk->nsq), pk->lat_msq.nl);
            EncParams PL;
            PL.nsq = m1 ? pk->lat_msq_m1.d_ctx : pk->lat_msq.d_ctx;
            PL.nR = m1 ? pk->d_lat_nR_m1 : pk->d_lat_nR;
            PL.fin = m1 ? pk->lat_msq.d_ctx : nullptr;
            PL.fb_table = nullptr;
            PL.fb_windows = 0;
            PL.fb_wbits = 0;
            PL.pt_words = pk->n_words;
            PL.ct_words = pk->ct_words;
            PL.r_words = pk->r_words;
            gl->encrypt(s, (int)((N + gl->epb - 1) / gl->epb), PL, d_m, d_ct, nullptr, d_out, (int)N, 3);
        } else {
            const GeoOps* g = pk->msq.geo;
            g->encrypt(s, grid_for(g, N, pk->dev.ncu), pk->enc_params(), d_m, d_ct, nullptr, d_out, (int)N, 3);
        }
        t.stop();
        HIP_CHECK(hipGetLastError());
    });
}
