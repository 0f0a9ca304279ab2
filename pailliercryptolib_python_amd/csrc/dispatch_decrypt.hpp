// pai_privkey_create / pai_privkey_destroy / pai_decrypt: CRT decryption, which kernel family serves a batch (ranges:
// path_ranges.hpp, section decrypt).  Replaces ipcl::PrivateKey::decrypt behind bindings/ipcl_bindings_classes.cpp:127-133.
// (Part of the C-API translation unit: included by paillier_capi.hip inside extern "C"; not a stand-alone header.)
#pragma once
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
            OrderScope order_11(sk->order, s);
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
            order_11.done();
            return;
        }
        if (N <= latency_max_elements(LAT_DEC, pk->key_bits, (size_t)dev.ncu)) {
            build_latency_consts(sk);
            if (sk->lat.usable) {
                // small batch: every integer is spread over 16-64 lanes, one product takes microseconds instead of
                // tens of microseconds; (element, prime) pairs fill the device instead of lanes
                pai_privkey::Lat& L = sk->lat;
                // one integer per wavefront while that leaves at most one wave per SIMD (2 N <= 4 x CUs), the denser
                // geometry (two integers per wavefront at 2048-bit keys) beyond
                const bool dense = L.dense && N >= lat_dense_min((size_t)dev.ncu) && !lat_dense_disabled();
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
                OrderScope order_12(sk->order, s);
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
                order_12.done();
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
        OrderScope order_13(sk->order, s);
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
        order_13.done();                       // table / u scratch are reused by the next call: ordered by stream or event
    });
}
