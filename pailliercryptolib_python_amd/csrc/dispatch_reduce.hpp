// pai_ct_prod / pai_ct_multiexp / pai_ct_invert*: the reductions of ipcl_python.py:746-930 (sum, dot, matmul) and the batch
// inversion behind negative multipliers (ipcl_python.py:426-437).
// (Part of the C-API translation unit: included by paillier_capi.hip inside extern "C"; not a stand-alone header.)
#pragma once
// one level of a product tree on single Montgomery products: out[i] = a[i] * b[i] * R^-1 mod n^2
static void tree_mul(const pai_pubkey* pk, hipStream_t s, const uint32_t* a, const uint32_t* b, int b_bcast, uint32_t* out, size_t n) {
    if (n == 0) return;
    const GeoOps* g = pk->msq.geo;
    g->modmul(s, grid_for(g, n, pk->dev.ncu), pk->msq.d_ctx, a, b, out, (int)n, pk->ct_words, b_bcast, MODMUL_MONT, nullptr);
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
        OrderScope order_10(pk->order, s);
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
        order_10.done();
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
