// Which kernel family serves a batch: every batch-size switch of the C API in ONE place and ONE convention.
// (Part of the C-API translation unit: included by paillier_capi.hip behind the run-time knobs, inside its anonymous namespace.)
//
// Convention: a range is a count of ELEMENTS PER COMPUTE UNIT times the device's CU count (ncu) — a batch fills a device by
// wavefronts per SIMD, so the cross-overs measured on the 256 CUs of an MI355X carry over to a partition with fewer.  Tables hold
// HALF-elements per CU (x2) where a measured cross-over is not a whole multiple.  Overrides (tests, probes; read at every call):
//   PAI_LATENCY_MAX=v   the small-batch switch of decrypt / DJN encrypt (2 v elements) and ct x pt (4 v) in ABSOLUTE elements;
//                       0 = throughput kernels for every batch size: it also switches the mid-size paths off (round 6)
//   PAI_LAT_ADD_MAX=v   largest ct + ct batch on the latency geometry, absolute (raw encrypt 2 v, aligned add / pow2 4 v)
//   PAI_POW2_DIGIT_MIN=v  pai_ct_pow2 on the digit engine from v elements, absolute
//   PAI_TUNE="name=v,..."  per-path limits in absolute elements, 0 = path off (names below)
// Each operation, smallest batches first (2048-bit keys, 256 CUs; measurements: profiles/r04/lat_*.jsonl, profiles/r05/
// path_switch_sweep*.jsonl, lat_pp_range.jsonl, enc_mid.jsonl, ctmul_mid.jsonl, dec_mid.jsonl):
//   decrypt   four-wave digit pairs (lat_pp: <= 6 ncu chains) | wave pairs (lat_rl: <= ncu) | window kernels on one / two integers
//             per wavefront (<= 8..9 ncu) | lane-group digit pairs, 4 lanes (dec_mid: .. 72 / 96 ncu) | one element per lane
//   encrypt   wave-shared fixed-base chain (lat_enc_tree: the whole latency range, <= 16 ncu) | lane-group digit pairs on the
//             digit engine's own table (enc_mid: 16 .. 160 ncu) | one element per lane
//   ct x pt   four-wave digit pairs (lat_mul_pp: <= 4 ncu) | wave pairs (lat_mul_rl: <= 2 ncu) | window kernel (<= 20 ncu) |
//             lane-group digit pairs (ctmul_mid: 20 .. 192 ncu) | one element per lane
//   ct + ct   one integer per wavefront (tagged <= 4 ncu; wire form <= 8 .. 24 ncu by key size: lat_add_wire_scale; aligned additions and
//             pow2 <= 16 ncu, raw encryption <= 8 ncu) | wave tiles: wire form by ONE most-significant-limb-first product (mont_msb.hpp) where
//             the key has the context (PAI_DISABLE=add_msb: off), else two Montgomery products
#pragma once

enum LatOp { LAT_DEC, LAT_ENC, LAT_MUL };

static bool latency_paths_forced_off() {
    const char* env = std::getenv("PAI_LATENCY_MAX");
    return env && std::strtoull(env, nullptr, 10) == 0;
}

// ---- the small-batch (latency) geometries hand over to the throughput kernels --------------------------------------------
// Batches up to this many elements take the latency paths: every integer spread over 16-64 lanes, (element, prime) pairs
// filling the device instead of lanes.  Measured cross-overs (profiles/r05/path_switch_sweep.jsonl: tools/latency_sweep.py,
// default against both forced paths), in half-elements per CU:  {decrypt, DJN encrypt, ct x pt}
static size_t latency_max_elements(LatOp op, int key_bits, size_t ncu) {
    if (const char* env = std::getenv("PAI_LATENCY_MAX")) return (size_t)std::strtoull(env, nullptr, 10) * (op == LAT_MUL ? 4 : 2);
    static const struct { int bits; unsigned x2[3]; } T[] = {
        {1024, {50, 68, 104}},        // 6 400 / 8 704 / 13 312 at 256 CUs: decrypt 3.4 ms at 6 144 against 3.6; encrypt 0.45 at 8 192 against 0.48; ct x pt 1.13 at 12 288 against 1.23
        {2048, {40, 92, 148}},        // 5 120 / 11 776 / 18 944: 12.1 ms at 4 096 against 14.8; 3.4 at 12 288 against 3.3; 4.0 at 16 384 against 4.6
        {3072, {48, 38, 44}},         // 6 144 / 4 864 / 5 632: 55.9 at 6 144 against 55.9; 3.3 at 4 096 against 4.0; 2.1 at 4 096 against 3.0
        {4096, {62, 25, 23}},         // 7 936 / 3 200 / 2 944: 98 at 6 144 against 127; 4.6 at 3 072 against 4.9; 1.8 at 2 048 against 2.7
    };
    for (const auto& t : T) if (key_bits <= t.bits) return (size_t)t.x2[op] * ncu / 2;
    return (size_t)T[3].x2[op] * ncu / 2;
}

// ---- decrypt ----------------------------------------------------------------------------------------------------------------
static bool lat_dense_disabled() {                  // PAI_DISABLE=lat_dense: small-batch stage A always on one integer per wavefront
    return knob_disabled("lat_dense");
}
static size_t lat_dense_min(size_t ncu) {           // two integers per wavefront (window kernels) from this many ciphertexts on
    return 2 * ncu + 1;
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
    if (latency_paths_forced_off()) return 0;
    // measured (profiles/r05/dec_mid.jsonl): 3072-bit keys 19.8 ms flat up to 8 192 against 38.2 at 4 096 and 55.9 beyond, 35 / 51 ms at
    // 16 384 / 24 576; 4096-bit 40 ms up to 8 192 against 67 / 127, 80 / 120 at 16 384 / 24 576; keys in between on the next wider geometry:
    // 1536-bit 5.6 ms against 11.4 at 8 192, 2560-bit 16.7 against 46.8, 3584-bit 35 against 112
    if (prime_bits < 700 || prime_bits > 2048) return 0;
    return prime_bits <= 1024 ? 72 * ncu : 96 * ncu;
}

// ---- encrypt ----------------------------------------------------------------------------------------------------------------
static size_t lat_enc_tree_max(size_t ncu) {        // PAI_TUNE lat_enc_tree: largest batch of the wave-shared small-batch encryption (0 disables)
    long long v;
    (void)ncu;
    return knob_tune("lat_enc_tree", &v) ? (size_t)v : (size_t)1 << 30;                         // measured ahead over the whole latency range (2048-bit keys: 0.29 vs 0.98 ms up to 256
}                                                   // elements, 0.54 vs 1.01 at 1024, 1.63 vs 1.91 at 4096; profiles/r04/lat_enc_tree.jsonl)
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
    if (latency_paths_forced_off()) return 0;
    // (the caller also needs the pair geometry's limb count to equal the digit engine's: 1024-class keys and 1537 .. 2048-bit keys)
    return n_bits > 900 && n_bits <= 2048 ? 160 * ncu : 0;
}

// ---- ct x pt ------------------------------------------------------------------------------------------------------------------
static size_t lat_mul_pp_max(size_t ncu) {          // PAI_TUNE lat_mul_pp: largest batch of the four-wave digit-pair ct * pt (0 disables)
    long long v;                                    // 2048-bit keys, 53-bit exponents: 0.22 ms up to 256, 0.31 / 0.43 at 512 / 1 024 against 0.37 / 0.49
    return knob_tune("lat_mul_pp", &v) ? (size_t)v : 4 * ncu;
}
static size_t lat_mul_rl_max(size_t ncu) {          // PAI_TUNE lat_mul_rl: largest batch of the wave-pair small-batch ct * pt (0 disables)
    long long v;
    return knob_tune("lat_mul_rl", &v) ? (size_t)v : 2 * ncu;
}
// PAI_TUNE ctmul_mid_min / ctmul_mid_max: batch range of ct x pt on lane-group digit pairs at keys <= 2048 bits (max 0 disables).
// Measured with 53-bit exponents (profiles/r05/ctmul_mid.jsonl): 2048-bit keys 1.4 - 1.5 ms flat up to 16 384 ciphertexts (one wave of
// 16 per SIMD), 2.9 ms at 32 768, against 2.2 / 4.1 ms of the small-batch kernels at 8 192 / 16 384 and 4.7 ms of the one-element-per-lane
// engine up to 65 536 (behind below ~5 000 and from ~55 000); 1024-bit keys 0.55 - 0.63 / 0.9 ms against 0.84 - 1.25 / 1.27
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
    if (latency_paths_forced_off()) return 0;
    return n_bits > 900 && n_bits <= 2048 ? 192 * ncu : 0;
}
// pai_ct_pow2 goes through the digit engine from this batch size on (PAI_POW2_DIGIT_MIN) when the largest shift is
// at least POW2_DIGIT_MIN_SHIFT
constexpr int POW2_DIGIT_MIN_SHIFT = 8;
static size_t pow2_digit_min_elements(size_t ncu) {
    if (const char* env = std::getenv("PAI_POW2_DIGIT_MIN")) return (size_t)std::strtoull(env, nullptr, 10);
    return 64 * ncu;
}

// ---- ct + ct ------------------------------------------------------------------------------------------------------------------
// Largest ct + ct batch on the latency geometry (PAI_LAT_ADD_MAX, 0 disables).  Measured at 2048-bit keys
// (profiles/r04/lat_add_probe.jsonl): wire-form a b 30 against 60 us up to 1 024 elements (39 / 65 at 2 048, level at 4 096), the
// tagged single product 29 against 35 us up to 1 024 (level at 2 048), aligned additions with shifts up to 13: 0.18 against 0.44 ms
// up to 1 024, 0.31 / 0.45 at 4 096 — hence the callers' scale factors 2 (wire form) / 1 (tagged) / 4 (aligned, pow2) / 2 (raw encrypt)
static size_t lat_add_max(size_t ncu) {
    if (const char* env = std::getenv("PAI_LAT_ADD_MAX")) return (size_t)std::strtoull(env, nullptr, 10);
    return 4 * ncu;
}
// wire-form ct + ct stays on one integer per wavefront up to this multiple of lat_add_max: measured against the
// most-significant-limb-first product on lane groups (profiles/r06/ctadd_msb_sweep.jsonl, 256 CUs, us per call, latency / lane groups):
// 1024-bit keys 29.3 / 35.5 at 6 144 (36.1 / 35.7 at 7 168); 2048 bits 48.9 / 52.2 at 5 120 (55.7 / 52.1 at 6 144); 3072 bits
// 51.2 / 51.4 at 2 048 (73.6 / 50.6 at 3 072); 4096 bits 88.1 / 90.0 at 4 096 (118 / 96 at 5 120)
static int lat_add_wire_scale(int key_bits) {
    return key_bits <= 1024 ? 6 : key_bits <= 2048 ? 5 : key_bits <= 3072 ? 2 : key_bits <= 4096 ? 4 : 2;
}

// ---- the switch points of one key, for tests and probes (pai_path_edges) --------------------------------------------------------
// every batch size E at which the path of `op` (0 decrypt, 1 DJN encrypt, 2 ct x pt, 3 ct + ct) may change between N = E and
// N = E + 1 on a device of ncu compute units (a superset: a path a key cannot take leaves its edge in the list)
static std::vector<size_t> path_edges(int op, int key_bits, size_t ncu) {
    std::vector<size_t> e;
    const int prime_bits = key_bits / 2;
    switch (op) {
    case 0:
        e = {lat_pp_max(ncu, 1, key_bits) / 2, lat_pp_max(ncu, 2, key_bits) / 2, lat_rl_max(ncu), lat_dense_min(ncu) - 1,
             dec_mid_min(ncu, prime_bits) - 1, dec_mid_max(ncu, prime_bits), latency_max_elements(LAT_DEC, key_bits, ncu)};
        break;
    case 1:
        e = {lat_enc_tree_max(ncu), enc_mid_min(ncu) - 1, enc_mid_max(ncu, key_bits), latency_max_elements(LAT_ENC, key_bits, ncu)};
        break;
    case 2:
        e = {lat_mul_pp_max(ncu), lat_mul_rl_max(ncu), ctmul_mid_min(ncu, key_bits) - 1, ctmul_mid_max(ncu, key_bits),
             latency_max_elements(LAT_MUL, key_bits, ncu)};
        break;
    case 3:
        e = {lat_add_max(ncu), 2 * lat_add_max(ncu), 4 * lat_add_max(ncu), (size_t)lat_add_wire_scale(key_bits) * lat_add_max(ncu),
             pow2_digit_min_elements(ncu) - 1};
        break;
    default:
        break;
    }
    std::sort(e.begin(), e.end());
    e.erase(std::unique(e.begin(), e.end()), e.end());
    e.erase(std::remove_if(e.begin(), e.end(), [](size_t v) { return v == 0 || v >= ((size_t)1 << 26); }), e.end());
    return e;
}
