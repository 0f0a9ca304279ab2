#!/bin/bash
# Builds variants of libpaillier_hip.so that differ only in the scheduler flags of the digit-pair kernels' translation
# units, for same-box A/B timing (PAI_NATIVE_LIB=<variant> python bench.py).  Usage: bash tools/build_variants.sh
set -e
cd "$(dirname "$0")/.."
python -m pailliercryptolib_python_amd.build > /dev/null
C=pailliercryptolib_python_amd/csrc
OUT=pailliercryptolib_python_amd/lib/alt
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1048576"
OTHERS=$(ls $C/build/*.o | grep -v padic_dec_kernels | grep -v padic_enc_kernels)
build() {  # tag, dec flags, enc flags, [1 = the flags also reach paillier_capi.hip (host constants)]
  OTH="$OTHERS"
  hipcc $BASE $2 -c $C/padic_dec_kernels.hip -o $OUT/dec_$1.o &
  hipcc $BASE $3 -c $C/padic_enc_kernels.hip -o $OUT/enc_$1.o &
  if [ "${4:-0}" = "1" ]; then
    hipcc $BASE $2 -c $C/paillier_capi.hip -o $OUT/capi_$1.o &
    OTH="$(echo $OTHERS | tr ' ' '\n' | grep -v paillier_capi) $OUT/capi_$1.o"
  fi
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$1.so $OTH $OUT/dec_$1.o $OUT/enc_$1.o
  rm -f $OUT/dec_$1.o $OUT/enc_$1.o $OUT/capi_$1.o
}
# Edit the list below for the experiment at hand; every variant lands in lib/alt/lib_<tag>.so.  Then, in ONE gpurun
# call (boxes differ by a few %):  for v in plain x plain; do PAI_NATIVE_LIB=$PWD/$OUT/lib_$v.so python bench.py ...; done
# Results so far (k_dec_a / k_encrypt per 2^20, 2048-bit key, same box): scheduler strategy max-ilp 508 vs 487 ms;
# trackers / metric-bias=0 within noise; modulus limbs from LDS instead of SGPRs 508 vs 488; second half of the
# squaring unrolled 476 vs 486; product loops unrolled as well 511; sliding window 5 / 6 / 7 bits 486 / 478 / 489;
# 12-row blocks for the 72-limb products 68.7 vs 67.4 ms (encrypt), 88.7 vs 90.0 ms (ct * pt); touching the next
# window's table lines one product ahead in k_encrypt_padic 72.1 vs 67.0 ms; first result digit written into the
# quotient-digit buffer with the LDS buffers rotating (no 36 registers held across the second half) 488 vs 481 ms;
# 56 / 72-limb decrypt (per 65536): modulus in SGPRs 174 vs 166 / 364 vs 391 ms; unpipelined chunk loop at 56 limbs 667 ms;
# 12-row blocks in the 72-limb decrypt kernel 945 vs 366 ms (spills); -O2, gcn-iterative-ilp within noise, post-RA
# scheduler off 501 vs 477 ms; 192-thread workgroups with the quotient digits in LDS for 56-limb decrypt 220 vs 166 ms per 65536;
# 56 limbs: squaring as product 146, + LDS-qualified accesses 151, + modulus in SGPRs 142 (adopted) vs 166 ms.
# Round 2 (tools/variant_dec.sh builds one variant of padic_dec_kernels.hip in 45 s): table digits fetched two row blocks
# ahead in the 36-limb multiplications 481.9 vs 474.9 ms (the extra 24 registers cost more than the latency they hide);
# 56 limbs: rolled two-half squaring 143.9 -> 122.2, + switch-dispatched symmetric first half 112.7 (adopted); 72 limbs:
# rolled 320.9 -> 284.4 (adopted), symmetric first half 383.4 (rejected: scratch 1356 -> 2752 B/lane).
build plain "" "" &
build trk "-mllvm -amdgpu-use-amdgpu-trackers" "-mllvm -amdgpu-use-amdgpu-trackers" &
wait
ls -la $OUT
# Fused product rule (tools/variant_enc.sh builds enc + dec variants; -DPAI_FUSED_*=false / -DPADIC_FUSED_56/72=false give
# the scratch-parked forms back): per 65536 — 4096-bit decrypt 282.9 -> 257.5, 3072-bit decrypt 117.4 -> 113.0, 2048-bit
# ct*pt 5.36 -> 4.82; DJN encrypt per 2^20: 8-row fused 69.8 (no gain), 12-row fused 64-67 (adopted, -DPAI_ENCRYPT_U);
# rejected: -DPADIC_U72=12 (1127 ms, spills), -DPADIC_SQR_SYM_MAX_NL=72 fused (887 ms), -DPAI_XLDS_ENCRYPT=true (70.2),
# fused r^n at 36 limbs (29.0 vs 25.1 ms), 12-row r^n (within noise).
# Lane-group digit pairs (tools/variant_tu.sh <tag> pair_kernels <flags>), DJN encrypt per 65536 at 3072 / 4096 bits:
# default (28x4 7 rows / 18x8 6 rows, prefetch, 2 waves per SIMD at 8 lanes) 17.9 / 44.3 ms; -DPAIR_PREFETCH=0 19.3 / 45.4;
# -DPAIR_WAVES_T8=1 - / 49.3; -DPAIR_G112=Geo<28,4,4,false> 18.9; Geo<14,8,7,false> 18.5; Geo<28,4,2,false> 18.7;
# -DPAIR_G144=Geo<18,8,9,false> 44.0; Geo<18,8,3,false> 44.1; Geo<36,4,6,false> without prefetch 58.1 (spills).
# PAI_DISABLE_PAIR=1 (products modulo n^2): 24.5 / 61.7.  Finishing as extra modes of k_encrypt instead of k_pair_finish
# cost k_encrypt<36x8> 61 -> 94 ms.
# Row-block size of the 36x4 lane-group kernels (tools/variant_tu.sh u12 geo_36x4 -DPAI_U_36X4=12): ct+ct 3.43 (6 rows) /
# 3.65 (12) / 3.47 (4) ms per 2^20, k_pow2 (delta 12) 22.7 / 24.7 / 21.3, k_add_aligned 13.7 / 13.8 / 13.6: 6 rows kept.
# 144-limb pair kernel on 4 lanes x 36 with the modulus slice re-read from LDS (-DPAIR_G144=Geo<36,4,6,true>): 55.8 ms
# without / 57.8 with the table prefetch, 12-row blocks 93.4 (1.4-1.8 KB of scratch per lane either way) vs 45.7 for 8 x 18.
# pai_ct_multiexp lane order (PAI_MEXP_BY_ROWS=1: lanes of a wave walk the rows of one output column instead of the columns
# of one row): 64x1024 @ 1024x64 0.0986 vs 0.0967 s, 1024x64 @ 64x64 0.1004 vs 0.1009 s — no difference, columns kept.
# One wave per SIMD for the 8-lane exponentiation kernels (-DPAI_WAVES_T8=1 on geo_28x8 / geo_36x8): k_mexp<36x8> 101 -> 54 ms
# (adopted), k_modexp_var_win<28x8> 22.1 -> 20.6 ms per 65536 (adopted), k_modexp_var_win<36x8> 32.2 -> 34.4, k_mexp<28x8>
# 31.7 -> 33.8, k_encrypt<36x8> 61.5 -> 59.4 (kept at two).  Modulus from LDS in k_mexp<36x8> (-DPAI_MEXP_NMLDS=true) 108 ms;
# digit / sign of the next member prefetched 120 ms.
# Row-block size of the 36x8 kernels (-DPAI_U_36X8): ct*pt at 4096 bits 31.5 (6 rows) / 33.5 (4) / 35.0 (3) / 43.4 (12) ms per 65536.
# pai_ct_multiexp window width for dot of 2^20 (one column): auto = 3 bits 59.7 ms; PAI_MEXP_WBITS=2 / 4 / 5: 66.0 / 63.2 / 78.1.
