#!/bin/bash
# round 5, GPU call 44: mid-size decryption at 3072- and 4096-bit keys (4 lanes x 14 / 18 limbs)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for b in 3072 4096; do timeout 900 python tools/dec_mid_probe.py $b 2>&1 | grep -E "bits|rror"; done | tee gpurun_out/r05_dec_mid44.jsonl
