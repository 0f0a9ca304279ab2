#!/bin/bash
# round 5, GPU call 47: mid-size DJN encryption on 4-lane digit pairs reading the digit engine's table, 2048- and 1024-bit keys
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for b in 2048 1024; do timeout 900 python tools/enc_mid_probe.py $b 2>&1 | grep -E "bits|rror"; done | tee gpurun_out/r05_enc_mid47.jsonl
