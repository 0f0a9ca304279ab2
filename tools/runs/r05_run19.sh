#!/bin/bash
# round 5, GPU call 19: how much of k_encrypt_padic is the table gather (random r against cache-resident working sets)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python tools/enc_table_locality.py 2048 2>&1 | grep bits | tee gpurun_out/r05_enc_locality19.jsonl
