#!/bin/bash
# round 5, GPU call 42: mid-size decryption with stage A on lane-group digit pairs (4 lanes x 9 limbs): parity + time against the other paths
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python tools/dec_mid_probe.py 2048 2>&1 | grep -E "bits|Error|error" | tee gpurun_out/r05_dec_mid42.jsonl
