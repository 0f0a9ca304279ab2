#!/bin/bash
# round 5, GPU call 40: key sizes on either side of the chain layouts' switch (small-batch decryption and ct * pt against the oracle)
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_gpu_keysizes.py -m gpu -q -x -k "boundaries" --durations=5 2>&1 | tail -12
