#!/bin/bash
# round 5, GPU call 4: pair_mul with the renamed window / compile-time forms: parity (key sizes, tables, ct x pt) and kernel times
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_keysizes.py tests/test_gpu_paillier_abi.py -m gpu -q -x -k "pair or g_factored or other_key_sizes or ct_mul or ct_add or addn or multiexp or tile" > gpurun_out/r05_t4.log 2>&1; tail -5 gpurun_out/r05_t4.log
timeout 600 python tools/keysize_sweep.py --bits 2048 3072 4096 2>&1 | tee gpurun_out/r05_keysize_sweep4.jsonl | cut -c1-900
