#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_path_edges.py tests/test_gpu_keysizes.py tests/test_gpu_api.py -m gpu -q -x -k "pow2 or aligned or edges or sub or chains or staged" 2>&1 | tail -5
timeout 200 python tools/fuzz_gpu.py 60 777 2>&1 | tail -2
