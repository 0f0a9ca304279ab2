#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py tests/test_gpu_path_edges.py -m gpu -q -x -k "by_division or add_forms" 2>&1 | tail -2
timeout 300 python tools/fuzz_gpu.py 90 4242 2>&1 | tail -1
