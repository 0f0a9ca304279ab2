#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_paillier_abi.py -m gpu -q -x -k "by_division" 2>&1 | tail -3
timeout 300 python tools/ctadd_div_time.py 2>&1 | tail -1
