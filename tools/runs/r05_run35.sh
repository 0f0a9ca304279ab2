#!/bin/bash
# round 5, GPU call 35: how far the pipeline carries (several workgroups per CU, several rounds) at 2048 / 3072 / 4096-bit keys
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for b in 2048 3072 4096; do timeout 600 python tools/lat_pp_probe.py $b wide 2>&1 | grep bits; done | tee gpurun_out/r05_lat_pp35.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['bits'], d['N'], 'pp', d['pp']['k_dec_a_ms'], 'rl', d['rl']['k_dec_a_ms'], 'win', d['window']['k_dec_a_ms'], 'default', d['default']['k_dec_a_ms'], all(d[k]['ok'] for k in ('pp','rl','window','default')))
"
