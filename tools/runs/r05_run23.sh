#!/bin/bash
# round 5, GPU call 23: rocprofv3 kernel stats + PMC passes of the cfg4 / cfg5 bench legs (lane-group pair kernels after round 5)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for c in cfg5 cfg4; do
  bash tools/profile_cmd.sh r05_$c python $PWD/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-configs > gpurun_out/r05_prof23_$c.log 2>&1
  tail -8 gpurun_out/r05_prof23_$c.log
done
