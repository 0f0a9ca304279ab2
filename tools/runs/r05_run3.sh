#!/bin/bash
# round 5, GPU call 3: whole GPU suite after the switch / knob clean-up (durations), smoke
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=50 > gpurun_out/r05_gputest3.log 2>&1
tail -70 gpurun_out/r05_gputest3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
