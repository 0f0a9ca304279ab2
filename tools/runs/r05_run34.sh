#!/bin/bash
# round 5, GPU call 34: is the role rotation worth anything? (PAI_X_NOROT: experiment switch, removed afterwards)
cd "$(dirname "$0")/../.."
timeout 600 python tools/lat_pp_probe.py 2048 wide 2>&1 | grep bits | head -6 | cut -c1-100
echo norot
PAI_X_NOROT=1 timeout 600 python tools/lat_pp_probe.py 2048 wide 2>&1 | grep bits | head -6 | cut -c1-100
