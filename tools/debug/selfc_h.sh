#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { timeout 900 python tools/debug/self_consistency.py 80 ${MODE:-4} g > gpurun_out/selfc_$1.txt 2>&1; echo "$1: $(grep -c trial gpurun_out/selfc_$1.txt) trials, off: $(grep trial gpurun_out/selfc_$1.txt | grep -c 'params off')"; }
ARCO_SIDE_SYNC=1 run mode4_latewait
MODE=3 ARCO_SIDE_SYNC=1 run mode3_sync
python tools/ab_modes.py 4 -4 3
