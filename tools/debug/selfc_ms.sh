#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { timeout 900 python tools/debug/self_consistency.py 80 4 g > gpurun_out/selfc_$1.txt 2>&1; echo "$1: $(grep -c trial gpurun_out/selfc_$1.txt) trials, off: $(grep trial gpurun_out/selfc_$1.txt | grep -c 'params off')"; }
ARCO_SIDE_SYNC=0 run nosync_default_stream
ARCO_SIDE_SYNC=0 SC_MAIN_STREAM=1 run nosync_own_main_stream
