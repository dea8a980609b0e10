#!/bin/bash
# hazard 2 (profiles/r04_notes.md section 8) under runtime switches, without the host-side wait; 80 amplified trials each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
run() { timeout 900 python tools/debug/self_consistency.py 80 4 g > gpurun_out/selfc_$1.txt 2>&1; echo "$1: $(grep -c trial gpurun_out/selfc_$1.txt) trials, off: $(grep trial gpurun_out/selfc_$1.txt | grep -c 'params off')"; }
export ARCO_SIDE_SYNC=0
run base
HSA_ENABLE_SDMA=0 run no_sdma
HIP_FORCE_DEV_KERNARG=0 run kernarg_host
HIP_FORCE_DEV_KERNARG=1 run kernarg_dev
