#!/bin/bash
# round 5: which part of the in-backward check removes the hazard (the canary fill or the host synchronisation), and what does a
# stream-ordered snapshot of the adjoint's output hold when the step fails
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
N=${N:-150}
run() { tag=$1; shift; env "$@" timeout 900 python tools/debug/self_consistency.py $N 4 g > gpurun_out/r05_selfc_$tag.txt 2>&1
        echo "$tag: $(grep -c '^trial' gpurun_out/r05_selfc_$tag.txt) trials, off: $(grep '^trial' gpurun_out/r05_selfc_$tag.txt | grep -c 'params off'), lerp events: $(grep -c 'LERP4' gpurun_out/r05_selfc_$tag.txt)"
        grep -A6 'LERP4' gpurun_out/r05_selfc_$tag.txt | head -24 | cut -c1-1100; }
export ARCO_SIDE_SYNC=0
run c_plain SC_CHECK=0
run c_canary_only SC_CHECK=0 SC_CANARY=1
run c_sync_only SC_CHECK=1 SC_CANARY=0
run c_deferred SC_CHECK=2 SC_CANARY=0
run c_deferred_canary SC_CHECK=2 SC_CANARY=1
run c_deferred_dbg0 SC_CHECK=2 SC_CANARY=1 SC_DBG_LERP=0
run c_deferred_dbg1 SC_CHECK=2 SC_CANARY=1 SC_DBG_LERP=1
run c_deferred_dbg2 SC_CHECK=2 SC_CANARY=1 SC_DBG_LERP=2
