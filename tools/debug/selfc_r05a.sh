#!/bin/bash
# round 5, first batch of discriminating probes of the two-queue hazard (ARCO_SIDE_SYNC=0, mode 4, amplified by the probes' clones)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
N=${N:-40}
run() { tag=$1; shift; env "$@" timeout 600 python tools/debug/self_consistency.py $N 4 g > gpurun_out/r05_selfc_$tag.txt 2>&1
        echo "$tag: $(grep -c '^trial' gpurun_out/r05_selfc_$tag.txt) trials, off: $(grep '^trial' gpurun_out/r05_selfc_$tag.txt | grep -c 'params off'), lerp events: $(grep -c 'LERP4' gpurun_out/r05_selfc_$tag.txt)"
        grep -A6 'LERP4' gpurun_out/r05_selfc_$tag.txt | head -16 | cut -c1-900; }
export ARCO_SIDE_SYNC=0
run base_canary SC_CHECK=1 SC_CANARY=1
run dbg0 SC_CHECK=1 SC_CANARY=1 SC_DBG_LERP=0
run dbg1_wt_stores SC_CHECK=1 SC_CANARY=1 SC_DBG_LERP=1
run dbg2_loop_weights SC_CHECK=1 SC_CANARY=1 SC_DBG_LERP=2
run one_hw_queue SC_CHECK=1 SC_CANARY=1 GPU_MAX_HW_QUEUES=1
run no_packet_capture SC_CHECK=1 SC_CANARY=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
