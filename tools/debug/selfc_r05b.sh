#!/bin/bash
# round 5: does the hazard reproduce on this box at all, and does the GraphedForward change (private salt / richer key) matter?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
N=${N:-150}
run() { tag=$1; shift; env "$@" timeout 900 python tools/debug/self_consistency.py $N 4 g > gpurun_out/r05_selfc_$tag.txt 2>&1
        echo "$tag: $(grep -c '^trial' gpurun_out/r05_selfc_$tag.txt) trials, off: $(grep '^trial' gpurun_out/r05_selfc_$tag.txt | grep -c 'params off'), lerp events: $(grep -c 'LERP4' gpurun_out/r05_selfc_$tag.txt)"
        grep -A6 'LERP4' gpurun_out/r05_selfc_$tag.txt | head -16 | cut -c1-900; }
export ARCO_SIDE_SYNC=0
run b_new SC_CHECK=1 SC_CANARY=1
cp arco_amd/graphs.py /tmp/graphs_new.py; cp tools/debug/graphs_r04.py.txt arco_amd/graphs.py
run b_old_graphs SC_CHECK=1 SC_CANARY=1
run b_old_graphs_plain SC_CHECK=0
run b_old_graphs_dbg0 SC_CHECK=1 SC_CANARY=1 SC_DBG_LERP=0
cp /tmp/graphs_new.py arco_amd/graphs.py
run b_new2 SC_CHECK=0
