#!/bin/bash
# round 5: hand-edited ISA variants of the product kernel (tools/debug/isa), deferred check (no host synchronisation inside backward)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
N=${N:-150}
run() { tag=$1; shift; env "$@" timeout 900 python tools/debug/self_consistency.py $N 4 g > gpurun_out/r05_selfc_$tag.txt 2>&1
        echo "$tag: $(grep -c '^trial' gpurun_out/r05_selfc_$tag.txt) trials, off: $(grep '^trial' gpurun_out/r05_selfc_$tag.txt | grep -c 'params off'), lerp events: $(grep -c 'LERP4' gpurun_out/r05_selfc_$tag.txt)"
        grep -A1 'LERP4' gpurun_out/r05_selfc_$tag.txt | grep -v '^trial\|FINGERPRINT\|^--' | head -${SHOW:-8} | cut -c1-700; }
export ARCO_SIDE_SYNC=0 SC_CHECK=2 SC_CANARY=1
for v in ${VARIANTS:-v0_product v1_nop_after_wait v2_nop_before_mixed v3_markers v4_copies}; do
  run d_$v SC_ISA=$v
done
