#!/bin/bash
# round 5: the fixed product kernel (single v_mul_f32 for the mixed weights) without the host-side wait
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
N=${N:-400}
run() { tag=$1; mode=$2; shift; shift; env "$@" timeout 1200 python tools/debug/self_consistency.py $N $mode g > gpurun_out/r05_selfc_$tag.txt 2>&1
        echo "$tag: $(grep -c '^trial' gpurun_out/r05_selfc_$tag.txt) trials, off: $(grep '^trial' gpurun_out/r05_selfc_$tag.txt | awk '{ if ($4+0 > 1e-5) print }' | wc -l), lerp events: $(grep -c 'LERP4' gpurun_out/r05_selfc_$tag.txt)"; }
export ARCO_SIDE_SYNC=0
run e_fixed_m4_amplified 4 SC_CHECK=0
run e_fixed_m3_amplified 3 SC_CHECK=0
run e_fixed_m4_deferred 4 SC_CHECK=2 SC_CANARY=1
run e_old_isa_m4 4 SC_CHECK=0 SC_ISA=v0_product
