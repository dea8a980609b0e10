#!/bin/bash
# on the GPU box: the 3x3x3 consumer-side activation re-measured behind the cheaper split (the loaders have slack now)
for nv in 2 4; do timeout 300 python tools/micro/fl_pro_bench.py $nv 2>&1 | tail -4; done
run() { l=$1; shift; env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | cut -c75-110 | sed "s/^/LA $l : /"; }
for i in 1 2 3; do run staged ARCO_BLOCK_FUSE3D=0; run fused ARCO_BLOCK_FUSE3D=1; done
