#!/bin/bash
# on the GPU box: hconv_fc_kernel defaults - parity, per-level timing, LiTS step A/B (same box, alternating)
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py tests/test_half_gpu.py -x -q 2>&1 | tail -3
for nv in 1 2; do HALF=1 timeout 300 python tools/micro/fl_bench.py $nv 2>&1 | tail -6 | cut -c1-130; done
run() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 ACT_DTYPE=f16 timeout 600 python tools/bench3d.py 1 160 160 96 2>&1 | grep "3D step" | sed "s/^/$* : /"; }
for i in 1 2 3; do
  run ARCO_HCONV_FC=0
  run ARCO_HCONV_FC=1
done
