#!/bin/bash
# A_T = 5 / 6 tiles of hconv_fc_kernel: parity of every forced shape, per-level timing with and without them, the LiTS-f16 step A/B
o=gpurun_out; mkdir -p $o
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py -x -q -m gpu -k "f16" 2>&1 | tail -3
for big in 0 1 0 1; do
  echo "== ARCO_HCONV_FC_BIG=$big"
  ARCO_HCONV_FC_BIG=$big HALF=1 timeout 300 python tools/micro/fl_bench.py 2 2>&1 | grep "3x3x3"
done
for big in 0 1 0 1; do
  echo "BIG=$big LiTS f16: $(ARCO_HCONV_FC_BIG=$big GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 timeout 600 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1 | cut -c1-200)"
done
