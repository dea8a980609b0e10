#!/bin/bash
# on the GPU box: hconv_fc_kernel - parity, the f16 test file, timing per LiTS level for the default and forced tile shapes
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py -x -q -k "f16" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_half_gpu.py -x -q 2>&1 | tail -3
for nv in 1 2; do
  for c in 0 44 34 24 14 42 32 22 12; do echo "nv $nv cfg $c"; HALF=1 ARCO_HCONV_FC_CFG=$c timeout 300 python tools/micro/fl_bench.py $nv 2>&1 | tail -6 | cut -c1-130; done
done
