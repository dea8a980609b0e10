#!/bin/bash
# on the GPU box: the chunk-granular 3x3x3 kernel - parity for the default choice and every forced tile shape, then timing
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py -x -q 2>&1 | tail -4
for nv in 2 4; do
  for c in 0 124 114 142 132 122 112; do
    echo "nv $nv cfg $c"; ARCO_CONV3D_FL_CFG=$c FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py $nv 2>&1 | tail -4 | sed 's/\[9[0-9]*\] *[0-9.]* us *[0-9.]* TF  //' | cut -c1-110
  done
done
