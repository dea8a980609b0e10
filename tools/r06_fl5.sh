#!/bin/bash
# on the GPU box: every tile shape of conv3d_fl_kernel per level at 2 and 4 volumes (calibration of the launch cost model)
for nv in 2 4; do
  for c in 0 44 34 24 14 42 32 22 12; do
    echo "nv $nv cfg $c"; ARCO_CONV3D_FL_CFG=$c FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py $nv 2>&1 | tail -4 | sed 's/\[9[0-9]*\] *[0-9.]* us *[0-9.]* TF  //' | cut -c1-110
  done
done
