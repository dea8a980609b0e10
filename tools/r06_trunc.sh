#!/bin/bash
# on the GPU box: the truncation form of the activation split (-DARCO_SPLIT_TRUNC) against the round-to-nearest one
T=$PWD/arco_amd/lib/libarco_hip_trunc.so; TC=$PWD/arco_amd/lib/libarco_hip_trunc_clock.so
echo "== accuracy tests on the truncation library"; ARCO_LIB=$T timeout 900 python -m pytest tests/test_split_mma_gpu.py tests/test_conv3d_fl_gpu.py -x -q 2>&1 | tail -3
echo "== stamps (truncation)"; ARCO_LIB=$TC timeout 300 python tools/micro/fc_clock.py 2>&1 | tail -4 | cut -c1-420
for i in 1 2; do
  echo "== fl_bench rn"; FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -4 | cut -c60-125
  echo "== fl_bench trunc"; ARCO_LIB=$T FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -4 | cut -c60-125
done
echo "== rw_bench rn"; timeout 300 python tools/micro/rw_bench.py 20 2>&1 | head -7
echo "== rw_bench trunc"; ARCO_LIB=$T timeout 300 python tools/micro/rw_bench.py 20 2>&1 | head -7
