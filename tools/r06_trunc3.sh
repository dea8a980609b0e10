#!/bin/bash
# on the GPU box: the round-to-nearest + truncation split as the default: whole-net accuracy, stamps, per-launch A/B
R=$PWD/arco_amd/lib/libarco_hip_rne.so
timeout 300 python tools/debug/dbg_g18.py 2>&1 | tail -3
ARCO_LIB=$PWD/arco_amd/lib/libarco_hip_fcclock.so timeout 300 python tools/micro/fc_clock.py 2>&1 | tail -4 | cut -c1-400
for i in 1 2; do
  echo "== fl_bench rne"; ARCO_LIB=$R FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -4 | cut -c60-125
  echo "== fl_bench new"; FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -4 | cut -c60-125
done
echo "== rw_bench rne"; ARCO_LIB=$R timeout 300 python tools/micro/rw_bench.py 20 2>&1 | head -6
echo "== rw_bench new"; timeout 300 python tools/micro/rw_bench.py 20 2>&1 | head -6
timeout 900 python -m pytest tests/test_split_mma_gpu.py tests/test_conv3d_fl_gpu.py tests/test_nets3d_gpu.py -x -q 2>&1 | tail -3
