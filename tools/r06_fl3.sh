#!/bin/bash
# on the GPU box: the depth-walking 3x3x3 kernel - parity, then per-level timing for each tile height and column height
for c in 92 93 94; do echo "cfg $c"; ARCO_CONV3D_FL_CFG=$c timeout 600 python -m pytest tests/test_conv3d_fl_gpu.py -x -q -k "equals_igemm" 2>&1 | tail -3; done
for c in 92 93 94; do echo "cfg $c"; ARCO_CONV3D_FL_CFG=$c timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -6; done
for s in 4 7 14 28; do echo "cfg 92 S=$s"; ARCO_CONV3D_DW_S=$s ARCO_CONV3D_FL_CFG=92 FL_SHAPES=2 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -2; done
for s in 7 14 28; do echo "cfg 93 S=$s"; ARCO_CONV3D_DW_S=$s ARCO_CONV3D_FL_CFG=93 FL_SHAPES=2 timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -2; done
