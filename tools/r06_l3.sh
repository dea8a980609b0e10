#!/bin/bash
# three-level 3-D head: unit parity, the 3-D suites, A/B of the LA and LiTS-f16 steps (ARCO_HEAD3D_LEVELS = 2 | 3)
o=gpurun_out; mkdir -p $o
timeout 900 python -m pytest tests/test_head_gpu.py -x -q -m gpu -k "three_level" 2>&1 | tail -15
timeout 1800 python -m pytest tests/test_nets3d_gpu.py tests/test_step3d_parity_gpu.py tests/test_half_gpu.py -x -q -m gpu 2>&1 | tail -8
for lv in 2 3 2 3; do
  echo "levels=$lv LA: $(ARCO_HEAD3D_LEVELS=$lv GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 timeout 600 python tools/bench3d.py 2 2>&1 | tail -1 | cut -c1-200)"
  echo "levels=$lv LiTS f16: $(ARCO_HEAD3D_LEVELS=$lv GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 timeout 600 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1 | cut -c1-200)"
done
