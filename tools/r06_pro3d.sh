#!/bin/bash
timeout 900 python -m pytest tests/test_block_fuse_gpu.py tests/test_conv3d_fl_gpu.py -x -q 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_nets3d_gpu.py tests/test_step3d_parity_gpu.py -x -q 2>&1 | tail -3
run() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/$* : /"; }
for i in 1 2 3; do
  run ARCO_BLOCK_FUSE=0
  run ARCO_BLOCK_FUSE=1
done
