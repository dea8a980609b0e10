#!/bin/bash
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py tests/test_nets3d_gpu.py -x -q 2>&1 | tail -2
run3() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/LA $* : /" | cut -c1-120; }
for i in 1 2 3; do run3 ARCO_CONV3D_FL_NO5=1; run3 X=1; done
