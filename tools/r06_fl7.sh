#!/bin/bash
# on the GPU box: parity of the 3x3x3 kernel family, the 3-D test files, per-level timing of the default choice, LA / LiTS step A/B
mkdir -p gpurun_out; o=gpurun_out
timeout 1800 python -m pytest tests/test_conv3d_fl_gpu.py tests/test_nets3d_gpu.py tests/test_step3d_parity_gpu.py tests/test_split_mma_gpu.py tests/test_vnet_norms_gpu.py -x -q 2>&1 | tail -4
for nv in 2 4; do FL_SHAPES=4 timeout 300 python tools/micro/fl_bench.py $nv 2>&1 | tail -4 | sed 's/\[96[0-9]*\] *[0-9.]* us *[0-9.]* TF  //' | cut -c1-110; done
run() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/$* : /"; }
for i in 1 2; do
  run ARCO_CONV3D_FL=0
  run ARCO_CONV3D_FC=0
  run ARCO_CONV3D_FC=1
done
