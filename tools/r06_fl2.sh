#!/bin/bash
# on the GPU box: the pipelined 3x3x3 kernel in the V-Net tests and in the LA / LiTS steps, A/B against igemm_kernel in one call
mkdir -p gpurun_out; o=gpurun_out
timeout 1500 python -m pytest tests/test_conv3d_fl_gpu.py tests/test_nets3d_gpu.py tests/test_step3d_parity_gpu.py tests/test_split_mma_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for on in 0 1; do
    echo "ARCO_CONV3D_FL=$on"
    ARCO_CONV3D_FL=$on EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step"
  done
done
