#!/bin/bash
# on the GPU box: schedule knobs of the LA step re-measured with the new 3x3x3 kernels (same box, alternating)
run() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/$* : /" | cut -c1-150; }
for i in 1 2; do
  run ARCO_WGRAD_SIDE=0
  run ARCO_WGRAD_SIDE=1
  run ARCO_WGRAD_SIDE=2
  run ARCO_WGRAD_SIDE=3
  run ARCO_TEACHER_SIDE=2
  run ARCO_LISTS_SIDE=0
done
