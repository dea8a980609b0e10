#!/bin/bash
# on the GPU box: LA step A/B over the 3x3x3 kernel choices (same box, alternating), then the step timeline
mkdir -p gpurun_out; o=gpurun_out
run() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/$* : /"; }
for i in 1 2; do
  run ARCO_CONV3D_FL=0
  run ARCO_CONV3D_DW=0
  run ARCO_CONV3D_DW=1
  run ARCO_CONV3D_DW=2
done
timeout 600 python tools/step_timeline3d.py > $o/r06_timeline3d_fl.txt 2>&1; tail -45 $o/r06_timeline3d_fl.txt
