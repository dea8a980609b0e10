#!/bin/bash
# A/B on ONE box: the default 2-D step (and the LA 3-D step) with the weight gradients on the side stream vs in line, alternating.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for side in 0 1 2 3; do
    echo "ARCO_WGRAD_SIDE=$side rep $rep 2d: $(ARCO_WGRAD_SIDE=$side python tools/prof_step.py 120 2>&1 | tail -1)"
  done
done
for side in 0 1 2 3; do
  echo "ARCO_WGRAD_SIDE=$side 3d LA: $(CONV_MMA=f32x3 EQV_PASS=1 ARCO_WGRAD_SIDE=$side python tools/bench3d.py 2 2>&1 | tail -1)"
done
