#!/bin/bash
# A/B on ONE box: the 3-D steps (LA fp32 / LiTS f16 storage) as the bench sub-records run them (graph_train 1), side stream 0 / 1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for side in 0 1; do
  echo "side=$side LA graph_train=1: $(GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 ARCO_WGRAD_SIDE=$side python tools/bench3d.py 2 2>&1 | tail -1)"
done
for side in 0 1; do
  echo "side=$side LiTS f16 graph_train=1: $(GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 ARCO_WGRAD_SIDE=$side python tools/bench3d.py 1 160 160 96 2>&1 | tail -1)"
done
done
