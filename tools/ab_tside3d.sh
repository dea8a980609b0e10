#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for v in 0 1 2; do
  echo "ARCO_TEACHER_SIDE=$v LA: $(ARCO_TEACHER_SIDE=$v GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 python tools/bench3d.py 2 2>&1 | tail -1 | cut -c60-110)"
  echo "ARCO_TEACHER_SIDE=$v LiTS f16: $(ARCO_TEACHER_SIDE=$v GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1 | cut -c60-110)"
done
done
