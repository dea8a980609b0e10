#!/bin/bash
# A/B on ONE box: the 3-D steps as the bench sub-records run them, the row-sparse head's adjoint with fp32 atomics (0) / order-independent fixed point (1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for det in 0 1; do
  echo "det=$det LA: $(ARCO_DET_SCATTER=$det GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 python tools/bench3d.py 2 2>&1 | tail -1)"
  echo "det=$det LiTS f16: $(ARCO_DET_SCATTER=$det GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1)"
done
done
