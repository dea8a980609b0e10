#!/bin/bash
# A/B on ONE box: minimum workgroups per launch of the split-bf16 implicit GEMMs (tile choice of the deep V-Net levels)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for w in 512 256 128 1024; do
  echo "ARCO_IGEMM_WANT3=$w LA: $(ARCO_IGEMM_WANT3=$w GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 python tools/bench3d.py 2 2>&1 | tail -1 | cut -c60-110)"
done
done
