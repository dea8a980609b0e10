#!/bin/bash
# A/B on ONE box: minimum work items per launch of the pipelined 3x3 kernels (which tile height a shape gets)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
  for t in 192 320 512 768; do
    echo "ARCO_CONV_SP_TILES=$t rep $rep: $(ARCO_CONV_SP_TILES=$t python tools/prof_step.py 120 2>&1 | tail -1)"
  done
done
