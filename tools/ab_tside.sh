#!/bin/bash
# usage: bash tools/ab_tside.sh [modes...]   (default 0 2 3 4) - alternating runs of the default 2-D step per ARCO_TEACHER_SIDE mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
modes=${@:-0 2 3 4}
for rep in 1 2 3; do
for v in $modes; do
  echo "ARCO_TEACHER_SIDE=$v: $(ARCO_TEACHER_SIDE=$v python tools/prof_step.py 120 2>&1 | tail -1)"
done
done
