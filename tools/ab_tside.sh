#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2 3; do
for v in 0 2 3; do
  echo "ARCO_TEACHER_SIDE=$v: $(ARCO_TEACHER_SIDE=$v python tools/prof_step.py 120 2>&1 | tail -1)"
done
done
