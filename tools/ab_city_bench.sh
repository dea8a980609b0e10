#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do for v in 3 4; do
  echo "ARCO_TEACHER_SIDE=$v: $(ARCO_TEACHER_SIDE=$v python bench.py --sub cityscapes_19c_512x1024 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done; done
