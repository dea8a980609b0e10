#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2 3; do for v in 0 1; do
  echo "ARCO_LISTS_SIDE=$v lits: $(ARCO_LISTS_SIDE=$v python tools/ab_modes3d.py 3 lits 2>&1 | tail -1)"
done; done
for rep in 1 2; do for v in 0 1; do
  echo "ARCO_LISTS_SIDE=$v la: $(ARCO_LISTS_SIDE=$v python tools/ab_modes3d.py 3 2>&1 | tail -1)"
done; done
