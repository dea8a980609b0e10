#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for m in 0 1 2 3 4; do
  timeout 600 python tools/debug/self_consistency.py 300 $m g > gpurun_out/selfc_m$m.txt 2>&1
  echo "mode $m: $(grep -c trial gpurun_out/selfc_m$m.txt) trials, off: $(grep trial gpurun_out/selfc_m$m.txt | awk '{ if ($4+0 > 1e-5) print }' | wc -l)"
  grep trial gpurun_out/selfc_m$m.txt | awk '{ if ($4+0 > 1e-5) print }' | head -3 | cut -c1-260
done
