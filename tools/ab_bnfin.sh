#!/bin/bash
# A/B on ONE box: BatchNorm finalize with 4 waves per channel (0) / 16 waves for rows of >= N slabs (default 4096; 2048)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for w in 0 4096 2048; do
  echo "wide=$w LA: $(ARCO_BN_FINALIZE_WIDE=$w GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 python tools/bench3d.py 2 2>&1 | tail -1 | cut -c75-130)"
  echo "wide=$w LiTS f16: $(ARCO_BN_FINALIZE_WIDE=$w GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1 | cut -c75-130)"
  echo "wide=$w 2-D: $(ARCO_BN_FINALIZE_WIDE=$w python bench.py --steps 30 --warmup 5 --no_subs --no_cpu_baseline --k2_0_steps 0 --sustain_s 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
done
