#!/bin/bash
# on the GPU box: tile-count knobs of the 2-D step re-measured at the end of round 6 (light bench, same box, baseline between the variants)
run2() { env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D $*', d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'])"; }
for i in 1 2; do
  run2 BASE=1
  run2 ARCO_CONV_SP_TILES=128
  run2 ARCO_CONV_SP_TILES=256
  run2 BASE=1
  run2 ARCO_WGRAD_TARGET=384
  run2 ARCO_WGRAD_TARGET=256
  run2 BASE=1
  run2 ARCO_WGRAD1_TARGET=384
  run2 ARCO_WGRAD1_TARGET=768
  run2 BASE=1
  run2 ARCO_IGEMM_WANT3=256
  run2 ARCO_IGEMM_WANT3=768
done
