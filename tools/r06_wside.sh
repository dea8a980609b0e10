#!/bin/bash
# on the GPU box: ARCO_WGRAD_SIDE 0 against 3 (same box, alternating): LA, LiTS-f16, the 2-D headline step
run3() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/LA $* : /" | cut -c1-120; }
runl() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 ACT_DTYPE=f16 timeout 600 python tools/bench3d.py 1 160 160 96 2>&1 | grep "3D step" | sed "s/^/LiTS $* : /" | cut -c1-120; }
run2() { env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D $*', d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'])"; }
for i in 1 2 3; do run3 ARCO_WGRAD_SIDE=0; run3 ARCO_WGRAD_SIDE=3; done
for i in 1 2; do runl ARCO_WGRAD_SIDE=0; runl ARCO_WGRAD_SIDE=3; done
for i in 1 2; do run2 ARCO_WGRAD_SIDE=0; run2 ARCO_WGRAD_SIDE=3; done
