#!/bin/bash
# on the GPU box: same-box A/B of the activation split - round-to-nearest library (rounds 2-5) against the new default: 2-D headline, LA, Cityscapes-shaped
root=$(pwd); R=$root/arco_amd/lib/libarco_hip_rne.so
run2() { l=$1; shift; env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D', sys.argv[1], d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'], d['roofline']['avg_launch_ms'])" $l; }
run3() { l=$1; shift; env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | cut -c75-110 | sed "s/^/LA $l : /"; }
for i in 1 2 3 4; do run2 rne ARCO_LIB=$R; run2 new X=1; done
for i in 1 2 3; do run3 rne ARCO_LIB=$R; run3 new X=1; done
