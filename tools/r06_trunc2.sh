#!/bin/bash
# on the GPU box: the new activation split as the default - whole GPU suite, then same-box A/B against the round-to-nearest library (2-D headline, LA)
root=$(pwd); o=$root/gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $o/r06g_tests.log; cat $o/r06g_tests.log
run2() { env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D', sys.argv[1], d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'], d['roofline']['avg_launch_ms'])" "$2"; }
run3() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | cut -c75-110 | sed "s/^/LA $2 : /"; }
for i in 1 2 3; do run2 ARCO_LIB=$root/arco_amd/lib/libarco_hip_rne.so rne; run2 X=1 new; done
for i in 1 2 3; do run3 ARCO_LIB=$root/arco_amd/lib/libarco_hip_rne.so rne; run3 X=1 new; done
