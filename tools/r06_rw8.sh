#!/bin/bash
# on the GPU box: eight MFMA waves in the resident-weights 3x3 kernel - per launch (incl. the 32-channel form) and the 2-D step, same box, alternating
for v in 0 2; do echo "ARCO_CONV_RW8=$v"; ARCO_CONV_RW8=$v timeout 300 python tools/micro/rw_bench.py 20 2>&1 | sed -n 6,7p; done
run2() { l=$1; shift; env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D', sys.argv[1], d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" $l; }
for i in 1 2 3 4; do run2 rw8=0 ARCO_CONV_RW8=0; run2 rw8=1 ARCO_CONV_RW8=1; done
run2 rw8=2 ARCO_CONV_RW8=2; run2 rw8=2 ARCO_CONV_RW8=2
