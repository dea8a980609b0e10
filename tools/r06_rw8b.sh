#!/bin/bash
run2() { l=$1; shift; env "$@" timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D', sys.argv[1], d['ms_per_step'], d['sustained_ms_per_step'], d['k2_0_ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" $l; }
for i in 1 2 3 4 5; do run2 rw8=0 ARCO_CONV_RW8=0; run2 rw8=1 ARCO_CONV_RW8=1; run2 rw8=2 ARCO_CONV_RW8=2; done
