#!/bin/bash
timeout 1500 python -m pytest tests/test_step3d_parity_gpu.py tests/test_half_gpu.py tests/test_configs_at_size_gpu.py -x -q -k "3d or cfg3 or cfg5 or f16 or volume" 2>&1 | tail -3
run3() { env "$@" EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 timeout 600 python tools/bench3d.py 2 2>&1 | grep "3D step" | sed "s/^/LA $* : /" | cut -c1-120; }
run3 ARCO_WGRAD_SIDE=0; run3 X=1; run3 ARCO_WGRAD_SIDE=0; run3 X=1
