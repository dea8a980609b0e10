#!/bin/bash
# on the GPU box: first run of the pipelined 3x3x3 kernel - parity test, per-level micro-bench (default and forced tile shapes)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv3d_fl_gpu.py -x -q 2>&1 | tail -15
timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -8
for c in 42 22 24 14 12 44; do echo "cfg $c"; ARCO_CONV3D_FL_CFG=$c timeout 300 python tools/micro/fl_bench.py 4 2>&1 | tail -6; done
