#!/bin/bash
# on the GPU box: bash tools/prof_run3d.sh <tag> [env assignments...]  -> kernel trace of tools/bench3d.py 2 (LA shape, trainer-default flags)
tag=$1; shift
root=$(pwd); export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export EQV_PASS=${EQV_PASS:-1} CONV_MMA=${CONV_MMA:-f32x3} GRAPH_TRAIN=${GRAPH_TRAIN:-1}
# SHAPE="1 160 160 96" ACT_DTYPE=f16: the configs[4] shard with f16 activation storage
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof3d_$tag -- python3 $root/tools/bench3d.py ${SHAPE:-2} > $root/gpurun_out/prof3d_${tag}.log 2>&1)
python3 $root/tools/prof_summary.py /tmp/prof3d_$tag $root/gpurun_out/prof3d_${tag}_kernel_stats.csv 12 > $root/gpurun_out/prof3d_${tag}_top.txt 2>&1
grep "3D step" $root/gpurun_out/prof3d_${tag}.log; head -40 $root/gpurun_out/prof3d_${tag}_top.txt
