#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_run.sh <tag> [N] [extra trainer flags]
# kernel trace of tools/prof_step.py -> gpurun_out/prof_<tag>/ + gpurun_out/prof_<tag>_kernel_stats.csv + top rows on stdout
tag=$1; n=${2:-40}; shift; shift
root=$(pwd)
export TMPDIR=/tmp
mkdir -p $root/gpurun_out
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python3 $root/tools/prof_step.py $n "$@" > $root/gpurun_out/prof_${tag}.log 2>&1)
python3 $root/tools/prof_summary.py /tmp/prof_$tag $root/gpurun_out/prof_${tag}_kernel_stats.csv $((n + 6)) > $root/gpurun_out/prof_${tag}_top.txt 2>&1
tail -2 $root/gpurun_out/prof_${tag}.log; head -60 $root/gpurun_out/prof_${tag}_top.txt
