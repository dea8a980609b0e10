#!/bin/bash
# on the GPU box: kernel trace of tests/dropin_user.py time N -> gpurun_out/prof_dropin_user_kernel_stats.csv + top rows
root=$(pwd); export TMPDIR=/tmp; n=${1:-6}
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_du -- python3 $root/tests/dropin_user.py time $n > $root/gpurun_out/prof_dropin_user.log 2>&1)
python3 $root/tools/prof_summary.py /tmp/prof_du $root/gpurun_out/prof_dropin_user_kernel_stats.csv $((n + 3)) > $root/gpurun_out/prof_dropin_user_top.txt 2>&1
grep DROPIN_USER_TIME $root/gpurun_out/prof_dropin_user.log | cut -c1-120; head -40 $root/gpurun_out/prof_dropin_user_top.txt
