#!/bin/bash
# on the GPU box: kernel trace of the default step -> idle / overlap report (tools/gap_report.py)
root=$(pwd); export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/gap_$1 -- python3 $root/tools/prof_step.py 30 > $root/gpurun_out/gap_$1.log 2>&1)
tail -1 $root/gpurun_out/gap_$1.log
python3 tools/gap_report.py /tmp/gap_$1 8 > $root/gpurun_out/gap_$1.txt 2>&1; cat $root/gpurun_out/gap_$1.txt
