#!/bin/bash
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_block_fuse_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -3 > $o/r06_c3_tests.log
python -m pytest tests/test_step_parity_gpu.py -x -q -k "cfg1" 2>&1 | tail -8 >> $o/r06_c3_tests.log
python -m pytest tests/test_step3d_parity_gpu.py -x -q -k "free_running" -s 2>&1 | tail -8 >> $o/r06_c3_tests.log
python -m pytest tests/test_dropin_user_gpu.py -x -q -k "literal" 2>&1 | tail -8 >> $o/r06_c3_tests.log
tools/ab_bench.sh fuse3 "ARCO_BLOCK_FUSE=0" "ARCO_BLOCK_FUSE=1" > $o/r06_ab3.log 2>&1
ARCO_BLOCK_FUSE=1 bash tools/prof_run.sh r06c3_fuse1 40 > $o/r06c3_prof_fuse1.txt 2>&1
python -m pytest tests/test_dice_parity_gpu.py -x -q -s 2>&1 | tail -8 >> $o/r06_c3_tests.log
cat $o/r06_c3_tests.log $o/r06_ab3.log
