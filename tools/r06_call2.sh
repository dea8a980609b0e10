#!/bin/bash
# round 6, GPU call 2: fused-block profile A/B, timeline A/B, the new parity tests, the Dice-parity run
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_block_fuse_gpu.py -x -q 2>&1 | tail -3 > $o/r06_c2_tests.log
python -m pytest tests/test_step_parity_gpu.py -x -q -k "cfg1" 2>&1 | tail -15 >> $o/r06_c2_tests.log
ARCO_BLOCK_FUSE=1 bash tools/prof_run.sh r06_fuse1 40 > $o/r06_prof_fuse1.txt 2>&1
ARCO_BLOCK_FUSE=0 bash tools/prof_run.sh r06_fuse0 40 > $o/r06_prof_fuse0.txt 2>&1
ARCO_BLOCK_FUSE=1 python3 tools/step_timeline.py > $o/r06_timeline_fuse1.txt 2>&1
ARCO_BLOCK_FUSE=0 python3 tools/step_timeline.py > $o/r06_timeline_fuse0.txt 2>&1
python3 tools/dice_parity.py --steps 200 --out $o/r06_dice_parity > $o/r06_dice_parity.log 2>&1
cat $o/r06_c2_tests.log; tail -3 $o/r06_dice_parity.log
