#!/bin/bash
# on the GPU box: the round's final evidence (last commit) -> gpurun_out/r06o_*
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp; mkdir -p $o
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $o/r06o_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06o_tests.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06o_bench.json 2> $o/r06o_bench.err
bash tools/prof_run.sh r06o 40 > $o/r06o_prof2d.txt 2>&1
bash tools/prof_run3d.sh r06o > $o/r06o_prof3d.txt 2>&1
bash tools/prof_run3d.sh r06o_lits "SHAPE=1 160 160 96" ACT_DTYPE=f16 > $o/r06o_prof3d_lits.txt 2>&1
bash tools/prof_bench.sh r06o > $o/r06o_prof_bench.txt 2>&1
cat $o/r06o_tests.log; head -3 $o/r06o_prof2d.txt | cut -c1-160; head -3 $o/r06o_prof3d.txt | cut -c1-160; head -3 $o/r06o_prof3d_lits.txt | cut -c1-160
