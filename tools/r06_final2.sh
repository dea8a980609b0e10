#!/bin/bash
# on the GPU box: the round's final evidence (last commit) -> gpurun_out/r06f_*
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp; mkdir -p $o
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $o/r06f_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06f_tests.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06f_bench.json 2> $o/r06f_bench.err
bash tools/prof_run3d.sh r06f > $o/r06f_prof3d.txt 2>&1
cat $o/r06f_tests.log; tail -c 300 $o/r06f_bench.json; head -3 $o/r06f_prof3d.txt | cut -c1-160; head -3 $o/r06f_prof3d_lits.txt | cut -c1-160; head -3 $o/r06f_prof2d.txt | cut -c1-160; tail -3 $o/r06f_prof_bench.txt | cut -c1-200
