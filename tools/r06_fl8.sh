#!/bin/bash
# on the GPU box: state after the 3x3x3 kernel family - whole GPU suite, bench line, LA profile
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp; mkdir -p $o
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $o/r06c_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06c_tests.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06c_bench.json 2> $o/r06c_bench.err
bash tools/prof_run3d.sh r06c > $o/r06c_prof3d.txt 2>&1
cat $o/r06c_tests.log; tail -c 600 $o/r06c_bench.json; head -12 $o/r06c_prof3d.txt | cut -c1-200
