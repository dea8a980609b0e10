#!/bin/bash
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $o/r06_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06_final_tests.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_last.json 2> $o/r06_bench_last.err
bash tools/prof_run.sh r06_last 40 > $o/r06_prof2d_last.txt 2>&1
bash tools/prof_bench.sh r06_last > $o/r06_prof_bench_last.txt 2>&1
cat $o/r06_final_tests.log; tail -c 400 $o/r06_bench_last.json
