#!/bin/bash
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $o/r06_c4_gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench_a.json 2> $o/r06_bench_a.err
tail -c 1500 $o/r06_bench_a.json; cat $o/r06_c4_gputests.log
