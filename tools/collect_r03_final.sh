#!/bin/bash
# on the GPU box: the round's final evidence in one call -> gpurun_out/r03f_*
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r03f_bench.json 2> $o/r03f_bench.err
bash tools/prof_run.sh r03f 40 > $o/r03f_prof2d.txt 2>&1
bash tools/prof_run3d.sh r03f > $o/r03f_prof3d.txt 2>&1
bash tools/prof_run3d.sh r03f_lits "SHAPE=1 160 160 96" ACT_DTYPE=f16 > $o/r03f_prof3d_lits.txt 2>&1
python3 tools/half_layer_bench.py lits > $o/r03f_half_layers_lits.txt 2>&1
bash tools/collect_r03h.sh > $o/r03f_pmc_hconv.txt 2>&1
tail -c 1500 $o/r03f_bench.json; head -3 $o/r03f_prof2d.txt | cut -c1-160; head -3 $o/r03f_prof3d.txt | cut -c1-160; head -3 $o/r03f_prof3d_lits.txt | cut -c1-160
