#!/bin/bash
# on the GPU box: the round's evidence in one call -> gpurun_out/r04h_*  (copied into profiles/ afterwards)
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r04h_bench.json 2> $o/r04h_bench.err
bash tools/prof_run.sh r04h 40 > $o/r04h_prof2d.txt 2>&1
bash tools/prof_run3d.sh r04h > $o/r04h_prof3d.txt 2>&1
bash tools/prof_run3d.sh r04h_lits "SHAPE=1 160 160 96" ACT_DTYPE=f16 > $o/r04h_prof3d_lits.txt 2>&1
mkdir -p $o/r04_pmc_conv_rw
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc4_hf -o f -- python3 $root/tools/pmc_conv_rw.py > $o/r04h_pmc_f.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc4_hw -o w -- python3 $root/tools/pmc_conv_rw.py > $o/r04h_pmc_w.log 2>&1)
grep -h "conv3x3_rw_kernel" $(find /tmp/pmc4_hf -name "*counter_collection.csv" | head -1) > $o/r04_pmc_conv_rw/fetch_size_counter_collection.csv
grep -h "conv3x3_rw_kernel" $(find /tmp/pmc4_hw -name "*counter_collection.csv" | head -1) > $o/r04_pmc_conv_rw/write_size_counter_collection.csv
head -1 $(find /tmp/pmc4_hf -name "*counter_collection.csv" | head -1) > $o/r04_pmc_conv_rw/header.csv
tail -c 800 $o/r04h_bench.json; head -3 $o/r04h_prof2d.txt | cut -c1-160; wc -l $o/r04_pmc_conv_rw/*.csv
