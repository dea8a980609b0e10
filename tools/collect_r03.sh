#!/bin/bash
# on the GPU box: the round's evidence in one call -> gpurun_out/r03_*
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
bash tools/prof_run.sh r03c 40 > $o/r03_prof2d.txt 2>&1
bash tools/prof_run3d.sh r03c > $o/r03_prof3d.txt 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python3 $root/tools/pmc_conv_rw.py > $o/r03_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python3 $root/tools/pmc_conv_rw.py > $o/r03_pmc_w.log 2>&1
mkdir -p $o/r03_pmc_conv_rw
cp $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $o/r03_pmc_conv_rw/fetch_size_counter_collection.csv
cp $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $o/r03_pmc_conv_rw/write_size_counter_collection.csv
cd $root
python3 tools/micro/hbm_ref.py > $o/r03_hbm_ref.txt 2>&1
python3 tools/unet_layer_bench.py 16 > $o/r03_unet_layers.txt 2>&1
python3 tools/kernel_bench3d.py 4 > $o/r03_layers3d.txt 2>&1
tail -3 $o/r03_hbm_ref.txt; tail -1 $o/r03_unet_layers.txt; head -3 $o/r03_prof2d.txt | cut -c1-160; head -3 $o/r03_prof3d.txt | cut -c1-160
wc -l $o/r03_pmc_conv_rw/*.csv
