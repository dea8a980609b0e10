#!/bin/bash
# on the GPU box: counters of the f16-storage kernels -> gpurun_out/r03h_*
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out; mkdir -p $o/r03_pmc_hconv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmch_$c -o x -- python3 $root/tools/pmc_hconv.py > $o/r03h_$c.log 2>&1
  cp $(find /tmp/pmch_$c -name "*counter_collection.csv" | head -1) $o/r03_pmc_hconv/$c.csv
done
export ARCO_LIB=$root/arco_amd/lib/libarco_hip_rows4880.so
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcv_$c -o x -- python3 $root/tools/pmc_hconv.py > $o/r03h_v_$c.log 2>&1
  cp $(find /tmp/pmcv_$c -name "*counter_collection.csv" | head -1) $o/r03_pmc_hconv/rows4880_$c.csv
done
unset ARCO_LIB
export ARCO_HCONV_RW=0
for c in FETCH_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$c -o x -- python3 $root/tools/pmc_hconv.py > $o/r03h_g_$c.log 2>&1
  cp $(find /tmp/pmcg_$c -name "*counter_collection.csv" | head -1) $o/r03_pmc_hconv/norw_$c.csv
done
cd $root; wc -l $o/r03_pmc_hconv/*.csv
