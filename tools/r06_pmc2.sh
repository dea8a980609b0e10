#!/bin/bash
# PMC traffic passes (separate FETCH_SIZE / WRITE_SIZE runs, counters only with --kernel-trace) for the round's last instantiations:
# conv3x3_rw_kernel<4,1,.,NCW=8> (tools/pmc_conv_rw.py) and the 3x3x3 kernels conv3d_fc_kernel / hconv_fc_kernel (tools/pmc_conv3d_fc.py)
root=$(pwd); o=$root/gpurun_out; mkdir -p $o/r06_pmc2
export TMPDIR=/tmp
run() { # name counter script [env]
  (cd /tmp && rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc6_$1_$2 -o c -- python3 $root/$3 > $o/r06_pmc2/$1_$2.log 2>&1)
  f=$(find /tmp/pmc6_$1_$2 -name "*counter_collection.csv" | head -1)
  head -1 $f > $o/r06_pmc2/header.csv
  grep -h "conv3x3_rw_kernel\|conv3d_fc_kernel\|conv3d_fl_kernel\|hconv_fc_kernel\|hconv_kernel\|igemm_kernel" $f > $o/r06_pmc2/$1_$2.csv
}
run rw FETCH_SIZE tools/pmc_conv_rw.py
run rw WRITE_SIZE tools/pmc_conv_rw.py
run fc FETCH_SIZE tools/pmc_conv3d_fc.py
run fc WRITE_SIZE tools/pmc_conv3d_fc.py
export HALF=1
run hfc FETCH_SIZE tools/pmc_conv3d_fc.py
run hfc WRITE_SIZE tools/pmc_conv3d_fc.py
wc -l $o/r06_pmc2/*.csv; tail -3 $o/r06_pmc2/fc_FETCH_SIZE.log
