#!/bin/bash
# on the GPU box: bash tools/ab_micro.sh <out tag> base var1 var2 ...  -> the two micro-benchmarks per library variant, twice, interleaved
tag=$1; shift
out=gpurun_out/ab_$tag.txt; : > $out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$(pwd)/arco_amd/lib/libarco_hip_$v.so; fi
  echo "=== $v (rep $rep)" >> $out
  ARCO_LIB=$lib python3 tools/micro/hbm_ref.py 2>&1 | grep -v amdgpu.ids >> $out
  ARCO_LIB=$lib python3 tools/micro/conv_sp_check.py 2>&1 | grep -v amdgpu.ids | sed -e 's/cfg .* err64/err64/' >> $out
done
done
cat $out
