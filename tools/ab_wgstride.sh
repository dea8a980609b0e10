#!/bin/bash
# on the GPU box: bash tools/ab_wgstride.sh base v1 v2 ...  -> tools/micro/wgrad_abl.py per library variant, twice, interleaved
out=gpurun_out/ab_wgstride.txt; : > $out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$(pwd)/arco_amd/lib/libarco_hip_$v.so; fi
  echo "=== $v (rep $rep)" >> $out
  ARCO_LIB=$lib python3 tools/micro/wgrad_abl.py 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
