#!/bin/bash
out=gpurun_out/ab_wgstride.txt; : > $out
for rep in 1 2; do
for v in base z72 x120 z72x120 x136 z72x136; do
  if [ "$v" = base ]; then lib=""; else lib=$(pwd)/arco_amd/lib/libarco_hip_$v.so; fi
  echo "=== $v (rep $rep)" >> $out
  ARCO_LIB=$lib python3 tools/micro/wgrad_abl.py 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
