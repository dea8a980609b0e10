#!/bin/bash
# timing-only ablation builds of conv3x3_rw_kernel (results are wrong by construction): which part of the launch does it wait for?
cd "$(dirname "$0")/../.."
for v in NO_MFMA NO_SPLIT NO_LOAD NO_STORE "NO_MFMA -DRW_NO_SPLIT" "NO_LOAD -DRW_NO_SPLIT" "NO_MFMA -DRW_NO_SPLIT -DRW_NO_STORE"; do
  name=rw_$(echo "$v" | tr -d ' -' | tr 'A-Z' 'a-z' | sed 's/drw_//g')
  bash tools/build_variant.sh $name "-DRW_$v" conv_sp.hip > /dev/null 2>&1 && echo "built $name"
done
