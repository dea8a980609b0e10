#!/bin/bash
# A/B builds of the C-ABI library: bash tools/build_variant.sh <name> "<-D flags>" [sources to recompile with the flags, default conv_sp.hip]
# -> arco_amd/lib/libarco_hip_<name>.so (the other objects are taken from the regular build); select it with ARCO_LIB=<path>
set -e
name=$1; flags=$2; shift; shift
srcs=${@:-conv_sp.hip}
cd "$(dirname "$0")/../arco_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../build/var_$name
objs=""
for s in loss_front igemm conv_sp gemm_sp conv_h elementwise det_scatter glue sampler_host augment; do
  if [[ " $srcs " == *" $s.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value $flags -c $s.hip -o ../../build/var_$name/$s.o
    objs="$objs ../../build/var_$name/$s.o"
  else
    objs="$objs $s.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -lpthread -o ../lib/libarco_hip_$name.so
echo built arco_amd/lib/libarco_hip_$name.so
