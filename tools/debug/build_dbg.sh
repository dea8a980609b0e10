#!/bin/bash
# builds the instrumented probe kernels of the two-queue hazard and assembles the hand-edited ISA variants (not part of the product library)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 lerp4_dbg.hip -o liblerp4dbg.so || exit 1
LL=/opt/rocm/lib/llvm/bin
for s in isa/*.s; do
  $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $s -o /tmp/isa_tmp.o && $LL/ld.lld -shared /tmp/isa_tmp.o -o ${s%.s}.co || exit 1
done
ls -la isa/*.co
