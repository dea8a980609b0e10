"""Builds build/libsptrace.so: conv_sp.hip with s_memtime stamps (per step: before the wait, after the barrier, after the
MFMA issue; per chunk end; per tile end) of one workgroup, printed by the host on the 30th launch of an instantiation."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "arco_amd/csrc/conv_sp.hip")).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (old, s.count(old))
    s = s.replace(old, new)
rep('#include "igemm_args.h"', '#include "../arco_amd/csrc/igemm_args.h"')
rep('template <int N> __device__ __forceinline__ void wait_vm()', '''__device__ unsigned long long sp_trace[4096];
#define STAMP() do { if (blockIdx.x == TRACE_BLOCK && lane == 0 && ti < 1023) sp_trace[wid * 1024 + ti] = __builtin_amdgcn_s_memtime(); ++ti; } while (0)
template <int N> __device__ __forceinline__ void wait_vm()''')
rep('  // ---- prologue.', '  int ti = 0;\n  STAMP();\n  // ---- prologue.')
rep('  for (int gc = 0; gc < total_gc; ++gc) {\n    const bool more', '  STAMP();\n  for (int gc = 0; gc < total_gc; ++gc) {\n    const bool more')
rep('      if (tail) wait_vm<0>();\n      else if (!post', '      STAMP();\n      if (tail) wait_vm<0>();\n      else if (!post')
rep('      __builtin_amdgcn_s_barrier();\n      if (S == 0) refill(4, d0);', '      __builtin_amdgcn_s_barrier();\n      STAMP();\n      if (S == 0) refill(4, d0);')
rep('      if (S == 3 && !tail) load_A(d2);\n', '      STAMP();\n      if (S == 3 && !tail) load_A(d2);\n')
rep('    if (d0.c + 1 == nchunks) {         // ---- tile done', '    STAMP();\n    if (d0.c + 1 == nchunks) {         // ---- tile done')
rep('    d0 = d1; d1 = d2; advance(d2);\n  }\n}', '    STAMP();\n    d0 = d1; d1 = d2; advance(d2);\n  }\n  if (blockIdx.x == TRACE_BLOCK && lane == 0) sp_trace[wid * 1024 + 1023] = ti;\n}')
rep('''  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(256), G::LDS_BYTES, st, b);
  return arco_launch_status();''', '''  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(256), G::LDS_BYTES, st, b);
  static int nl = 0;
  if (++nl == 30) {
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(sp_trace), sizeof(h));
    for (int w = 0; w < 4; w += 3) {
      const int n = (int)h[w * 1024 + 1023];
      fprintf(stderr, "TRACE A_T=%d C_T=%d wave %d (%d stamps):", A_T, C_T, w, n);
      for (int i = 1; i < n && i < 1023; ++i) fprintf(stderr, " %llu", h[w * 1024 + i] - h[w * 1024 + i - 1]);
      fprintf(stderr, "\\n");
    }
  }
  return arco_launch_status();''')
os.makedirs(os.path.join(root, "build"), exist_ok=True)
open(os.path.join(root, "build/conv_sp_trace.hip"), "w").write(s)
c = os.path.join(root, "arco_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-DTRACE_BLOCK=7", "-c",
                       os.path.join(root, "build/conv_sp_trace.hip"), "-o", os.path.join(root, "build/conv_sp_trace.o")])
objs = [os.path.join(c, f) for f in ("igemm.o", "loss_front.o", "elementwise.o", "glue.o", "sampler_host.o", "augment.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", os.path.join(root, "build/conv_sp_trace.o")] + objs + ["-lpthread", "-o", os.path.join(root, "build/libsptrace.so")])
print("built build/libsptrace.so")
