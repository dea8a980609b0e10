"""Builds build/libsptrace.so: conv_sp.hip with s_memtime stamps of one workgroup (consumer wave 0 and producer wave 4:
per step before the wait / after the barrier / at the end of the step's work; tile ends), printed by the host on the 30th
launch of an instantiation.  ARCO_LIB=build/libsptrace.so python tools/micro/conv_sp_check.py 2"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "arco_amd/csrc/conv_sp.hip")).read()
MARK = "// ---------------------------------------------------------------------------------------------------------------------------\n// Resident-weights variant"
part = 0          # 0: everything up to the resident-weights kernel, 1: from there on


def rep(old, new, cnt=1):
    global s
    i = s.index(MARK)
    a, b = s[:i], s[i:]
    t = a if part == 0 else b
    assert t.count(old) == cnt, (old, t.count(old))
    t = t.replace(old, new)
    s = t + b if part == 0 else a + t
rep('#include "igemm_args.h"', '#include "../arco_amd/csrc/igemm_args.h"')
rep('template <int N> __device__ __forceinline__ void wait_vm()', '''__device__ unsigned long long sp_trace[2 * 2048];
#define STAMP() do { if (blockIdx.x == TRACE_BLOCK && lane == 0 && wid == 0 && ti < 2047) sp_trace[(producer ? 2048 : 0) + ti] = __builtin_amdgcn_s_memtime(); ++ti; } while (0)
template <int N> __device__ __forceinline__ void wait_vm()''')
rep('  const bool producer = threadIdx.x >= 256;', '  const bool producer = threadIdx.x >= 256;\n  int ti = 0;\n  STAMP();')
# producer step
rep('        wait_vm<NS>();\n        wait_lgkm0();\n        __builtin_amdgcn_s_barrier();\n        refill(', '        STAMP();\n        wait_vm<NS>();\n        wait_lgkm0();\n        __builtin_amdgcn_s_barrier();\n        STAMP();\n        refill(')
rep('        if (S == 4) load_A(d3, gc + 3 < total_gc, SET);\n      };', '        if (S == 4) load_A(d3, gc + 3 < total_gc, SET);\n        STAMP();\n      };')
rep('    wait_vm<0>();\n    return;', '    if (blockIdx.x == TRACE_BLOCK && lane == 0 && wid == 0) sp_trace[2048 + 2047] = ti;\n    wait_vm<0>();\n    return;')
# consumer step
rep('      wait_lgkm0();\n      __builtin_amdgcn_s_barrier();\n      // LDS reads in the shadow', '      STAMP();\n      wait_lgkm0();\n      __builtin_amdgcn_s_barrier();\n      STAMP();\n      // LDS reads in the shadow')
rep('  auto tile_end = [&]() {', '  auto tile_end = [&]() {\n    STAMP();')
rep("            a.stat_sq[(n0 + ct * 16 + 4 * g + r) * nslab + slab] = v2;\n          }\n        }\n    }\n  };", "            a.stat_sq[(n0 + ct * 16 + 4 * g + r) * nslab + slab] = v2;\n          }\n        }\n    }\n    STAMP();\n  };")
rep('      advance(d0);\n    }\n  }\n}', '      advance(d0);\n    }\n  }\n  if (blockIdx.x == TRACE_BLOCK && lane == 0 && wid == 0) sp_trace[2047] = ti;\n}')
part = 1          # (launch_sp sits behind the resident-weights kernel)
rep('''  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), G::LDS_BYTES, st, b);
  return arco_launch_status();''', '''  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), G::LDS_BYTES, st, b);
  static int nl = 0;
  if (++nl == 30) {
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(sp_trace), sizeof(h));
    for (int w = 0; w < 2; ++w) {
      const int n = (int)h[w * 2048 + 2047];
      fprintf(stderr, "TRACE A_T=%d C_T=%d %s (%d stamps):", A_T, C_T, w ? "producer" : "consumer", n);
      for (int i = 1; i < n && i < 2047; ++i) fprintf(stderr, " %llu", h[w * 2048 + i] - h[w * 2048 + i - 1]);
      fprintf(stderr, "\\n");
    }
  }
  return arco_launch_status();''')
# ---- resident-weights kernel: consumer stamps around the chunk rendezvous and tile ends, producer stamps per chunk
part = 1
rep('  const bool producer = threadIdx.x >= 256;\n  const int tiles_x = a.W >> 4, tiles_y = a.H / TH, tiles_img = tiles_x * tiles_y;\n  const int total_tiles = a.n_mblocks;',
    '  const bool producer = threadIdx.x >= 256;\n  int ti = 0;\n  STAMP();\n  const int tiles_x = a.W >> 4, tiles_y = a.H / TH, tiles_img = tiles_x * tiles_y;\n  const int total_tiles = a.n_mblocks;')
rep('      if (S == 4) {                    // the one rendezvous of the chunk: the next chunk\'s activations are complete\n        wait_lgkm0();\n        __builtin_amdgcn_s_barrier();\n      }',
    '      if (S == 4) {\n        STAMP();\n        wait_lgkm0();\n        __builtin_amdgcn_s_barrier();\n        STAMP();\n      }')
rep('      wait_vm<NA>();                   // chunk k + 2 has landed (only the loads of chunk k + 3 are younger)\n      store_all(As, 0);\n      load_A(dn, k + 4 < total_gc, 0);',
    '      STAMP();\n      wait_vm<NA>();\n      STAMP();\n      store_all(As, 0);\n      STAMP();\n      load_A(dn, k + 4 < total_gc, 0);\n      STAMP();')
rep('    wait_vm<0>();\n    return;\n  }\n\n  // ================================================================== consumer waves\n  const int tl = g >> 1;\n  int aoff[5];\n#pragma unroll\n  for (int s_ = 0;',
    '    if (blockIdx.x == TRACE_BLOCK && lane == 0 && wid == 0) sp_trace[2048 + 2047] = ti;\n    wait_vm<0>();\n    return;\n  }\n\n  const int tl = g >> 1;\n  int aoff[5];\n#pragma unroll\n  for (int s_ = 0;')
rep('  auto tile_end = [&]() {\n    asm volatile("s_nop 15\\n\\ts_nop 15" ::: "memory");\n    float s1[C_T][4], s2[C_T][4];',
    '  auto tile_end = [&]() {\n    STAMP();\n    asm volatile("s_nop 15\\n\\ts_nop 15" ::: "memory");\n    float s1[C_T][4], s2[C_T][4];')
rep('      chunk(std::integral_constant<int, 1>{}, gc + 1, d0.c, gc + 2 < total_gc);\n      if (d0.c + 1 == nchunks) tile_end();\n      advance(d0);\n    }\n  }\n}',
    '      chunk(std::integral_constant<int, 1>{}, gc + 1, d0.c, gc + 2 < total_gc);\n      if (d0.c + 1 == nchunks) tile_end();\n      advance(d0);\n    }\n  }\n  if (blockIdx.x == TRACE_BLOCK && lane == 0 && wid == 0) sp_trace[2047] = ti;\n}')
rep("""  hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks < cus ? mblocks : cus)), dim3(512), lds, st, b);
  return arco_launch_status();""", """  hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks < cus ? mblocks : cus)), dim3(512), lds, st, b);
  static int nl = 0;
  if (++nl == 30) {
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(sp_trace), sizeof(h));
    for (int w = 0; w < 2; ++w) {
      const int n = (int)h[w * 2048 + 2047];
      fprintf(stderr, "TRACE rw A_T=%d C_T=%d %s (%d stamps):", A_T, C_T, w ? "producer" : "consumer", n);
      for (int i = 1; i < n && i < 2047; ++i) fprintf(stderr, " %llu", h[w * 2048 + i] - h[w * 2048 + i - 1]);
      fprintf(stderr, "\\n");
    }
  }
  return arco_launch_status();""")
os.makedirs(os.path.join(root, "build"), exist_ok=True)
open(os.path.join(root, "build/conv_sp_trace.hip"), "w").write(s)
c = os.path.join(root, "arco_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-uninitialized", "-DTRACE_BLOCK=7", "-c",
                       os.path.join(root, "build/conv_sp_trace.hip"), "-o", os.path.join(root, "build/conv_sp_trace.o")])
objs = [os.path.join(c, f) for f in ("igemm.o", "loss_front.o", "elementwise.o", "glue.o", "sampler_host.o", "augment.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", os.path.join(root, "build/conv_sp_trace.o")] + objs + ["-lpthread", "-o", os.path.join(root, "build/libsptrace.so")])
print("built build/libsptrace.so")
