// Do VALU work of one wave and MFMA work of another wave on the SAME SIMD overlap on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a chain-free MFMA loop, waves 4-7 a VALU loop.
// mode 1: MFMA waves only; 2: VALU waves only; 3: both.   hipcc -O3 --offload-arch=gfx950 valu_mfma_overlap.hip -o valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, float* out) {
  const int wid = threadIdx.x >> 6;
  if (wid < 4) {
    if (!(mode & 1)) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n_mfma; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    if (!(mode & 2)) return;
    if (mode & 4) {        // integer / bit ops instead of fp32 FMAs
      unsigned y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
      for (int i = 0; i < n_valu; ++i) {
        y0 = (y0 << 3) ^ (y0 >> 5); y1 = (y1 << 3) ^ (y1 >> 5); y2 = (y2 << 3) ^ (y2 >> 5); y3 = (y3 << 3) ^ (y3 >> 5);
        y4 = (y4 << 3) ^ (y4 >> 5); y5 = (y5 << 3) ^ (y5 >> 5); y6 = (y6 << 3) ^ (y6 >> 5); y7 = (y7 << 3) ^ (y7 >> 5);
      }
      out[blockIdx.x * 512 + threadIdx.x] = (float)(y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7);
      return;
    }
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < n_valu; ++i) {
      x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f; x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f;
      x4 = x4 * 1.0001f + 0.5f; x5 = x5 * 1.0001f + 0.5f; x6 = x6 * 1.0001f + 0.5f; x7 = x7 * 1.0001f + 0.5f;
    }
    out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_mfma = 20000, n_valu = 40000;       // 80 000 MFMAs / 320 000 VALU ops per wave
  for (int mode : {1, 2, 3, 6, 7}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, out); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("mode %d (%s): %.3f ms%s\n", mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU (fp32 FMA) waves only" : mode == 3 ? "MFMA + fp32-FMA waves" : mode == 6 ? "VALU (shift/xor) waves only" : "MFMA + shift/xor waves", ms,
                           mode == 1 ? "  -> cycles per MFMA at 2.4 GHz: " : "");
      if (rep == 2 && mode == 1) printf("      %.1f\n", ms * 1e-3 * 2.4e9 / (4.0 * n_mfma));
      if (rep == 2 && mode == 2) printf("      cycles per VALU op at 2.4 GHz: %.2f\n", ms * 1e-3 * 2.4e9 / (8.0 * n_valu));
    }
  }
  return 0;
}
