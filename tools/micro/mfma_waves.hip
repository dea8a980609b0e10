// v_mfma_f32_16x16x32_bf16 issue rate per SIMD with 1, 2 and 4 MFMA waves per SIMD (same MFMA count per SIMD), and with 1 / 2 / 4
// independent accumulators per wave.   hipcc -O3 --offload-arch=gfx950 mfma_waves.hip -o mfma_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(int n, float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j % NACC], 0, 0, 0);
  out[blockIdx.x * 1024 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
template <int NACC>
void run(float* out, hipEvent_t e0, hipEvent_t e1) {
  for (int waves : {4, 8, 16}) {                 // per CU: 1, 2, 4 per SIMD
    const int n = 80000 / (waves / 4) / 4;       // 80 000 MFMAs per SIMD in all
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(waves * 64), 0, 0, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%d accumulator(s) per wave, %d wave(s) per SIMD: %.3f ms -> %.1f cycles per MFMA per SIMD at 2.4 GHz\n", NACC, waves / 4, ms, ms * 1e-3 * 2.4e9 / 80000.0);
  }
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<1>(out, e0, e1); run<2>(out, e0, e1); run<4>(out, e0, e1);
  return 0;
}
