// MFMA issue-rate micro-benchmark on gfx950: shader cycles per instruction for f32 16x16x4, bf16 16x16x16 (legacy K=16)
// and bf16 16x16x32, one wave per SIMD, 8 independent accumulators, hand-allocated AGPRs.
// plus the 16x16x32 rate with 1 / 2 / 4 dependent accumulation chains (back-to-back MFMAs on the same accumulator).
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"
#define EIGHT(INS) INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" INS " a[8:11], %0, %1, a[8:11]\n" INS " a[12:15], %0, %1, a[12:15]\n" \
                   INS " a[16:19], %0, %1, a[16:19]\n" INS " a[20:23], %0, %1, a[20:23]\n" INS " a[24:27], %0, %1, a[24:27]\n" INS " a[28:31], %0, %1, a[28:31]\n"
#define DEP1(INS) INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n" \
                  INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[0:3], %0, %1, a[0:3]\n"
#define DEP2(INS) INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" \
                  INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n"
#define DEP4(INS) INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" INS " a[8:11], %0, %1, a[8:11]\n" INS " a[12:15], %0, %1, a[12:15]\n" \
                  INS " a[0:3], %0, %1, a[0:3]\n" INS " a[4:7], %0, %1, a[4:7]\n" INS " a[8:11], %0, %1, a[8:11]\n" INS " a[12:15], %0, %1, a[12:15]\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  const float x = threadIdx.x * 1e-3f, y = x + 1.f;
  s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(x + i); b8[i] = (__bf16)(x - i); }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile(EIGHT("v_mfma_f32_16x16x4_f32") :: "v"(x), "v"(y) : CLOB);
    if (MODE == 1) asm volatile(EIGHT("v_mfma_f32_16x16x16_bf16") :: "v"(a4), "v"(b4) : CLOB);
    if (MODE == 2) asm volatile(EIGHT("v_mfma_f32_16x16x32_bf16") :: "v"(a8), "v"(b8) : CLOB);
    if (MODE == 3) asm volatile(DEP1("v_mfma_f32_16x16x32_bf16") :: "v"(a8), "v"(b8) : CLOB);
    if (MODE == 4) asm volatile(DEP2("v_mfma_f32_16x16x32_bf16") :: "v"(a8), "v"(b8) : CLOB);
    if (MODE == 5) asm volatile(DEP4("v_mfma_f32_16x16x32_bf16") :: "v"(a8), "v"(b8) : CLOB);
  }
  long long t1 = clock64();
  float s;
  asm volatile("s_nop 15\n s_nop 15\n v_accvgpr_read_b32 %0, a0" : "=v"(s) :: CLOB);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 6 * 8);
  const int iters = 10000;
  const char* names[6] = {"f32 16x16x4", "bf16 16x16x16", "bf16 16x16x32", "x32 dependent chain", "x32 2 chains", "x32 4 chains"};
  const double flop[6] = {2.0 * 16 * 16 * 4, 2.0 * 16 * 16 * 16, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32};
  for (int m = 0; m < 6; ++m) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      if (m == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c[6]; (void)hipMemcpy(c, cyc, 48, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8;
    printf("%-16s %.1f clock64-ticks/MFMA  %.3f ms  %.1f TFLOP/s chip (1 wave/SIMD)\n", names[m], c[m] / n, ms, flop[m] * n * 4 * 256 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
