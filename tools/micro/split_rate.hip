// VALU cost of the fp32 -> 3 x bf16 split in isolation: the element-wise C++ form (what round 2-4 shipped) against the packed-pair
// form (v_cvt_pk_bf16_f32 per pair + bit expansion + v_pk_add_f32), waves of pure split work.  hipcc -O3 --offload-arch=gfx950 split_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_old(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  unsigned short h[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 b0 = (__bf16)v[e]; const float r1 = v[e] - (float)b0;
    const __bf16 b1 = (__bf16)r1; const float r2 = r1 - (float)b1;
    const __bf16 b2 = (__bf16)r2;
    h[0][e] = __builtin_bit_cast(unsigned short, b0); h[1][e] = __builtin_bit_cast(unsigned short, b1); h[2][e] = __builtin_bit_cast(unsigned short, b2);
  }
  p0 = u32x2{(unsigned)h[0][0] | ((unsigned)h[0][1] << 16), (unsigned)h[0][2] | ((unsigned)h[0][3] << 16)};
  p1 = u32x2{(unsigned)h[1][0] | ((unsigned)h[1][1] << 16), (unsigned)h[1][2] | ((unsigned)h[1][3] << 16)};
  p2 = u32x2{(unsigned)h[2][0] | ((unsigned)h[2][1] << 16), (unsigned)h[2][2] | ((unsigned)h[2][3] << 16)};
}
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ void pair(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = cvt_pk(x0, x1);
  const f32x2_t x = {x0, x1};
  const f32x2_t f0 = {__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h0 & 0xffff0000u)};
  const f32x2_t r1 = x - f0;
  h1 = cvt_pk(r1[0], r1[1]);
  const f32x2_t f1 = {__builtin_bit_cast(float, h1 << 16), __builtin_bit_cast(float, h1 & 0xffff0000u)};
  const f32x2_t r2 = r1 - f1;
  h2 = cvt_pk(r2[0], r2[1]);
}
__device__ __forceinline__ void split_new(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  unsigned a0, a1, a2, b0, b1, b2;
  pair(v[0], v[1], a0, a1, a2); pair(v[2], v[3], b0, b1, b2);
  p0 = u32x2{a0, b0}; p1 = u32x2{a1, b1}; p2 = u32x2{a2, b2};
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const f32x4* in, unsigned* out, int iters) {
  f32x4 v[4];
  for (int i = 0; i < 4; ++i) v[i] = in[threadIdx.x * 4 + i];
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 p0, p1, p2;
      if (MODE == 0) split_old(v[i], p0, p1, p2); else split_new(v[i], p0, p1, p2);
      acc ^= p0[0] ^ p0[1] ^ p1[0] ^ p1[1] ^ p2[0] ^ p2[1];
      v[i][0] += 1.0f; v[i][2] += 0.5f;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  f32x4* in; unsigned* out;
  (void)hipMalloc(&in, 256 * 4 * 16); (void)hipMalloc(&out, 2048 * 256 * 4); (void)hipMemset(in, 0x3f, 256 * 4 * 16);
  const int iters = 2000;
  for (int waves = 1; waves <= 2; ++waves)
    for (int m = 0; m < 2; ++m) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256 * waves), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k<1>, dim3(256 * waves), dim3(256), 0, 0, in, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned h[4]; (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f ns per f32x4 piece and wave, check %08x\n", m ? "packed-pair form" : "element-wise form", waves, ms,
             ms * 1e6 / ((double)iters * 4), h[0]);
    }
  return 0;
}
