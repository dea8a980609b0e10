// Does the bf16 three-way split of the loader waves overlap with the MFMA waves on the same SIMD?
// waves 0-3: MFMA loop; waves 4-7: split loop, (a) v_cvt_pk_bf16_f32 + v_sub_f32 (fp pipeline), (b) integer rounding + v_sub_f32.
// Also checks that (a) and (b) produce identical bits.   hipcc -O3 --offload-arch=gfx950 split_overlap.hip -o split_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_fp(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
  const __bf16 b0 = (__bf16)v; const float r1 = v - (float)b0;
  const __bf16 b1 = (__bf16)r1; const float r2 = r1 - (float)b1;
  const __bf16 b2 = (__bf16)r2;
  h0 = __builtin_bit_cast(unsigned short, b0); h1 = __builtin_bit_cast(unsigned short, b1); h2 = __builtin_bit_cast(unsigned short, b2);
}
__device__ __forceinline__ unsigned rne_hi(unsigned u) { return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; }   // finite inputs
__device__ __forceinline__ void split_int(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
  const unsigned u0 = rne_hi(__builtin_bit_cast(unsigned, v)); const float r1 = v - __builtin_bit_cast(float, u0);
  const unsigned u1 = rne_hi(__builtin_bit_cast(unsigned, r1)); const float r2 = r1 - __builtin_bit_cast(float, u1);
  const unsigned u2 = rne_hi(__builtin_bit_cast(unsigned, r2));
  h0 = u0 >> 16; h1 = u1 >> 16; h2 = u2 >> 16;
}
__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_split, const float* in, unsigned* out) {
  const int wid = threadIdx.x >> 6;
  if (wid < 4) {
    if (!(mode & 1)) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n_mfma; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = __builtin_bit_cast(unsigned, c0[0] + c1[1] + c2[2] + c3[3]);
  } else {
    if (!(mode & 6)) return;
    float v[8]; unsigned acc = 0;
    for (int e = 0; e < 8; ++e) v[e] = in[(threadIdx.x * 8 + e) & 4095];
    for (int i = 0; i < n_split; ++i) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned h0, h1, h2;
        if (mode & 2) split_fp(v[e], h0, h1, h2); else split_int(v[e], h0, h1, h2);
        acc ^= h0 + (h1 << 8) + (h2 << 16);
        v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[e]) ^ ((h2 & 1u) << 3));      // keep the loop honest
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
  }
}
__global__ void check(const float* in, int n, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned a0, a1, a2, b0, b1, b2;
  split_fp(in[i], a0, a1, a2); split_int(in[i], b0, b1, b2);
  if (a0 != b0 || a1 != b1 || a2 != b2) atomicAdd(bad, 1u);
}
int main() {
  const int N = 1 << 22;
  float* h = (float*)malloc(N * 4);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < N; ++i) {                       // random bit patterns with finite exponents (incl. tiny / huge magnitudes, some denormals)
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    unsigned u = (unsigned)s; if (((u >> 23) & 0xFF) == 0xFF) u &= 0xFF7FFFFFu;
    if (i % 1000 == 0) u &= 0x807FFFFFu;            // a denormal
    memcpy(&h[i], &u, 4);
  }
  float* in; unsigned* out; unsigned* bad; hipMalloc(&in, N * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&bad, 4);
  hipMemcpy(in, h, N * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(check, dim3(N / 256), dim3(256), 0, 0, in, N, bad);
  unsigned nb; hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
  printf("integer-rounded split vs cvt split: %u of %d values differ\n", nb, N);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_mfma = 20000, n_split = 2000;
  for (int mode : {1, 2, 3, 4, 5}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_split, in, out); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("mode %d (%s): %.3f ms\n", mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "cvt split only" : mode == 3 ? "MFMA + cvt split" : mode == 4 ? "int split only" : "MFMA + int split", ms);
  }
  return 0;
}
