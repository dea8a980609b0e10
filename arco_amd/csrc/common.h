// Shared helpers for the arco_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ARCO_OK 0
#define ARCO_ERR_ARG (-1)
#define ARCO_ERR_LAUNCH (-2)
#define ARCO_ERR_UNSUPPORTED (-3)

#define ARCO_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return ARCO_ERR_ARG; \
  } while (0)

#include <stdio.h>
static inline int arco_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "arco_hip: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
  return e == hipSuccess ? ARCO_OK : ARCO_ERR_LAUNCH;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
// four consecutive channels of an activation row as fp32, from fp32 or f16 storage (the *_h entry points: f16 activation storage)
__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4f(const _Float16* p) { return __builtin_convertvector(*reinterpret_cast<const f16x4_t*>(p), f32x4); }
__device__ __forceinline__ void st4f(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4f(_Float16* p, f32x4 v) { *reinterpret_cast<f16x4_t*>(p) = __builtin_convertvector(v, f16x4_t); }
// eight consecutive channels of an f16 row in one 16-byte access (the BN apply passes on f16 activation storage)
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8 ld8f(const _Float16* p) { return __builtin_convertvector(*reinterpret_cast<const f16x8_t*>(p), f32x8); }
__device__ __forceinline__ void st8f(_Float16* p, f32x8 v) { *reinterpret_cast<f16x8_t*>(p) = __builtin_convertvector(v, f16x8_t); }
__device__ __forceinline__ f32x8 ld8p(const float* p) {      // eight per-channel parameters
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- the stateless dropout mask (nn.Dropout / nn.Dropout3d of the blocks, unetWithArgs.py:43, vnetWithArgs.py:195,238)
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
// keep-decision of the dropout mask for element index e (stateless: recomputed in backward)
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t e, float p) {
  const uint32_t h = pcg_hash((uint32_t)e ^ pcg_hash((uint32_t)(e >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}
// the same decision for element indices below 2^32 with the seed half hashed once per launch (drop_key32): the loaders of the
// convolution kernels evaluate it per staged element
__device__ __forceinline__ uint32_t drop_key32(uint64_t seed) { return pcg_hash((uint32_t)seed) ^ (uint32_t)(seed >> 32); }
__device__ __forceinline__ bool drop_keep32(uint32_t key, uint32_t e, float p) {
  return (float)(pcg_hash(e ^ key) >> 8) * (1.0f / 16777216.0f) >= p;
}

// bit layout of the per-pixel class code (C <= 21)
#define ARCO_MAXC 21
#define ARCO_BIT_LV(c) (c)
#define ARCO_BIT_ANCHOR(c) (21 + (c))
#define ARCO_BIT_NEG(c) (42 + (c))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
