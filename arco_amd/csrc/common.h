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

// "has this kernel's dynamic-LDS limit been raised on the CURRENT device?" - one bit per device in a per-kernel mask.
// hipFuncSetAttribute acts on the current device's copy of the kernel, and one process may drive several devices (a reference
// trainer wraps the drop-in modules in nn.DataParallel, train_arco_2d.py:227-240): a process-wide flag would leave the second
// device at the 64 KB default and its first launch would fail.  (Benign race: two threads may both set the attribute.)
static inline bool arco_first_on_device(unsigned long long& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
  const unsigned long long bit = 1ull << dev;
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

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
// keep-decision of the dropout mask for element index e (stateless: recomputed in backward).  Round 6: SIXTEEN random bits per element, two
// elements per hash - the mask is also drawn inside the convolution loaders (consumer-side activation, igemm_args.h), where the hash is
// most of the cost; the drop probability is p rounded to a multiple of 2^-16 (|p' - p| < 8e-6; the survivors keep the scale 1 / (1 - p))
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t e, float p) {
  const uint64_t e2 = e >> 1;
  const uint32_t h = pcg_hash((uint32_t)e2 ^ pcg_hash((uint32_t)(e2 >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
  return ((e & 1) ? (h >> 16) : (h & 0xffffu)) >= drop_thr16(p);
}
// the same decisions for the four elements e0 .. e0 + 3 (e0 % 4 == 0, e0 < 2^33) with the seed half hashed once per launch (drop_key32)
__device__ __forceinline__ uint32_t drop_key32(uint64_t seed) { return pcg_hash((uint32_t)seed) ^ (uint32_t)(seed >> 32); }
__device__ __forceinline__ void drop_keep_quad(uint32_t key, uint32_t e0, uint32_t thr, bool (&keep)[4]) {
  const uint32_t h0 = pcg_hash((e0 >> 1) ^ key), h1 = pcg_hash(((e0 >> 1) + 1u) ^ key);
  keep[0] = (h0 & 0xffffu) >= thr; keep[1] = (h0 >> 16) >= thr; keep[2] = (h1 & 0xffffu) >= thr; keep[3] = (h1 >> 16) >= thr;
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
