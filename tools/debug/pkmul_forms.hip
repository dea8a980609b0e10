// Which instruction forms give wrong results in lanes 48-63 while another wave of the SIMD executes MFMAs?  (follow-up of
// tools/debug/pkmul_repro.hip, profiles/r05_notes.md section 1.)  Every checking kernel executes ONE packed-fp32 instruction form
// REPS times per wave on per-lane operands and compares both halves with plain v_mul_f32 / v_add_f32 / v_fma_f32 results.
//   hipcc -O3 --offload-arch=gfx950 pkmul_forms.hip -o pkmul_forms && ./pkmul_forms [seconds per form] [mfma kind]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <chrono>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REPS 8
__device__ __forceinline__ float mkf(unsigned x) {      // a float in [0.25, 1)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return 0.25f + (float)(x & 0xffffff) * (0.75f / 16777216.f);
}
// err layout: [form][rep][group][half] wrong counts, then [form][group] exact-zero counts
#define ERR_STRIDE (REPS * 8 + 4)
template <int FORM>
__global__ __launch_bounds__(256) void form_kernel(unsigned* __restrict__ err, unsigned salt, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 16 + salt * 0x9e3779b9U;
  float acc = 0.f;
  unsigned* e = err + FORM * ERR_STRIDE;
#pragma unroll 1
  for (int r = 0; r < REPS; ++r) {
    f32x2 a = {mkf(id + 4 * r), mkf(id + 4 * r + 1)}, b = {mkf(id + 4 * r + 2), mkf(id + 4 * r + 3)}, c = {mkf(id + r + 77), mkf(id + r + 99)}, d;
    float lo, hi;
    if (FORM == 0) {        // cross selection, distinct pairs: the product kernel's mixed weights
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(b[1])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a[1]), "v"(b[0]));
    } else if (FORM == 1) { // plain packed multiply
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(b[0])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a[1]), "v"(b[1]));
    } else if (FORM == 2) { // cross selection, one pair
      asm volatile("v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(a[1])); hi = lo;
    } else if (FORM == 3) { // broadcast of one half (the compiler's scalar * vector form)
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(b[0])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a[0]), "v"(b[1]));
    } else if (FORM == 4) { // packed add, cross selection
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(a), "v"(b));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(lo) : "v"(a[0]), "v"(b[1])); asm volatile("v_add_f32 %0, %1, %2" : "=v"(hi) : "v"(a[1]), "v"(b[0]));
    } else if (FORM == 5) { // packed fma, plain
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a[0]), "v"(b[0]), "v"(c[0])); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
    } else if (FORM == 6) { // packed fma, cross selection on the second source
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a[0]), "v"(b[1]), "v"(c[0])); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a[1]), "v"(b[0]), "v"(c[1]));
    } else if (FORM == 7) { // swapped halves of the second source only: op_sel:[0,1] op_sel_hi:[1,0] is exactly this - here on the FIRST source
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(a[1]), "v"(b[0])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(a[0]), "v"(b[1]));
    } else if (FORM == 8) { // inline constant with selection (the product's hy = 1 - ly)
      asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=&v"(d) : "v"(a));
      asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(lo) : "v"(a[0])); asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(hi) : "v"(a[1]));
    } else if (FORM == 9) { // 64-bit integer VALU for comparison (addresses): v_lshl_add_u64
      unsigned long long x = ((unsigned long long)__float_as_uint(a[0]) << 32) | __float_as_uint(a[1]), y = ((unsigned long long)__float_as_uint(b[0]) << 20) | __float_as_uint(b[1]), z;
      asm volatile("v_lshl_add_u64 %0, %1, 3, %2" : "=&v"(z) : "v"(x), "v"(y));
      const unsigned long long zr = (x << 3) + y;
      d[0] = __uint_as_float((unsigned)z); d[1] = __uint_as_float((unsigned)(z >> 32)); lo = __uint_as_float((unsigned)zr); hi = __uint_as_float((unsigned)(zr >> 32));
    } else {                // plain 32-bit multiply as the control
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d[0]) : "v"(a[0]), "v"(b[1])); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d[1]) : "v"(a[1]), "v"(b[0]));
      lo = a[0] * b[1]; hi = a[1] * b[0];
    }
    if (__float_as_uint(d[0]) != __float_as_uint(lo)) { atomicAdd(e + (r * 4 + g) * 2, 1u); if (d[0] == 0.f) atomicAdd(e + REPS * 8 + g, 1u); }
    if (__float_as_uint(d[1]) != __float_as_uint(hi)) { atomicAdd(e + (r * 4 + g) * 2 + 1, 1u); if (d[1] == 0.f) atomicAdd(e + REPS * 8 + g, 1u); }
    acc += d[0] + d[1];
  }
  if (acc == 12345.f) sink[0] = acc;
}

template <int KIND>
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  if (KIND == 0) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 3 * i)); }
    for (int it = 0; it < iters; ++it) { acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0); }
  } else if (KIND == 1) {
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    for (int it = 0; it < iters; ++it) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0); }
  } else if (KIND == 2) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 3 * i)); }
    for (int it = 0; it < iters; ++it) { acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0); }
  }
  const f32x4 s = acc0 + acc1;
  if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define NFORMS 11
template <int F> static void launch_form(hipStream_t s, unsigned* err, unsigned salt, float* sink) {
  hipLaunchKernelGGL(form_kernel<F>, dim3(4096), dim3(256), 0, s, err, salt, sink);
}
int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  const int same_stream = argc > 3 ? atoi(argv[3]) : 0;
  float *sink, *junk; unsigned* err;
  CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&junk, 4096)); CK(hipMalloc(&err, NFORMS * ERR_STRIDE * 4));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  if (same_stream) sb = sa;
  const char* names[NFORMS] = {"pk_mul cross sel, two pairs", "pk_mul plain", "pk_mul cross sel, one pair", "pk_mul broadcast op_sel_hi:[0,1]", "pk_add cross sel",
                               "pk_fma plain", "pk_fma cross sel", "pk_mul cross sel on src0", "pk_add 1.0 neg sel", "v_lshl_add_u64", "v_mul_f32 control"};
  const char* kinds[] = {"mfma_f32_16x16x32_bf16", "mfma_f32_16x16x4_f32", "mfma_f32_16x16x32_f16", "no co-tenant"};
  printf("co-tenant: %s%s\n", kinds[kind], same_stream ? " (SAME stream: no concurrency)" : "");
  CK(hipMemset(err, 0, NFORMS * ERR_STRIDE * 4));
  for (int f = 0; f < NFORMS; ++f) {
    CK(hipDeviceSynchronize());
    long launches = 0; unsigned salt = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int rep = 0; rep < 10; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(mfma_spin_kernel<0>, dim3(1024), dim3(256), 0, sb, junk, 4000);
        if (kind == 1) hipLaunchKernelGGL(mfma_spin_kernel<1>, dim3(1024), dim3(256), 0, sb, junk, 2000);
        if (kind == 2) hipLaunchKernelGGL(mfma_spin_kernel<2>, dim3(1024), dim3(256), 0, sb, junk, 4000);
        for (int k = 0; k < 8; ++k) {
          switch (f) {
            case 0: launch_form<0>(sa, err, salt, sink); break; case 1: launch_form<1>(sa, err, salt, sink); break;
            case 2: launch_form<2>(sa, err, salt, sink); break; case 3: launch_form<3>(sa, err, salt, sink); break;
            case 4: launch_form<4>(sa, err, salt, sink); break; case 5: launch_form<5>(sa, err, salt, sink); break;
            case 6: launch_form<6>(sa, err, salt, sink); break; case 7: launch_form<7>(sa, err, salt, sink); break;
            case 8: launch_form<8>(sa, err, salt, sink); break; case 9: launch_form<9>(sa, err, salt, sink); break;
            default: launch_form<10>(sa, err, salt, sink); break;
          }
          ++launches; ++salt;
        }
      }
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    }
    unsigned he[ERR_STRIDE];
    CK(hipMemcpy(he, err + f * ERR_STRIDE, sizeof(he), hipMemcpyDeviceToHost));
    unsigned long tot = 0, lo_[4] = {0, 0, 0, 0}, hi_[4] = {0, 0, 0, 0}, byrep[REPS] = {0};
    for (int r = 0; r < REPS; ++r) for (int g = 0; g < 4; ++g) { lo_[g] += he[(r * 4 + g) * 2]; hi_[g] += he[(r * 4 + g) * 2 + 1]; byrep[r] += he[(r * 4 + g) * 2] + he[(r * 4 + g) * 2 + 1]; }
    for (int g = 0; g < 4; ++g) tot += lo_[g] + hi_[g];
    printf("%-34s: %7ld launches x 16384 waves x %d, wrong %8lu | lo half by 16-lane group [%lu %lu %lu %lu] hi half [%lu %lu %lu %lu] | exact zeros by group [%u %u %u %u] | by repetition [",
           names[f], launches, REPS, tot, lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3], he[REPS * 8], he[REPS * 8 + 1], he[REPS * 8 + 2], he[REPS * 8 + 3]);
    for (int r = 0; r < REPS; ++r) printf("%lu ", byrep[r]);
    printf("]\n"); fflush(stdout);
  }
  return 0;
}
