// the product kernel lerp4_cat_rows_bwd_kernel (arco_amd/csrc/elementwise.hip), alone in a translation unit, as the base of the
// hand-edited ISA variants of tools/debug/isa/*.s (profiles/r05_notes.md section 1)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void lerp4_bwd_isa(const float* __restrict__ dX, long ldx, int Clo,
                                                                const float* __restrict__ lylx, const int64_t* __restrict__ pix,
                                                                long n, float* __restrict__ dV, long ldv,
                                                                float* __restrict__ dhi, long ldhi, int Chi) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float ly = lylx[2 * j], lx = lylx[2 * j + 1], hy = 1.f - ly, hx = 1.f - lx;
  const float* g = dX + j * ldx;
  float* v0 = dV + (4 * j) * ldv;
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(g + c);
    *reinterpret_cast<f32x4*>(v0 + c) = d * (hy * hx);
    *reinterpret_cast<f32x4*>(v0 + ldv + c) = d * (hy * lx);
    *reinterpret_cast<f32x4*>(v0 + 2 * ldv + c) = d * (ly * hx);
    *reinterpret_cast<f32x4*>(v0 + 3 * ldv + c) = d * (ly * lx);
  }
  float* h = dhi + pix[j] * ldhi;
  for (int c = lane; c < Chi; c += 64) atomicAdd(h + c, g[Clo + c]);
}
