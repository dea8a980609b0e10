// Instrumented twins of lerp4_cat_rows_bwd_kernel (arco_amd/csrc/elementwise.hip) for the two-queue hazard probes
// (profiles/r05_notes.md section 1).  NOT part of the product library: built by tools/debug/build_dbg.sh into
// tools/debug/liblerp4dbg.so and loaded by tools/debug/self_consistency.py when SC_DBG_LERP is set.
//   variant 0: the product kernel's code + a per-lane record of what the stores used (weights, row addresses, hardware ids)
//   variant 1: the four row stores as write-through stores (sc0 sc1: leave the XCD's L2)
//   variant 2: weights recomputed inside the loop (no loop-invariant weight registers)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REC 16
template <int VARIANT>
__global__ __launch_bounds__(256) void lerp4_bwd_dbg_kernel(const float* __restrict__ dX, long ldx, int Clo,
                                                           const float* __restrict__ lylx, const int64_t* __restrict__ pix,
                                                           long n, float* __restrict__ dV, long ldv,
                                                           float* __restrict__ dhi, long ldhi, int Chi, uint32_t* __restrict__ dbg) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const uint64_t t0 = __builtin_readcyclecounter();
  const float ly = lylx[2 * j], lx = lylx[2 * j + 1], hy = 1.f - ly, hx = 1.f - lx;
  const float* g = dX + j * ldx;
  float* v0 = dV + (4 * j) * ldv;
  float w0 = hy * hx, w1 = hy * lx, w2 = ly * hx, w3 = ly * lx;
  float dfirst = 0.f;
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(g + c);
    if (c < 256) dfirst = d[0];
    if (VARIANT == 2) {
      const float ly2 = __builtin_nontemporal_load(lylx + 2 * j), lx2 = __builtin_nontemporal_load(lylx + 2 * j + 1);
      w0 = (1.f - ly2) * (1.f - lx2); w1 = (1.f - ly2) * lx2; w2 = ly2 * (1.f - lx2); w3 = ly2 * lx2;
    }
    const f32x4 r0 = d * w0, r1 = d * w1, r2 = d * w2, r3 = d * w3;
    if (VARIANT == 1) {
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(v0 + c), "v"(r0) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(v0 + ldv + c), "v"(r1) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(v0 + 2 * ldv + c), "v"(r2) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(v0 + 3 * ldv + c), "v"(r3) : "memory");
    } else {
      *reinterpret_cast<f32x4*>(v0 + c) = r0;
      *reinterpret_cast<f32x4*>(v0 + ldv + c) = r1;
      *reinterpret_cast<f32x4*>(v0 + 2 * ldv + c) = r2;
      *reinterpret_cast<f32x4*>(v0 + 3 * ldv + c) = r3;
    }
  }
  float* h = dhi + pix[j] * ldhi;
  for (int c = lane; c < Chi; c += 64) atomicAdd(h + c, g[Clo + c]);
  if (dbg) {
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t* r = dbg + (j * 64 + lane) * REC;
    r[0] = __float_as_uint(w0); r[1] = __float_as_uint(w1); r[2] = __float_as_uint(w2); r[3] = __float_as_uint(w3);
    const int c = lane * 4;
    r[4] = (uint32_t)((const char*)(v0 + c) - (const char*)dV); r[5] = (uint32_t)((const char*)(v0 + ldv + c) - (const char*)dV);
    r[6] = (uint32_t)((const char*)(v0 + 2 * ldv + c) - (const char*)dV); r[7] = (uint32_t)((const char*)(v0 + 3 * ldv + c) - (const char*)dV);
    r[8] = hw_id; r[9] = xcc_id; r[10] = (uint32_t)t0; r[11] = (uint32_t)t1;
    r[12] = __float_as_uint(dfirst); r[13] = threadIdx.x; r[14] = __float_as_uint(ly); r[15] = __float_as_uint(lx);
  }
}

extern "C" int dbg_lerp4_cat_rows_bwd(int variant, const float* dX, long ldx, int Clo, const float* lylx, const int64_t* pix, long n,
                                      float* dV, long ldv, float* dhi, long ldhi, int Chi, uint32_t* dbg, void* stream) {
  if (n == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((n + 3) / 4), block(256);
  if (variant == 1) hipLaunchKernelGGL(lerp4_bwd_dbg_kernel<1>, grid, block, 0, s, dX, ldx, Clo, lylx, pix, n, dV, ldv, dhi, ldhi, Chi, dbg);
  else if (variant == 2) hipLaunchKernelGGL(lerp4_bwd_dbg_kernel<2>, grid, block, 0, s, dX, ldx, Clo, lylx, pix, n, dV, ldv, dhi, ldhi, Chi, dbg);
  else hipLaunchKernelGGL(lerp4_bwd_dbg_kernel<0>, grid, block, 0, s, dX, ldx, Clo, lylx, pix, n, dV, ldv, dhi, ldhi, Chi, dbg);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- hand-edited ISA variants of the product kernel (tools/debug/isa/*.s -> *.co), loaded as modules
static hipModule_t g_mod = nullptr;
static hipFunction_t g_fn = nullptr;
extern "C" int dbg_lerp4_module_load(const char* path) {
  if (hipModuleLoad(&g_mod, path) != hipSuccess) return -1;
  if (hipModuleGetFunction(&g_fn, g_mod, "lerp4_bwd_isa") != hipSuccess) return -2;
  return 0;
}
extern "C" int dbg_lerp4_module_launch(const float* dX, long ldx, int Clo, const float* lylx, const int64_t* pix, long n, float* dV,
                                       long ldv, float* dhi, long ldhi, int Chi, uint32_t* dbg, void* stream) {
  if (n == 0) return 0;
  void* args[] = {&dX, &ldx, &Clo, &lylx, &pix, &n, &dV, &ldv, &dhi, &ldhi, &Chi, &dbg};     // dbg: only variants whose metadata declares it read it
  return hipModuleLaunchKernel(g_fn, (unsigned)((n + 3) / 4), 1, 1, 256, 1, 1, 0, reinterpret_cast<hipStream_t>(stream), args, nullptr) == hipSuccess ? 0 : -2;
}
