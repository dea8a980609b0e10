// Helpers shared by the software-pipelined convolution kernels (conv_sp.hip: 3x3; conv3d_fl.hip: 3x3x3): LDS-DMA address-space
// types, counted waits, DPP row sums, the asm MFMA forms with pinned accumulators, the CU count of the device.
#pragma once
#include "igemm_args.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;


// a wave-uniform pointer moved into SGPRs (v_readfirstlane): the scalar-base operand of a global load must not sit in VGPRs,
// where the register allocator may leave a value it knows to be uniform (seen under register pressure: "invalid operand")
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// lgkmcnt(0) through the builtin: hipcc models the instruction, so it knows the fragment registers read in the previous
// step have landed (after an asm wait it re-waits lgkmcnt(0) in front of the first MFMA, behind 24 fresh reads)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {      // v + (v of the lane the DPP control names)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {    // sum over the 16 lanes of a DPP row, in every lane
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  return dpp_add<0x140>(v);  // row_mirror
}

// The accumulators are pinned in AGPRs ("+a"): left to itself hipcc keeps them in VGPRs next to 150 fragment registers,
// runs out, and shuttles every accumulator through a scratch AGPR quad around each MFMA chain (4 v_accvgpr_write + hazard
// nops per chain).  The s_nop covers an operand a VALU instruction has just written (hipcc pads nothing inside asm).
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& x, const bf16x8& y) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
}

__device__ __forceinline__ void mfma_first(f32x4& c, const bf16x8& x, const bf16x8& y) {     // c = x . y (a tile's first product)
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(x), "v"(y));
}

static int conv_sp_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

