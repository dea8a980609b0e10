// Argument block and split-bf16 helpers shared by the implicit-GEMM kernels (igemm.hip) and the pipelined 3x3 kernel
// (conv_sp.hip).
#pragma once
#include "common.h"

// include/arco_hip.h: ArcoActPro (the C-ABI descriptor of a consumer-side activation); repeated here for the kernels
struct ArcoActPro {
  const float* mean; const float* istd; const float* gamma; const float* beta;   // [groups][K] statistics, [K] affine parameters
  float slope;                       // LeakyReLU slope (0: ReLU)
  int groups;                        // BatchNorm groups of the producing layer: images [g*NB/G, (g+1)*NB/G) use statistics row g
  int drop_mode; float p;            // 0: none; 1: nn.Dropout(p) with the stateless mask of bn_act_fwd_kernel
  unsigned long long seed; const unsigned long long* seed_dev;
};

struct IgemmArgs {
  const float* A; long lda;
  const float* Wp; int Npad, Kpad, N, K;
  float* C; long ldc;
  const float* bias;
  const float* R; long ldr;
  float* stat_sum; float* stat_sq;   // [N][n_mblocks] block partials (nullable)
  int n_mblocks; int n_nblocks;
  int NB, H, W;                      // images (planes for 3-D), rows, cols (TAPS==9); TAPS==1 uses M only
  int D3;                            // 3-D: planes per volume (depth taps active when DEPTH==3); 2-D: 1
  long M;                            // total pixels
  int ksplit; long slab_stride;      // split-K (TAPS==1): blockIdx.y = K slab, output slab y at C + y*slab_stride
  int stat_groups;                   // BN groups: images [g*NB/G, (g+1)*NB/G) feed the stat slabs [g*n_mblocks/G, ...)
  int mma;                           // 0: fp32 MFMA (default); 1 / 2: operands rounded to f16 / bf16 in registers, fp32 accumulate (3x3x3, 1x1x1);
                                     // 3: split-bf16; 4: f16 activation STORAGE (A / C are f16 tensors, conv_h.hip)
  int Kg;                            // mma == 3: 16-k groups per packed weight row (= ceil32(K) / 16)
  int batch; long batchA, batchW, batchC;   // TAPS==1 batched GEMM: blockIdx.z = problem, operands at + z * stride (floats)
  // gemm_sp.hip only: the residual is the TRILINEAR (align_corners) upsample of a low-resolution tensor, sampled in the epilogue -
  // FeatureExtractor_3d's `fea_i(cat(up(x), f_i)) + cat(up(x), f_i)` with the wide block pushed under the upsample
  // (model_3D.py:46-58; arco_amd/model_3D.py forward_lowres2): Rup [NV, uD, uH, uW] rows of N channels -> output rows [NV, oD, oH, oW]
  const float* Rup; long ldrup; int uD, uH, uW, oD, oH, oW;
  // Consumer-side activation (pro.mean != nullptr): A holds the PRE-activation z of the producing convolution, and the loader forms
  // a = dropout(lrelu((z - mean) * istd * gamma + beta)) element by element while it stages the tile - the BatchNorm-apply pass
  // between the two convolutions of a block (unetWithArgs.py:36-44, vnetWithArgs.py:16-25) and its HBM round trip are gone.
  // The arithmetic is bn_act_fwd_kernel's, operation for operation: the staged values are bit-identical to the tensor that pass wrote.
  ArcoActPro pro;
};

// the per-channel parameters of four consecutive channels and the prologue applied to one staged quad
struct ProQuad { f32x4 mu, is, ga, be; };
__device__ __forceinline__ f32x4 pro_bn_lrelu(f32x4 z, const ProQuad& q, float slope) {
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = (z[e] - q.mu[e]) * q.is[e] * q.ga[e] + q.be[e];
    y[e] = v >= 0.f ? v : v * slope;
  }
  return y;
}
// v where keep != 0, +0.0 elsewhere - as a bitwise AND with an all-ones / all-zeros word.  Written `keep ? v : 0` behind the prologue
// arithmetic, hipcc sinks that arithmetic into per-element branches (wgrad_split_kernel<16,16,PRO>: 111 us against 72 us for either
// the arithmetic or the select alone, tools/micro/wgrad_pro_bench.py); the AND keeps the loader straight-line
__device__ __forceinline__ f32x4 pro_mask(f32x4 v, unsigned keep) {
  // (whole-vector bit casts: `__builtin_bit_cast(unsigned, v[e])` on a vector ELEMENT reads element 0 for every e with this clang)
  typedef unsigned int pm_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned m = 0u - (keep & 1u);
  const pm_u32x4 u = __builtin_bit_cast(pm_u32x4, v) & pm_u32x4{m, m, m, m};
  return __builtin_bit_cast(f32x4, u);
}
// nn.Dropout on the quad at element index e0 .. e0 + 3 (element = pixel * C + channel, as bn_act_fwd_kernel counts)
__device__ __forceinline__ f32x4 pro_dropout(f32x4 y, uint32_t key, uint32_t e0, uint32_t thr, float keep_scale) {
  bool keep[4];
  drop_keep_quad(key, e0, thr, keep);
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = keep[e] ? y[e] * keep_scale : 0.f;
  return y;
}

// The 8-corner trilinear blend as ONE fixed chain of multiplies and fused multiply-adds: every kernel that interpolates (the resize
// kernel, the row gather of the row-sparse head, the GEMM epilogue of gemm_sp.hip) evaluates exactly this chain, so their results
// agree bit for bit whatever the surrounding code lets the compiler contract (the plain expression `hz * (...) + lz * (...)` left
// the choice of which product of each sum becomes the fma to the instruction scheduler: the fused epilogue differed from the resize
// kernel in 1-2 ulp on 80 % of the elements)
__device__ __forceinline__ float tl_blend1(float v000, float v001, float v010, float v011, float v100, float v101, float v110, float v111,
                                           float hx, float lx, float hy, float ly, float hz, float lz) {
  const float a0 = __builtin_fmaf(lx, v001, hx * v000), a1 = __builtin_fmaf(lx, v011, hx * v010);
  const float a2 = __builtin_fmaf(lx, v101, hx * v100), a3 = __builtin_fmaf(lx, v111, hx * v110);
  const float b0 = __builtin_fmaf(ly, a1, hy * a0), b1 = __builtin_fmaf(ly, a3, hy * a2);
  return __builtin_fmaf(lz, b1, hz * b0);
}

// align_corners source index / weight (torch upsample index math in fp32; shared by the resize kernels and the fused epilogue)
// (contraction off: `src - i0` must subtract from the ROUNDED product, as torch's CPU kernel does - left to -ffp-contract=fast the
//  compiler fused it into fma(scale, o, -i0) in some kernels and not in others: interpolation weights one ulp apart)
__device__ __forceinline__ void ac_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
#ifndef ARCO_AC_FAST               // (A/B only: the pre-round-5 behaviour)
#pragma clang fp contract(off)
#endif
  const float src = scale * (float)o;
  i0 = (int)src; if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 < in_size - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// LDS words are written as packed bf16 pairs and read back as MFMA operands: every such type-punned access goes through a
// may_alias type (without it the compiler may assume that the loads cannot see the stores - observed: a B fragment built
// from one repeated dword)
typedef unsigned int u32x2_ma __attribute__((ext_vector_type(2), may_alias));
typedef unsigned int u32x4_ma __attribute__((ext_vector_type(4), may_alias));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_bf16x8(const void* p) { return __builtin_bit_cast(bf16x8, u32x4(*reinterpret_cast<const u32x4_ma*>(p))); }
// x = b0 + b1 + b2 exactly, b_i bf16 (RNE remainders): the split of an fp32 operand into its three bf16 MFMA terms.
// A packed-pair form written at the instruction level (-DARCO_SPLIT_PACKED: per PAIR of elements one v_cvt_pk_bf16_f32 per term - its
// result IS the packed LDS word -, two bit operations to widen a term back to fp32, one v_pk_add_f32 for the remainder: 10 VALU
// instructions per pair where the element-wise C++ form compiles to 14) is 21-23 % faster as pure VALU work (tools/micro/split_rate.hip:
// 53 vs 67 ns per f32x4 piece and wave) and bit-identical - and changes nothing where it matters: same box, alternating, the dense
// GEMM 160.4 vs 159.4 TFLOP/s, the headline step 11.17 / 11.56 vs 11.08 / 11.21 ms (tools/debug/ab_split.sh, round 5).  The
// kernels that split are bound by on-chip operand traffic and rendezvous, not by the loaders' instruction count
// (profiles/r05_notes.md section 2).  Kept behind the macro; the element-wise form stays the default.
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = cvt_pk_bf16(x0, x1);
  const f32x2_t x = {x0, x1};
  const f32x2_t f0 = {__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h0 & 0xffff0000u)};
  const f32x2_t r1 = x - f0;
  h1 = cvt_pk_bf16(r1[0], r1[1]);
  const f32x2_t f1 = {__builtin_bit_cast(float, h1 << 16), __builtin_bit_cast(float, h1 & 0xffff0000u)};
  const f32x2_t r2 = r1 - f1;
  h2 = cvt_pk_bf16(r2[0], r2[1]);
}
// The ACTIVATION split (round 6): the same exact three-term decomposition with fewer vector instructions.  t0 = bf16(x) by the hardware's
// round-to-nearest conversion of a PAIR (v_cvt_pk_bf16_f32: its result is the packed LDS word of plane 0); r1 = x - t0 (exact); t1 = the upper
// 16 bits of r1 (truncation: one v_and); t2 = r1 - t1 (exact, at most 8 significant bits: a bf16 value as it stands); x = t0 + t1 + t2 bit for bit.
// Eleven full-rate instructions per pair of elements (22 per 16-byte piece) where the all-round-to-nearest form (-DARCO_SPLIT_RNE: rounds 2-5; still
// what the weight pack kernels do) compiles to ~34.  The loader waves of the pipelined kernels share their SIMD's vector issue port with an MFMA wave
// and were the side the rendezvous waited for (tools/micro/fc_clock.py, profiles/r06_notes.md section 13).  Rounding the FIRST term keeps the
// remainders zero-mean: truncating it too (-DARCO_SPLIT_TRUNC0, measured) biases every dropped cross product the same way, the bias adds up over K,
// and the whole V-Net's gradients moved from 0.4 % to 0.9 % (worst tensor 1.4 % -> 9 %) off the float64 reference through extra ReLU flips.
// |t1| <= 2^-8 |x|, |t2| < 2^-15 |x|: the three dropped cross products are bounded by 2^-23 + 2^-24 of |x||w| (all-RNE: 2^-23), zero-mean.
__device__ __forceinline__ unsigned perm_hi16(unsigned hi, unsigned lo) {      // {hi[31:16], lo[31:16]}
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
__device__ __forceinline__ void split3_pair_rt(float x0, float x1, unsigned& h0, unsigned& h1, unsigned& h2) {
#ifdef ARCO_SPLIT_TRUNC0
  h0 = perm_hi16(__float_as_uint(x1), __float_as_uint(x0));
#else
  h0 = cvt_pk_bf16(x0, x1);
#endif
  const float r0 = x0 - __uint_as_float(h0 << 16), r1 = x1 - __uint_as_float(h0 & 0xffff0000u);
  const unsigned u0 = __float_as_uint(r0) & 0xffff0000u, u1 = __float_as_uint(r1) & 0xffff0000u;
  h1 = perm_hi16(u1, u0);
  h2 = perm_hi16(__float_as_uint(r1 - __uint_as_float(u1)), __float_as_uint(r0 - __uint_as_float(u0)));
}
__device__ __forceinline__ void split3_bf16x4(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
#ifndef ARCO_SPLIT_RNE
  {
    const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    unsigned a0, a1, a2, c0, c1, c2;
    split3_pair_rt(x0, x1, a0, a1, a2);
    split3_pair_rt(x2, x3, c0, c1, c2);
    p0 = u32x2{a0, c0}; p1 = u32x2{a1, c1}; p2 = u32x2{a2, c2};
    return;
  }
#endif
#ifndef ARCO_SPLIT_PACKED           // default: the element-wise form of rounds 2-4 (see the note above split3_pair)
  unsigned short h[3][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 b0 = (__bf16)v[e];
    const float r1 = v[e] - (float)b0;
    const __bf16 b1 = (__bf16)r1;
    const float r2 = r1 - (float)b1;
    const __bf16 b2 = (__bf16)r2;
    h[0][e] = __builtin_bit_cast(unsigned short, b0); h[1][e] = __builtin_bit_cast(unsigned short, b1);
    h[2][e] = __builtin_bit_cast(unsigned short, b2);
  }
  p0 = u32x2{(unsigned)h[0][0] | ((unsigned)h[0][1] << 16), (unsigned)h[0][2] | ((unsigned)h[0][3] << 16)};
  p1 = u32x2{(unsigned)h[1][0] | ((unsigned)h[1][1] << 16), (unsigned)h[1][2] | ((unsigned)h[1][3] << 16)};
  p2 = u32x2{(unsigned)h[2][0] | ((unsigned)h[2][1] << 16), (unsigned)h[2][2] | ((unsigned)h[2][3] << 16)};
  return;
#endif
  unsigned a0, a1, a2, b0, b1, b2;
  split3_pair(v[0], v[1], a0, a1, a2);
  split3_pair(v[2], v[3], b0, b1, b2);
  p0 = u32x2{a0, b0}; p1 = u32x2{a1, b1}; p2 = u32x2{a2, b2};
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 to_f16x4(f32x4 v) { return __builtin_convertvector(v, f16x4); }
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
  s16x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const __bf16 b = (__bf16)v[e]; r[e] = __builtin_bit_cast(short, b); }
  return r;
}

// conv_sp.hip: the software-pipelined split-bf16 3x3 kernel.  Returns -1 when the shape is not one it takes (the caller
// then falls through to igemm_kernel), otherwise the launch status; q != nullptr: query only (q[0] = M-tiles = BN stat
// slabs per channel, q[1] = instantiation id, q[2] = KC*100 + DEPTH*10).
int conv_sp_dispatch(const IgemmArgs& a, hipStream_t st, int* q);
// conv_sp.hip: the resident-weights 3x3x3 kernel of the 16 -> 16 full-resolution level (split-bf16); -1 when the shape is not taken
int conv3d_rw_dispatch(const IgemmArgs& a, hipStream_t st, int* q);
// conv3d_fl.hip: the pipelined flat-tile 3x3x3 kernel of the V-Net levels below full resolution (-1: shape not taken)
int conv3d_fl_dispatch(const IgemmArgs& a, hipStream_t st, int* q);

// gemm_sp.hip: the software-pipelined split-bf16 1x1 GEMM (wide, many-tile launches without BN statistics); -1 when the shape is not taken
int gemm_sp_dispatch(const IgemmArgs& a, hipStream_t st, int* q);

// conv_h.hip: f16 activation storage (mma == 4).  hconv_dispatch: forward / data gradient of the 3x3x3 and 1x1x1 convolutions on
// f16 tensors (a.A, a.C point at f16 rows, a.Wp at the f16 pack [tap][Npad][ceil32(K)], a.Kpad = ceil32(K)); q as above.
// hwgrad_dispatch: their weight gradient from f16 dZ / X (fp32 slabs in ws, fp32 dW).
int hconv_dispatch(const IgemmArgs& a, int taps, hipStream_t st, int* q);
int hwgrad_dispatch(const void* dZ, long ld_dz, int Cout, const void* in, long ld_in, int Cin, int taps, int NB, int D3, int H, int W,
                    float* ws, float* dW, int accumulate, hipStream_t st);
// igemm.hip: fixed-order sum of the weight-gradient slabs [chunk][tap][CoutPad][CinPad] into dW (torch layout)
void launch_wgrad_reduce(hipStream_t st, const float* ws, int chunks, int taps, int CoutPad, int CinPad, int Cout, int Cin,
                         float* dW, int accumulate);
