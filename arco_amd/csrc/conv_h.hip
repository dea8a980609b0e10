// f16 ACTIVATION STORAGE for the volume path (BASELINE.json configs[4], "fp16 MFMA conv": the V-Net body of
// vnetWithArgs.py:5-31,67-118,145-252 with every activation and activation gradient held as f16 in HBM and LDS; weights,
// BatchNorm statistics, loss and optimizer stay fp32).  Kernels:
//
//   hconv_kernel     3x3x3 (planes of H x W, depth tap = outer loop) and 1x1x1 convolutions, forward and - with flipped +
//                    transposed packed weights - data gradient.  f16 tiles go from HBM to LDS untouched (16-byte pieces,
//                    rows of 16 channels unpadded / of 32 channels padded to 96 bytes: conflict-free ds_read_b128 fragments), one
//                    v_mfma_f32_16x16x32_f16 per 16 pixels x 16 channels x 32 k (3x3: the 16 channels of TWO taps),
//                    fp32 accumulate, D = W . X^T so that a lane ends with 4 consecutive channels of one pixel (8-byte
//                    stores); BatchNorm partial statistics of the ROUNDED outputs in the epilogue.  Double-buffered LDS:
//                    chunk c+1 is loaded to registers and written to the other buffer around the MFMAs of chunk c.
//   hwgrad_kernel    weight gradient dW[tap][co][ci] = sum_pix dZ[pix][co] X[pix + tap][ci] (K = pixels).  Both operands
//                    sit in LDS exactly as they sit in HBM ([pixel][channel] rows) and are read through
//                    ds_read_b64_tr_b16 - the transposing LDS read of gfx950 hands every lane 4 consecutive PIXELS of one
//                    channel, which is what the MFMA wants as k - so staging is a plain 16-byte copy (the split-bf16 kernel
//                    transposes and splits on the VALU).  Row strides are 8 x odd dwords: the eight rows a 32-lane half
//                    reads land on eight distinct 8-bank groups.  The four waves of a workgroup take one 32-pixel K step of
//                    a 128-pixel tile each and own ALL taps x sub-tiles (a tap's input fragment feeds CO_T MFMAs); partners
//                    are summed once per launch.  Slabs [chunk][tap][CoutPad][CinPad] as in igemm.hip -> wgrad_reduce.
//   hconv_rw_kernel  the resident-weights form of hconv_kernel for the 16-channel full-resolution level (further down).
//   cast kernels     the fp32 <-> f16 boundary of the V-Net (outputs up, gradients down with the loss scale).
#include "sp_util.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int HFLAT_WPMAX = 64;          // flat-position tiles: padded row stride W + 2 <= 64 (as igemm.hip)

__device__ __forceinline__ h8 lds_h8(const unsigned* p) { return __builtin_bit_cast(h8, u32x4(*reinterpret_cast<const u32x4_ma*>(p))); }

// thread -> piece map of the staging stores.  With 4 pieces per row (32 channels, 24-dword rows) a ds_write_b128 lane octet would hold
// rows r, r + 1, which overlap in 8 of the 32 banks; rows r, r + 2 do not: swap the two low bits of the row index
template <int PIECES> __device__ __forceinline__ int hrow_perm(int r) {
  return PIECES == 4 ? ((r & ~3) | ((r & 1) << 1) | ((r >> 1) & 1)) : r;
}

template <int TAPS, int BM, int BN, int WM, int WN, int DEPTH, bool FLAT>
struct HGeom {
  static constexpr int KC = TAPS == 9 ? 16 : 32;
  // dwords per LDS row: 8 (3x3: 16 channels, no padding) / 24 (1x1: 32 channels + 8).  A ds_read_b128 is served in the lane groups
  // {0-3, 12-15, 20-27}, ...: eight rows r of one k-half and eight rows of the next half (+4 dwords) - conflict-free iff the row
  // stride is an EVEN number s of 4-dword slots (slots s*r are even, s*r + 1 odd, each set distinct mod 16); s = 3 (a 48-byte row)
  // made 5 of every 16 lanes collide
#ifdef ARCO_HCONV_ROWS_48_80      // A/B build: the padded rows measured first (5 of 16 lanes per ds_read_b128 group on a busy bank)
  static constexpr int LDK = KC / 2 + 4;
#else
  static constexpr int LDK = KC == 16 ? 8 : 24;
#endif
  static constexpr int TH = BM / 16;
  static constexpr int AROWS = TAPS == 9 ? (FLAT ? BM + 2 * HFLAT_WPMAX + 2 : (TH + 2) * 18) : BM;
  static constexpr int BROWS = TAPS * BN;
  static constexpr int PA = KC / 8;                         // 16-byte pieces per row
  static constexpr int NA = (AROWS * PA + 255) / 256, NBI = (BROWS * PA + 255) / 256;
  static constexpr int BUF = (AROWS + BROWS + 1) * LDK;     // dwords per buffer (+ the zero weight row)
};

template <int TAPS, int BM, int BN, int WM, int WN, int DEPTH, bool FLAT>
__global__ __launch_bounds__(256) void hconv_kernel(IgemmArgs a) {
  using G = HGeom<TAPS, BM, BN, WM, WN, DEPTH, FLAT>;
  constexpr int KC = G::KC, LDK = G::LDK, AROWS = G::AROWS, BROWS = G::BROWS, PA = G::PA, NA = G::NA, NBI = G::NBI, BUF = G::BUF;
  constexpr int TH = G::TH, A_T = BM / 16 / WM, C_T = BN / 16 / WN;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
  const _Float16* const Ag = reinterpret_cast<const _Float16*>(a.A);
  const _Float16* const Wg = reinterpret_cast<const _Float16*>(a.Wp);
  _Float16* const Cg = reinterpret_cast<_Float16*>(a.C);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN, li = lane & 15, g = lane >> 4;
  int mblk, nblk;
  {   // XCD-aware tile order: the N-tiles of one M-tile (shared input rows) on consecutive slots of one XCD
    const int T = a.n_mblocks * a.n_nblocks, L = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    mblk = v / a.n_nblocks; nblk = v - mblk * a.n_nblocks;
  }
  const int n0 = nblk * BN;
  long m0 = 0; int img = 0, y0 = 0, x0 = 0, f0 = 0;
  const int Wp = a.W + 2;
  if (TAPS == 9 && FLAT) {
    const int per_plane = (a.H * Wp + BM - 1) / BM;
    img = mblk / per_plane; f0 = (mblk - img * per_plane) * BM;
  } else if (TAPS == 9) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + TH - 1) / TH;
    int t = mblk;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; img = t / tiles_y;
    y0 = ty * TH; x0 = tx * 16;
  } else {
    m0 = (long)mblk * BM;
  }
  long m_lim = a.M;
  if (TAPS == 1 && a.stat_groups > 1) {       // GEMM form with grouped BN statistics: M-blocks laid out per group
    const int mpg = a.n_mblocks / a.stat_groups, g_ = mblk / mpg;
    const long Mg = a.M / a.stat_groups;
    m0 = g_ * Mg + (long)(mblk - g_ * mpg) * BM;
    m_lim = (g_ + 1) * Mg;
  }

  f32x4 acc[A_T][C_T];
#pragma unroll
  for (int i = 0; i < A_T; ++i)
#pragma unroll
    for (int j = 0; j < C_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  const _Float16* srcA[NA]; int ldsA[NA], kA[NA];
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int idx = tid + it * 256;
    srcA[it] = nullptr; ldsA[it] = -1; kA[it] = 0;
    if (idx < AROWS * PA) {
      const int q = idx % PA, row = hrow_perm<PA>(idx / PA);
      long pix = -1;
      if (TAPS == 9 && FLAT) {
        const int pidx = f0 + row - 1;
        if (pidx >= 0 && row < BM + 2 * Wp + 2) {
          const int py = pidx / Wp, px = pidx - py * Wp;
          const int y = py - 1, x = px - 1;
          if (y >= 0 && y < a.H && x >= 0 && x < a.W) pix = ((long)img * a.H + y) * a.W + x;
        }
      } else if (TAPS == 9) {
        const int hy = row / 18, hx = row - hy * 18;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        if (y >= 0 && y < a.H && x >= 0 && x < a.W) pix = ((long)img * a.H + y) * a.W + x;
      } else {
        const long m = m0 + row;
        if (m < m_lim) pix = m;
      }
      ldsA[it] = row * LDK + q * 4; kA[it] = 8 * q;
      if (pix >= 0) srcA[it] = Ag + pix * a.lda + 8 * q;
    }
  }
  const _Float16* srcB[NBI]; int ldsB[NBI];
#pragma unroll
  for (int it = 0; it < NBI; ++it) {
    const int idx = tid + it * 256;
    srcB[it] = nullptr; ldsB[it] = -1;
    if (idx < BROWS * PA) {
      const int q = idx % PA, row = hrow_perm<PA>(idx / PA);
      const int tap = row / BN, n = row - tap * BN;
      ldsB[it] = (AROWS + row) * LDK + q * 4;
      if (n0 + n < a.Npad) srcB[it] = Wg + ((long)tap * a.Npad + n0 + n) * a.Kpad + 8 * q;
    }
  }

  const int nchunks = (a.K + KC - 1) / KC;
  const bool vecA = (a.K & 7) == 0 && (a.lda & 7) == 0;
  const int plane = DEPTH == 3 ? img % a.D3 : 0;
  const long plane_elems = (long)a.H * a.W * a.lda;
  const long wslice = (long)9 * a.Npad * a.Kpad;
  u32x4 ra[NA], rb[NBI];
  auto load_chunk = [&](int itc) {
    const int dd = DEPTH == 3 ? itc / nchunks : 0;
    const int kc0 = (DEPTH == 3 ? itc - dd * nchunks : itc) * KC;
    const bool plane_ok = DEPTH == 3 ? (plane + dd - 1 >= 0 && plane + dd - 1 < a.D3) : true;
    const long aoff = DEPTH == 3 ? (long)(dd - 1) * plane_elems + kc0 : kc0;
    const long boff = DEPTH == 3 ? (long)dd * wslice + kc0 : kc0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      u32x4 v = u32x4{0, 0, 0, 0};
      if (TAPS == 1 && !vecA) {        // rows that are not whole 16-byte pieces (the 2-class logits' gradient: K = 2)
        if (srcA[it]) {
          h8 e8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 8; ++e) if (kc0 + kA[it] + e < a.K) e8[e] = srcA[it][aoff + e];
          v = __builtin_bit_cast(u32x4, e8);
        }
      } else if (srcA[it] && plane_ok && kc0 + kA[it] < a.K) v = *reinterpret_cast<const u32x4*>(srcA[it] + aoff);
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
      u32x4 v = u32x4{0, 0, 0, 0};
      if (srcB[it]) v = *reinterpret_cast<const u32x4*>(srcB[it] + boff);
      rb[it] = v;
    }
  };
  auto store_chunk = [&](unsigned* buf) {
#pragma unroll
    for (int it = 0; it < NA; ++it) if (ldsA[it] >= 0) *reinterpret_cast<u32x4_ma*>(buf + ldsA[it]) = ra[it];
#pragma unroll
    for (int it = 0; it < NBI; ++it) if (ldsB[it] >= 0) *reinterpret_cast<u32x4_ma*>(buf + ldsB[it]) = rb[it];
  };
  auto compute = [&](const unsigned* buf) {
    const unsigned* As = buf; const unsigned* Bs = buf + AROWS * LDK;
    constexpr int NSTEP = TAPS == 9 ? 5 : 1;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int tapL = TAPS == 9 ? 2 * st + (g >> 1) : 0;          // lanes g = 0,1: tap 2s, g = 2,3: tap 2s+1
      const bool zt = tapL > 8;                                    // the 10th half-step: zero weight row
      const int tapA = zt ? 8 : tapL;
      const int dy = TAPS == 9 ? tapA / 3 : 0, dx = TAPS == 9 ? tapA % 3 : 0;
      const int koff = TAPS == 9 ? (g & 1) * 4 : g * 4;
      h8 fa[A_T];
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
        const int s = wm * A_T + at;
        const int row = TAPS == 9 ? (FLAT ? s * 16 + li + dy * Wp + dx : (s + dy) * 18 + li + dx) : s * 16 + li;
        fa[at] = lds_h8(As + row * LDK + koff);
      }
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct) {
        const int row = zt ? BROWS : tapA * BN + (wn * C_T + ct) * 16 + li;
        const h8 fb = lds_h8(Bs + row * LDK + koff);
#pragma unroll
        for (int at = 0; at < A_T; ++at)
          acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa[at], acc[at][ct], 0, 0, 0);
      }
    }
  };

  const int niter = DEPTH * nchunks;
  for (int i = tid; i < 2 * LDK; i += 256) smem_u[(i / LDK) * BUF + (AROWS + BROWS) * LDK + i % LDK] = 0u;
  load_chunk(0);
  store_chunk(smem_u);
  __syncthreads();
  for (int c = 0; c < niter; ++c) {
    unsigned* cur = smem_u + (c & 1) * BUF;
    unsigned* nxt = smem_u + ((c + 1) & 1) * BUF;
    const bool more = c + 1 < niter;
    if (more) load_chunk(c + 1);
    compute(cur);
    if (more) store_chunk(nxt);
    __syncthreads();
  }

  // ---- epilogue: bias, round to f16, 8-byte stores, BN partial statistics of the rounded values
  float t1[C_T][4], t2[C_T][4];
  const bool v4 = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 7) == 0);
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct) {
    const int nb = n0 + (wn * C_T + ct) * 16 + 4 * g;
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[r] = (a.bias && nb + r < a.N) ? a.bias[nb + r] : 0.f; t1[ct][r] = 0.f; t2[ct][r] = 0.f; }
#pragma unroll
    for (int at = 0; at < A_T; ++at) {
      const int s = wm * A_T + at;
      long pix = -1;
      if (TAPS == 9 && FLAT) {
        const int f = f0 + s * 16 + li, y = f / Wp, xq = f - y * Wp;
        if (y < a.H && xq >= 1 && xq <= a.W) pix = ((long)img * a.H + y) * a.W + xq - 1;
      } else if (TAPS == 9) {
        const int y = y0 + s, x = x0 + li;
        if (y < a.H && x < a.W) pix = ((long)img * a.H + y) * a.W + x;
      } else {
        const long m = m0 + s * 16 + li;
        if (m < m_lim) pix = m;
      }
      if (pix < 0) continue;
      const f32x4 v = acc[at][ct] + bv;
      const h4 hv = __builtin_convertvector(v, h4);
      if (v4 && nb + 3 < a.N) {
        *reinterpret_cast<h4*>(Cg + pix * a.ldc + nb) = hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = (float)hv[r]; t1[ct][r] += x; t2[ct][r] += x * x; }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < a.N) {
            Cg[pix * a.ldc + nb + r] = hv[r];
            const float x = (float)hv[r]; t1[ct][r] += x; t2[ct][r] += x * x;
          }
      }
    }
  }
  if (a.stat_sum) {
    float* red = reinterpret_cast<float*>(smem_u);   // [2][WM][BN]
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v1 = t1[ct][r], v2 = t2[ct][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
        if (li == 0) {
          const int nl = (wn * C_T + ct) * 16 + 4 * g + r;
          red[(0 * WM + wm) * BN + nl] = v1;
          red[(1 * WM + wm) * BN + nl] = v2;
        }
      }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += 256) {
      const int n = n0 + nl;
      if (n < a.N) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { v1 += red[(0 * WM + w) * BN + nl]; v2 += red[(1 * WM + w) * BN + nl]; }
        a.stat_sum[(long)n * a.n_mblocks + mblk] = v1;
        a.stat_sq[(long)n * a.n_mblocks + mblk] = v2;
      }
    }
  }
}

template <int TAPS, int BM, int BN, int WM, int WN, int DEPTH, bool FLAT>
static int launch_hconv(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = HGeom<TAPS, BM, BN, WM, WN, DEPTH, FLAT>;
  int mblocks;
  if (TAPS == 9 && FLAT) mblocks = a.NB * ((a.H * (a.W + 2) + BM - 1) / BM);
  else if (TAPS == 9) mblocks = a.NB * ((a.H + G::TH - 1) / G::TH) * ((a.W + 15) / 16);
  else if (a.stat_groups > 1) mblocks = a.stat_groups * (int)((a.M / a.stat_groups + BM - 1) / BM);
  else mblocks = (int)((a.M + BM - 1) / BM);
  if (q) {
    q[0] = mblocks;
    q[1] = TAPS * 1000000 + BM * 1000 + BN + (FLAT ? 500000 : 0);
    q[2] = G::KC * 100 + DEPTH * 10 + 1;
    return ARCO_OK;
  }
  size_t sh = (size_t)2 * G::BUF * 4;
  const size_t red = (size_t)2 * WM * BN * sizeof(float);
  if (sh < red) sh = red;
  auto kern = hconv_kernel<TAPS, BM, BN, WM, WN, DEPTH, FLAT>;
  static unsigned long long attr_set = 0;
  if (sh > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); }
  IgemmArgs b = a;
  b.n_mblocks = mblocks;
  b.n_nblocks = (a.Npad + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks * b.n_nblocks)), dim3(256), sh, st, b);
  return arco_launch_status();
}

static long h_rect_blocks(const IgemmArgs& a, int bm, int bn) {
  return (long)a.NB * ((a.H + bm / 16 - 1) / (bm / 16)) * ((a.W + 15) / 16) * ((a.Npad + bn - 1) / bn);
}
static long h_flat_blocks(const IgemmArgs& a, int bm, int bn) {
  return (long)a.NB * ((a.H * (a.W + 2) + bm - 1) / bm) * ((a.Npad + bn - 1) / bn);
}
template <bool FLAT>
static int hconv_dispatch3(const IgemmArgs& a, hipStream_t st, int* q) {
  static const long want = getenv("ARCO_HCONV_WANT") ? atol(getenv("ARCO_HCONV_WANT")) : 256;      // (the deep levels run better on the larger tiles even with one workgroup per CU: 128 ch 39.8 -> 37.6 us)
  auto blocks = [&](int bm, int bn) { return FLAT ? h_flat_blocks(a, bm, bn) : h_rect_blocks(a, bm, bn); };
  if (a.Npad <= 16) return launch_hconv<9, 128, 16, 4, 1, 3, FLAT>(a, st, q);
  if (a.Npad <= 32) {
    static const int big = getenv("ARCO_HCONV_256") ? atoi(getenv("ARCO_HCONV_256")) : 1;      // 256-pixel tiles (4 x 2 MFMA tiles per wave: 6 fragment reads per 8 MFMAs instead of 4 per 4): 32 -> 32 @80x80x48 x2 63 -> 54 us; 0 = off
    if (big && !FLAT && blocks(256, 32) >= want) return launch_hconv<9, 256, 32, 4, 1, 3, false>(a, st, q);
    if (blocks(128, 32) >= want) return launch_hconv<9, 128, 32, 4, 1, 3, FLAT>(a, st, q);
    return launch_hconv<9, 64, 32, 2, 2, 3, FLAT>(a, st, q);
  }
  if (blocks(128, 64) >= want) return launch_hconv<9, 128, 64, 4, 1, 3, FLAT>(a, st, q);
  if (blocks(64, 64) >= want) return launch_hconv<9, 64, 64, 2, 2, 3, FLAT>(a, st, q);
  if (blocks(64, 32) >= want) return launch_hconv<9, 64, 32, 2, 2, 3, FLAT>(a, st, q);
  return launch_hconv<9, 32, 32, 2, 2, 3, FLAT>(a, st, q);
}


// ---------------------------------------------------------------------------------------------------------------------
// Resident-weights form for the shallow levels (3x3x3, K = N = 16 on 8 x 16 tiles / 256 threads, K = N = 32 on 16 x 16 tiles / 512
// threads; planes with H % TH == 0 and W % 16 == 0: the V-Net's 160x160x96 / 80x80x48 levels).  hconv_kernel re-stages a tile's weights (27 taps: as many bytes as the activations at these
// widths) and reads every input plane three times; here a persistent workgroup
//   * copies ALL 27 taps' weights to LDS once per launch,
//   * owns units of (volume, TH x 16 tile, S consecutive planes) and walks the depth axis with a RING of three input plane
//     tiles in LDS: producing output plane p needs only plane p + 1 as new input (loaded to registers during plane p - 1's
//     MFMAs) - every input plane tile is read once per unit instead of three times, and never again for the weights,
//   * pairs the 27 taps freely across the three resident planes: 14 MFMA steps per 16 channels instead of 15,
// with the same MFMA / fragment layout as hconv_kernel (rows of 16 channels = 32 bytes, no padding: conflict-free).
// ---------------------------------------------------------------------------------------------------------------------
template <int NC, int C_T, int TH>
struct HRwGeom {
  static constexpr int HR = (TH + 2) * 18;                // halo rows of a TH x 16 tile
  static constexpr int NT = 32 * TH, NW = TH / 2;         // threads / waves: a wave owns two tile rows
  static constexpr int WROWS = 27 * C_T * 16 + 1;         // weight rows per 16-k chunk (+ the zero row)
  static constexpr int W_DW = NC * WROWS * 8, A_DW = NC * HR * 8;
  static constexpr int NP = (HR * NC * 2 + NT - 1) / NT;  // 16-byte activation pieces per thread and plane
  static constexpr size_t LDS_BYTES = (size_t)(W_DW + 3 * A_DW + 2 * NW * C_T * 16) * 4;
};

template <int NC, int C_T, int TH>
__global__ __launch_bounds__(32 * TH) void hconv_rw_kernel(IgemmArgs a, int S, int nseg) {
  using G = HRwGeom<NC, C_T, TH>;
  constexpr int HR = G::HR, WROWS = G::WROWS, NP = G::NP, NT = G::NT, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_r[];
  unsigned* const Ws = smem_r;                        // [NC][WROWS][8]
  unsigned* const As = smem_r + G::W_DW;              // [3][NC][HR][8]
  float* const red = reinterpret_cast<float*>(smem_r + G::W_DW + 3 * G::A_DW);     // [2][NW][C_T * 16]
  const _Float16* const Ag = reinterpret_cast<const _Float16*>(a.A);
  const _Float16* const Wg = reinterpret_cast<const _Float16*>(a.Wp);
  _Float16* const Cg = reinterpret_cast<_Float16*>(a.C);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4, h = g >> 1;
  const int tiles_x = a.W >> 4, tiles_y = a.H / TH;
  const int units = (a.NB / a.D3) * tiles_y * tiles_x * nseg;

  // weights: global pack [tap][Npad][Kpad = 32] halves -> LDS [chunk][tap * N + n][16 halves]
  for (int i = tid; i < NC * (WROWS - 1) * 2; i += NT) {
    const int half = i & 1, r = (i >> 1) % (WROWS - 1), c = (i >> 1) / (WROWS - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(Wg + (long)r * a.Kpad + c * 16 + half * 8);      // r = tap * Npad + n, Npad == N
    *reinterpret_cast<u32x4_ma*>(Ws + (c * WROWS + r) * 8 + half * 4) = v;
  }
  for (int i = tid; i < NC * 8; i += NT) Ws[((i >> 3) * WROWS + WROWS - 1) * 8 + (i & 7)] = 0u;

  // piece geometry (plane-invariant): halo row, chunk, half
  int prow[NP], pc[NP], ph[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int idx = tid + i * NT;
    const int qq = idx % (2 * NC);
    prow[i] = idx < HR * NC * 2 ? idx / (2 * NC) : -1; pc[i] = qq >> 1; ph[i] = qq & 1;
  }
  f32x4 bv[C_T];
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[ct][r] = a.bias ? a.bias[ct * 16 + 4 * g + r] : 0.f;

  u32x4 R[NP];
  // XCD-aware unit order: workgroups are dealt round-robin over the 8 XCDs (private L2s); XCD x takes the x-th CONTIGUOUS eighth of
  // the units, so the units that share halo rows / planes (neighbouring tiles and depth segments) meet in one L2
  const bool xmap = (gridDim.x & 7) == 0;
  const int G8 = xmap ? (int)gridDim.x >> 3 : (int)gridDim.x, U8 = xmap ? (units + 7) >> 3 : units;
  const int u_lo = xmap ? ((int)blockIdx.x & 7) * U8 : 0, u_hi = xmap ? min(units, u_lo + U8) : units;
  for (int unit = u_lo + (xmap ? (int)blockIdx.x >> 3 : (int)blockIdx.x); unit < u_hi; unit += G8) {
    const int seg = unit % nseg; int col = unit / nseg;
    const int tx = col % tiles_x; col /= tiles_x;
    const int ty = col % tiles_y; const int v = col / tiles_y;
    const int y0 = ty * TH, x0 = tx * 16, p0 = seg * S, p1 = min(a.D3, p0 + S);
    auto load_plane = [&](int p) {
      const bool pok = p >= 0 && p < a.D3;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        u32x4 val = u32x4{0, 0, 0, 0};
        if (prow[i] >= 0 && pok) {
          const int hy = prow[i] / 18, hx = prow[i] - hy * 18;
          const int y = y0 + hy - 1, x = x0 + hx - 1;
          if (y >= 0 && y < a.H && x >= 0 && x < a.W)
            val = *reinterpret_cast<const u32x4*>(Ag + ((((long)v * a.D3 + p) * a.H + y) * a.W + x) * a.lda + pc[i] * 16 + ph[i] * 8);
        }
        R[i] = val;
      }
    };
    auto store_plane = [&](int slot) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (prow[i] >= 0) *reinterpret_cast<u32x4_ma*>(As + slot * G::A_DW + (pc[i] * HR + prow[i]) * 8 + ph[i] * 4) = R[i];
    };
    float t1[C_T][4], t2[C_T][4];
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { t1[ct][r] = 0.f; t2[ct][r] = 0.f; }

    int sA = 0, sB = 1, sC = 2;                 // ring slots of planes p - 1, p, p + 1
    load_plane(p0 - 1); store_plane(sA);
    load_plane(p0); store_plane(sB);
    load_plane(p0 + 1);
    for (int p = p0; p < p1; ++p) {
      store_plane(sC);
      __syncthreads();
      if (p + 1 < p1) load_plane(p + 2);        // in flight during this plane's MFMAs
      f32x4 acc[2][C_T];
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) acc[at][ct] = f32x4{0, 0, 0, 0};
      const int sb0 = sA * G::A_DW, sb1 = sB * G::A_DW, sb2 = sC * G::A_DW;
      // fragment reads of step st + 1 are issued in front of step st's MFMAs (two register sets; left to itself the compiler reads a
      // step's fragments, waits for them and only then issues its MFMAs - an LDS round trip per step on the critical path)
      h8 fa[2][2], fb[2][C_T];
      auto read_step = [&](int st, int set) {
        const int c = st / 14, sp_ = st % 14;
        // lanes g = 0,1 take tap 2s, g = 2,3 tap 2s + 1 (tap 27: the zero weight row on tap 26's activations)
        const int tp0 = 2 * sp_, tp1 = 2 * sp_ + 1 > 26 ? 26 : 2 * sp_ + 1;
        const int dz0 = tp0 / 9, dz1 = tp1 / 9;
        const int o0 = ((tp0 % 9) / 3) * 18 + tp0 % 3, o1 = ((tp1 % 9) / 3) * 18 + tp1 % 3;
        const int base0 = (dz0 == 0 ? sb0 : (dz0 == 1 ? sb1 : sb2)) + o0 * 8;
        const int base1 = (dz1 == 0 ? sb0 : (dz1 == 1 ? sb1 : sb2)) + o1 * 8;
        const int abase = (h ? base1 : base0) + c * HR * 8 + (g & 1) * 4;
        const int wrow = h ? (2 * sp_ + 1 > 26 ? WROWS - 1 : tp1 * C_T * 16 + li) : tp0 * C_T * 16 + li;
        const int wstep = (h && 2 * sp_ + 1 > 26) ? 0 : 16;         // the zero row serves every channel group
#pragma unroll
        for (int at = 0; at < 2; ++at) fa[set][at] = lds_h8(As + abase + ((2 * wid + at) * 18 + li) * 8);
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) fb[set][ct] = lds_h8(Ws + (c * WROWS + wrow + ct * wstep) * 8 + (g & 1) * 4);
      };
      read_step(0, 0);
#pragma unroll
      for (int st = 0; st < NC * 14; ++st) {
        if (st + 1 < NC * 14) read_step(st + 1, (st + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
          for (int at = 0; at < 2; ++at)
            acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[st & 1][ct], fa[st & 1][at], acc[at][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // epilogue of the plane: bias, f16, 8-byte stores, statistics of the rounded values
#pragma unroll
      for (int at = 0; at < 2; ++at) {
        const long pix = ((((long)v * a.D3 + p) * a.H + y0 + 2 * wid + at) * a.W + x0 + li);
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) {
          const h4 hv = __builtin_convertvector(acc[at][ct] + bv[ct], h4);
          *reinterpret_cast<h4*>(Cg + pix * a.ldc + ct * 16 + 4 * g) = hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float x = (float)hv[r]; t1[ct][r] += x; t2[ct][r] += x * x; }
        }
      }
      __syncthreads();                          // slot sA (plane p - 1) is free from here on
      const int t = sA; sA = sB; sB = sC; sC = t;
    }
    if (a.stat_sum) {       // the unit's sums go to the slab of its first plane tile, zeros to the slabs of its other planes
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v1 = t1[ct][r], v2 = t2[ct][r];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
          if (li == 0) { red[(0 * NW + wid) * C_T * 16 + ct * 16 + 4 * g + r] = v1; red[(1 * NW + wid) * C_T * 16 + ct * 16 + 4 * g + r] = v2; }
        }
      __syncthreads();
      if (tid < C_T * 16) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { v1 += red[(0 * NW + w) * C_T * 16 + tid]; v2 += red[(1 * NW + w) * C_T * 16 + tid]; }
        for (int p = p0; p < p1; ++p) {
          const long slab = (((long)v * a.D3 + p) * tiles_y + ty) * tiles_x + tx;
          a.stat_sum[(long)tid * a.n_mblocks + slab] = p == p0 ? v1 : 0.f;
          a.stat_sq[(long)tid * a.n_mblocks + slab] = p == p0 ? v2 : 0.f;
        }
      }
      __syncthreads();
    }
  }
}

static int hconv_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
static bool hconv_rw_eligible(const IgemmArgs& a) {
  static const int mode = getenv("ARCO_HCONV_RW") ? atoi(getenv("ARCO_HCONV_RW")) : 1;      // A/B switch: bit 0: 16 channels (default), bit 1: 32 (16 x 16 tiles, 512 threads: measured level with hconv_kernel, 502 vs 521 TFLOP/s in the step - off)
  const bool on = a.K == 16 ? (mode & 1) : (mode & 2);
  const int th = a.K == 16 ? ((mode & 4) && a.H % 16 == 0 ? 16 : 8) : 16;
  return on && a.K == a.N && (a.K == 16 || a.K == 32) && a.Npad == a.N && a.H % th == 0 && (a.W & 15) == 0 && (a.lda & 7) == 0 &&
         (a.ldc & 3) == 0 && a.R == nullptr && a.Kpad == 32;
}
// BN slabs: one per plane tile, NB * (H / TH) * (W / 16) (a function of the plane count alone, like the other kernels' counts)
template <int NC, int C_T, int TH>
static int launch_hconv_rw(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = HRwGeom<NC, C_T, TH>;
  const long tiles = (long)(a.H / TH) * (a.W >> 4);
  if (q) { q[0] = (int)(a.NB * tiles); q[1] = 9 * 1000000 + 400000 + TH * 1000 + C_T * 16; q[2] = 16 * 100 + 30 + 1; return ARCO_OK; }
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  const long cols = (long)(a.NB / a.D3) * tiles;
  int per_cu = (int)(160 * 1024 / G::LDS_BYTES); if (per_cu > 4) per_cu = 4; if (per_cu < 1) per_cu = 1;
  const long slots = (long)hconv_cus() * per_cu;
  // planes per unit: long units amortise the two halo planes; short ones fill the chip when there are few tile columns
  int S = 8;
  while (S > 2 && cols * ((a.D3 + S - 1) / S) < 3 * slots) S >>= 1;
  if (S > a.D3) S = a.D3;
  const int nseg = (a.D3 + S - 1) / S;
  const long units = cols * nseg;
  auto kern = hconv_rw_kernel<NC, C_T, TH>;
  static unsigned long long attr_set = 0;
  if (G::LDS_BYTES > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
  IgemmArgs b = a;
  b.n_mblocks = (int)(a.NB * tiles);
  const long grid = slots < units ? slots : units;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, st, b, S, nseg);
  return arco_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// hconv_fc_kernel<A_T, C_T>: the f16-storage counterpart of conv3d_fc_kernel (conv3d_fl.hip; round 6) for the 3x3x3 levels
// below full resolution - the 3x3x3 convolution as a 3x3 over 3 K virtual channels on flat tiles of 64 A_T positions, persistent
// workgroups of 4 MFMA + 4 loader waves, one rendezvous per 16-channel chunk.  With f16 storage a product is ONE
// v_mfma_f32_16x16x32_f16 (six in the split-bf16 mode), so a chunk is 5 A_T C_T MFMAs per wave: the kernel lives or dies by what
// surrounds them.  hconv_kernel staged every chunk through registers between two __syncthreads() (phase-serialised: 0.14 of the f16
// peak).  Here the loader waves only ISSUE: activation tiles and weights both go HBM -> LDS by LDS-DMA (an f16 tile needs no
// conversion; positions that are padding, and planes outside the volume, are fetched from a zero row), into rings of four chunk
// buffers filled two chunks ahead behind counted s_waitcnt vmcnt(N).  Same products in the same order as hconv_kernel: bit-identical.
// LDS: [4][A rows][8] A + [4][10][BN][8] weights + bias (A_T = 4, C_T = 4: 49.2 + 81.9 KB).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(64))) unsigned int hconv_fc_zero_row[16];

template <int A_T, int C_T>
struct HFcGeom {
  static constexpr int BM = 64 * A_T, WPMAX = 63, AROWS = BM + 2 * WPMAX + 2, BN = C_T * 16, NBUF = 4;
  static constexpr int NA = (AROWS * 2 + 255) / 256;            // LDS-DMA instructions per thread and chunk: activation pieces (2 per row)
  static constexpr int NW = (10 * BN * 2 + 255) / 256;          // ... weight pieces
  static constexpr int WBUF_DW = NW * 256 * 4;
  static constexpr int BIAS_DW = 256;
  static constexpr int A_DW = NA * 256 * 4;                     // whole wave-instructions (every loader wave issues all NA of them)
  static constexpr size_t LDS_BYTES = (size_t)(NBUF * (A_DW + WBUF_DW) + BIAS_DW) * 4;
};

__device__ __forceinline__ void mfma_acc_h(f32x4& c, const h8& x, const h8& y) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
}

template <int A_T, int C_T>
__global__ __launch_bounds__(512) void hconv_fc_kernel(IgemmArgs a) {
  using G = HFcGeom<A_T, C_T>;
  constexpr int BM = G::BM, BN = G::BN, NA = G::NA, NW = G::NW, NBUF = G::NBUF;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_f[];
  const int Wp = a.W + 2, npos = a.H * Wp, per_plane = (npos + BM - 1) / BM;
  const int arows = BM + 2 * Wp + 2;
  constexpr int A_DW = G::A_DW;
  unsigned* const As = smem_f;                                  // [NBUF][A_DW]: rows of 16 channels = 8 dwords
  unsigned* const Ws = As + NBUF * A_DW;                        // [NBUF][5 steps][tapL][BN][8]
  float* const bias_s = reinterpret_cast<float*>(Ws + NBUF * G::WBUF_DW);
  const _Float16* const Ag = reinterpret_cast<const _Float16*>(a.A);
  const _Float16* const Wg = reinterpret_cast<const _Float16*>(a.Wp);
  _Float16* const Cg = reinterpret_cast<_Float16*>(a.C);
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const bool producer = threadIdx.x >= 256;
  const int nk = a.K >> 4, nvc = 3 * nk;
  const int total_tiles = a.n_mblocks * a.n_nblocks;
  const bool xcd_map = (gridDim.x & 7) == 0;
  const int G8 = xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int T8 = xcd_map ? (total_tiles + 7) >> 3 : total_tiles;
  const int tile0 = xcd_map ? ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int tile_end_x = xcd_map ? min(total_tiles, (((int)blockIdx.x & 7) + 1) * T8) : total_tiles;
  const int my_tiles = tile0 < tile_end_x ? (tile_end_x - tile0 + G8 - 1) / G8 : 0;
  const int total_gc = my_tiles * nvc;
  if (my_tiles == 0) return;
  const bool has_stats = a.stat_sum != nullptr;
  struct Desc { int j, vc, dz, kc, img, f0, nblk, mblk, pl; };
  auto decode = [&](Desc& d) {
    const int v = tile0 + d.j * G8;
    d.mblk = v / a.n_nblocks; d.nblk = v - d.mblk * a.n_nblocks;
    d.img = d.mblk / per_plane;
    d.f0 = (d.mblk - d.img * per_plane) * BM;
    d.pl = d.img % a.D3;
  };
  auto advance = [&](Desc& d) {
    ++d.vc;
    if (++d.kc == nk) { d.kc = 0; ++d.dz; }
    if (d.vc == nvc) { d.vc = 0; d.dz = 0; d.kc = 0; ++d.j; decode(d); }
  };
  Desc d0{0, 0, 0, 0, 0, 0, 0, 0, 0};
  decode(d0);

  if (producer) {
    // ================================================================ loader waves: LDS-DMA only
    const float inv_wp = 1.0f / (float)Wp;
    const long plane_el = (long)a.H * a.W * a.lda;       // f16 elements per plane
    const _Float16* const zrow = reinterpret_cast<const _Float16*>(hconv_fc_zero_row);
    // activation piece i of this thread: staged row (tid + i * 256) >> 1, half = channel octet; its source offset (elements within the
    // plane) and whether it is a pixel belong to the TILE (tile_geom at a tile's first chunk)
    int aoffs[NA]; unsigned okm_t = 0;
    auto tile_geom = [&](const Desc& d) {
      okm_t = 0;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int idx = tid + it * 256, row = idx >> 1, half = idx & 1;
        const int pidx = d.f0 + row - 1;
        const int py = (int)(((float)pidx + 0.5f) * inv_wp), px = pidx - py * Wp;
        const bool ok = row < arows && pidx >= 0 && py >= 1 && py <= a.H && px >= 1 && px <= a.W;
        aoffs[it] = ok ? ((py - 1) * a.W + px - 1) * (int)a.lda + half * 8 : 0;
        okm_t |= ok ? (1u << it) : 0u;
      }
    };
    int woff[NW]; bool wz[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int p = (i * 4 + wid) * 64 + lane;
      const int half = p & 1, n = (p >> 1) % BN, t10 = p / (2 * BN);      // t10 = 2 s + tapL
      wz[i] = t10 >= 9;
      woff[i] = t10 < 9 ? (t10 * a.Npad + n) * a.Kpad + half * 8 : 0;
    }
    const long wslice = (long)9 * a.Npad * a.Kpad;
    auto load_chunk = [&](const Desc& d, bool real, int buf) {      // NW + NA LDS-DMA instructions per wave, real or not (exact vmcnt counts)
      const _Float16* wb = Wg + d.dz * wslice + (long)d.nblk * BN * a.Kpad + d.kc * 16;
      unsigned* const wdst = Ws + buf * G::WBUF_DW;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const _Float16* src = wz[i] ? zrow + (lane & 1) * 8 : wb + woff[i];
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(wdst + (i * 4 + wid) * 256), 16, 0, 0);
      }
      if (d.vc == 0) tile_geom(d);
      const int pz = d.pl + d.dz - 1;
      const bool pok = real && pz >= 0 && pz < a.D3;
      const _Float16* ab = Ag + (long)(pok ? d.img + d.dz - 1 : 0) * plane_el + d.kc * 16;
      unsigned* const adst = As + buf * A_DW;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const _Float16* src = (pok && ((okm_t >> it) & 1u)) ? ab + aoffs[it] : zrow + (lane & 1) * 8;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(adst + (it * 4 + wid) * 256), 16, 0, 0);
      }
    };
    constexpr int NC = NW + NA;
    for (int i = tid; i < G::BIAS_DW; i += 256) bias_s[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    Desc dl = d0;                                   // the chunk whose DMA is issued next
    load_chunk(dl, true, 0); advance(dl);
    load_chunk(dl, total_gc > 1, 1); advance(dl);
    load_chunk(dl, total_gc > 2, 2); advance(dl);
    wait_vm<2 * NC>();                              // chunk 0 has landed
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();                   // B0
    // barrier k (the consumers pass it at the head of step 4 of chunk k): chunk k + 1 is complete in LDS; behind it the DMA of
    // chunk k + 3 goes into the buffers chunk k - 1 used
    for (int k = 0; k < total_gc; ++k) {
      wait_vm<NC>();                                // chunk k + 1 (only chunk k + 2's DMA is younger)
      __builtin_amdgcn_s_barrier();
      load_chunk(dl, k + 3 < total_gc, (k + 3) & 3); advance(dl);
    }
    wait_vm<0>();
    return;
  }

  // ================================================================== MFMA waves
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s_ = 0; s_ < 5; ++s_) {
    const int tap = 2 * s_ + tl > 8 ? 8 : 2 * s_ + tl;
    aoff[s_] = ((tap / 3) * Wp + tap % 3) * 8;
  }
  const int laneA = (wid * A_T * 16 + li) * 8 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 8 + (g & 1) * 4;              // within a step's [tapL][n] block
  h8 fa[A_T], fb[2][C_T];
  f32x4 acc[A_T][C_T];
#pragma unroll
  for (int i = 0; i < A_T; ++i)
#pragma unroll
    for (int j = 0; j < C_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();        // B0
#pragma unroll
  for (int at = 0; at < A_T; ++at) fa[at] = lds_h8(As + laneA + aoff[0] + at * 16 * 8);
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct) fb[0][ct] = lds_h8(Ws + laneB + ct * 16 * 8);

  auto chunk = [&](auto CP_, int gc, bool more) {
    constexpr int CP = decltype(CP_)::value;               // B set of step 0 (5 steps per chunk: alternates per chunk)
    const unsigned* Acur = As + (gc & 3) * A_DW;
    const unsigned* Anxt = As + ((gc + 1) & 3) * A_DW;
    const unsigned* Wc = Ws + (gc & 3) * G::WBUF_DW;
    const unsigned* Wn = Ws + ((gc + 1) & 3) * G::WBUF_DW;
    auto step = [&](auto S_) {
      constexpr int S = decltype(S_)::value;
      constexpr int P = (CP + S) & 1, Q = P ^ 1;
      if (S == 4) {                    // the one rendezvous of the chunk: the next chunk is complete in LDS
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
      }
      constexpr int BPA = (C_T + A_T - 1) / A_T;            // next-step B reads per pixel tile
      const unsigned* An = S < 4 ? Acur : Anxt;
      const unsigned* Wb = (S < 4 ? Wc : Wn) + ((S + 1) % 5) * (2 * BN * 8);
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) mfma_acc_h(acc[at][ct], fb[P][ct], fa[at]);      // D = W . X^T
        if (S < 4 || more) fa[at] = lds_h8(An + laneA + aoff[(S + 1) % 5] + at * 16 * 8);
#pragma unroll
        for (int k = at * BPA; k < (at + 1) * BPA && k < C_T; ++k) fb[Q][k] = lds_h8(Wb + laneB + k * 16 * 8);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
  };

  const float inv_wp = 1.0f / (float)Wp;
  auto tile_end = [&]() {              // bias, round to f16, 8-byte stores, BN partial statistics of the rounded values (one slab per wave)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int n0 = d0.nblk * BN;
    long pix[A_T];
#pragma unroll
    for (int at = 0; at < A_T; ++at) {
      const int f = d0.f0 + (wid * A_T + at) * 16 + li;
      const int y = (int)(((float)f + 0.5f) * inv_wp), xq = f - y * Wp;
      pix[at] = (y < a.H && xq >= 1 && xq <= a.W) ? ((long)d0.img * a.H + y) * a.W + xq - 1 : -1;
    }
    float s1[C_T][4], s2[C_T][4];
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct) {
      const int n = n0 + ct * 16 + 4 * g;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; }
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
        if (pix[at] >= 0) {
          const f32x4 v = acc[at][ct] + bv;
          const h4 hv = __builtin_convertvector(v, h4);
          *reinterpret_cast<h4*>(Cg + pix[at] * a.ldc + n) = hv;
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float x = (float)hv[r]; s1[ct][r] += x; s2[ct][r] += x * x; }
        }
        acc[at][ct] = f32x4{0, 0, 0, 0};
      }
    }
    if (has_stats) {
      const long slab = (long)d0.mblk * 4 + wid, nslab = (long)a.n_mblocks * 4;
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v1 = row16_sum(s1[ct][r]), v2 = row16_sum(s2[ct][r]);
          if (li == 0) {
            a.stat_sum[(n0 + ct * 16 + 4 * g + r) * nslab + slab] = v1;
            a.stat_sq[(n0 + ct * 16 + 4 * g + r) * nslab + slab] = v2;
          }
        }
    }
  };
  for (int gc = 0; gc < total_gc; gc += 2) {
    chunk(std::integral_constant<int, 0>{}, gc, gc + 1 < total_gc);
    if (d0.vc + 1 == nvc) tile_end();
    advance(d0);
    if (gc + 1 < total_gc) {
      chunk(std::integral_constant<int, 1>{}, gc + 1, gc + 2 < total_gc);
      if (d0.vc + 1 == nvc) tile_end();
      advance(d0);
    }
  }
}

template <int A_T, int C_T>
static int launch_hfc(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = HFcGeom<A_T, C_T>;
  const int mblocks = a.NB * ((a.H * (a.W + 2) + G::BM - 1) / G::BM);
  if (q) { q[0] = 4 * mblocks; q[1] = 9260000 + A_T * 1000 + G::BN; q[2] = 1631; return ARCO_OK; }
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  const size_t lds = G::LDS_BYTES;
  static_assert(G::LDS_BYTES <= 160 * 1024, "LDS");
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = a.Npad / G::BN;
  const int total = mblocks * b.n_nblocks, cus = conv_sp_cus();
  auto kern = hconv_fc_kernel<A_T, C_T>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), lds, st, b);
  return arco_launch_status();
}

// -1: shape not taken.  A/B: ARCO_HCONV_FC=0 off; ARCO_HCONV_FC_CFG=<A_T><C_T> forces a tile shape.  arco_conv3d_fl_set(0) switches it off too.
extern "C" int arco_conv3d_fl_set(int on);
static int hconv_fc_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  static const int on = getenv("ARCO_HCONV_FC") ? atoi(getenv("ARCO_HCONV_FC")) : 1;
  static const int forced = getenv("ARCO_HCONV_FC_CFG") ? atoi(getenv("ARCO_HCONV_FC_CFG")) : 0;
  if (!on) return -1;
  { const int cur = arco_conv3d_fl_set(1); if (!cur) { arco_conv3d_fl_set(0); return -1; } }
  if ((a.K & 15) != 0 || a.K < 16 || (a.N & 31) != 0 || a.N != a.Npad || a.N > 256 || a.W + 2 > 63 || (a.lda & 7) != 0 || (a.ldc & 3) != 0) return -1;
  if ((long)a.H * a.W * a.lda >= (1l << 30)) return -1;
  // the 32-channel level on planes whose width is a multiple of 16 stays on hconv_kernel<9,256,32> (rectangular 16 x 16 tiles,
  // measured 53 against 63 us at 80 x 80 x 48 x 2 volumes)
  if (!forced && (a.W & 15) == 0 && a.Npad <= 32) return -1;
  int best = forced;
  if (!best) {
    // Launch model (fit of tools/micro/fl_bench.py HALF=1, profiles/r06_notes.md section 9): with one MFMA per product these launches
    // are bound by the L2 -> LDS stream (every tile re-reads the 27 taps' weights: 20 KB of weights against 3-10 KB of activations per
    // chunk), ~5.9 TB/s over the chip: time = LDS-DMA bytes of all rounds / 5.9 TB/s (a partly filled last round counts as a full
    // one), or the MFMA / rendezvous path if that is longer, + 1.9 us of epilogue per round
    double bc = 1e300;
    // (A_T = 5 / 6 with 64-wide blocks, round 6: a chunk's 20 KB of weights serve 320 / 384 positions - the stream per position drops
    //  from 148 to 90 B at 40 x 24 planes; ARCO_HCONV_FC_BIG=0 leaves them out)
    static const int big = getenv("ARCO_HCONV_FC_BIG") ? atoi(getenv("ARCO_HCONV_FC_BIG")) : 1;
    const int cand[10] = {64, 54, 44, 34, 24, 14, 42, 32, 22, 12};      // (<6,2> / <5,2> exist for ARCO_HCONV_FC_CFG only: 58-69 us against hconv_kernel's 53 at 80 x 80 x 48)
    const int cus = conv_sp_cus();
    for (int i = big ? 0 : 2; i < 10; ++i) {
      const int a_t = cand[i] / 10, c_t = cand[i] % 10;
      if ((a.N % (16 * c_t)) != 0) continue;
      const long tiles = (long)a.NB * ((a.H * (a.W + 2) + 64 * a_t - 1) / (64 * a_t)) * (a.Npad / (16 * c_t));
      const long rounds = (tiles + cus - 1) / cus;
      const double chunks = 3.0 * (a.K >> 4);
      const double bytes = (64.0 * a_t + 2.0 * (a.W + 2) + 2.0) * 32.0 + 10.0 * 16 * c_t * 32.0;
      const double eff_tiles = tiles <= cus ? (double)tiles : (double)rounds * cus;
      const double t_bw = eff_tiles * chunks * bytes / 5.9e6;
      const double t_mfma = (double)rounds * chunks * (5.0 * a_t * c_t * 0.008 + 0.10);
      const double c = (t_bw > t_mfma ? t_bw : t_mfma) + 1.9 * rounds;
      if (c < bc * 0.999) { bc = c; best = cand[i]; }
    }
  }
  switch (best) {
    case 64: if ((a.N & 63) == 0) return launch_hfc<6, 4>(a, st, q); break;
    case 54: if ((a.N & 63) == 0) return launch_hfc<5, 4>(a, st, q); break;
    case 44: if ((a.N & 63) == 0) return launch_hfc<4, 4>(a, st, q); break;
    case 34: if ((a.N & 63) == 0) return launch_hfc<3, 4>(a, st, q); break;
    case 24: if ((a.N & 63) == 0) return launch_hfc<2, 4>(a, st, q); break;
    case 14: if ((a.N & 63) == 0) return launch_hfc<1, 4>(a, st, q); break;
    case 62: return launch_hfc<6, 2>(a, st, q);
    case 52: return launch_hfc<5, 2>(a, st, q);
    case 42: return launch_hfc<4, 2>(a, st, q);
    case 32: return launch_hfc<3, 2>(a, st, q);
    case 22: return launch_hfc<2, 2>(a, st, q);
    case 12: return launch_hfc<1, 2>(a, st, q);
  }
  return -1;
}

// Streaming narrow 1x1x1 convolutions on f16 storage (see conv1x1_narrow_out_kernel in igemm.hip): the V-Net's out_conv and its data
// gradient over the full-resolution f16 map (157 MB at the LiTS size) ran 115 us on 256 x 16 GEMM tiles; fp32 FMAs of the exact
// f16 x f16 products, one rounding to f16 at the store - as the MFMA route (fp32 accumulation), in a different summation order.
template <int Q>
__global__ __launch_bounds__(256) void hconv1x1_narrow_out_kernel(IgemmArgs a) {
  const _Float16* const Ag = reinterpret_cast<const _Float16*>(a.A);
  const _Float16* const Wg = reinterpret_cast<const _Float16*>(a.Wp);
  _Float16* const Cg = reinterpret_cast<_Float16*>(a.C);
  const int lq = threadIdx.x % Q;
  float w[4][8], bs[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    bs[n] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = 8 * lq + e; w[n][e] = (n < a.N && k < a.K) ? (float)Wg[(long)n * a.Kpad + k] : 0.f; }
  }
  const long rstride = (long)gridDim.x * (256 / Q);
  for (long row = (long)blockIdx.x * (256 / Q) + threadIdx.x / Q; row < a.M; row += rstride) {
    const h8 x = *reinterpret_cast<const h8*>(Ag + row * a.lda + 8 * lq);
    float p[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float acc = (float)x[0] * w[n][0];
#pragma unroll
      for (int e = 1; e < 8; ++e) acc = fmaf((float)x[e], w[n][e], acc);
      p[n] = acc;
    }
#pragma unroll
    for (int off = 1; off < Q; off <<= 1)
#pragma unroll
      for (int n = 0; n < 4; ++n) p[n] += __shfl_xor(p[n], off);
    if (lq == 0) {
      _Float16* o = Cg + row * a.ldc;
#pragma unroll
      for (int n = 0; n < 4; ++n) if (n < a.N) o[n] = (_Float16)(p[n] + bs[n]);
    }
  }
}
__global__ __launch_bounds__(256) void hconv1x1_narrow_in_kernel(IgemmArgs a) {
  const _Float16* const Ag = reinterpret_cast<const _Float16*>(a.A);
  const _Float16* const Wg = reinterpret_cast<const _Float16*>(a.Wp);
  _Float16* const Cg = reinterpret_cast<_Float16*>(a.C);
  const int Q = a.N >> 3, lq = threadIdx.x % Q;
  float w[4][8], bq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bq[e] = a.bias ? a.bias[8 * lq + e] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k][e] = k < a.K ? (float)Wg[(long)(8 * lq + e) * a.Kpad + k] : 0.f;
  }
  const long rstride = (long)gridDim.x * (256 / Q);
  for (long row = (long)blockIdx.x * (256 / Q) + threadIdx.x / Q; row < a.M; row += rstride) {
    const _Float16* x = Ag + row * a.lda;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = bq[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < a.K) { const float xv = (float)x[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = fmaf(xv, w[k][e], y[e]); }
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)y[e];
    *reinterpret_cast<h8*>(Cg + row * a.ldc + 8 * lq) = o;
  }
}
static int hconv1x1_stream_dispatch(const IgemmArgs& a, hipStream_t st) {
  static const int on = getenv("ARCO_CONV1X1_STREAM") ? atoi(getenv("ARCO_CONV1X1_STREAM")) : 1;
  if (!on || a.stat_sum || a.R || a.pro.mean || a.Rup || a.ksplit > 1 || a.batch > 1 || a.M < 65536) return -1;
  long blocks;
  if (a.N >= 1 && a.N <= 4 && (a.K == 8 || a.K == 16 || a.K == 32) && (a.lda & 7) == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0) {
    const int Q = a.K / 8;
    blocks = (a.M + (256 / Q) * 8 - 1) / ((256 / Q) * 8); if (blocks > 16384) blocks = 16384;
    if (Q == 1) hipLaunchKernelGGL(hconv1x1_narrow_out_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (Q == 2) hipLaunchKernelGGL(hconv1x1_narrow_out_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(hconv1x1_narrow_out_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return arco_launch_status();
  }
  if (a.K >= 1 && a.K <= 4 && (a.N == 8 || a.N == 16 || a.N == 32) && (a.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0) {
    const int Q = a.N / 8;
    blocks = (a.M + (256 / Q) * 8 - 1) / ((256 / Q) * 8); if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(hconv1x1_narrow_in_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return arco_launch_status();
  }
  return -1;
}

// entry of the f16-storage convolutions (called from arco_conv3d_fwd with mma == 4).  a.Kpad = ceil32(K) (the f16 pack)
int hconv_dispatch(const IgemmArgs& a, int taps, hipStream_t st, int* q) {
  if ((taps != 1 && ((a.K & 7) != 0 || (a.lda & 7) != 0)) || a.R != nullptr) return ARCO_ERR_UNSUPPORTED;
  if (!q && (((reinterpret_cast<uintptr_t>(a.A) & 15) != 0 && (a.K & 7) == 0) || (reinterpret_cast<uintptr_t>(a.Wp) & 15) != 0)) return ARCO_ERR_ARG;
  if (taps == 27) {
    if (hconv_rw_eligible(a)) {
      static const int mode = getenv("ARCO_HCONV_RW") ? atoi(getenv("ARCO_HCONV_RW")) : 1;
      if (a.K == 16) return ((mode & 4) && a.H % 16 == 0) ? launch_hconv_rw<1, 1, 16>(a, st, q) : launch_hconv_rw<1, 1, 8>(a, st, q);
      return launch_hconv_rw<2, 2, 16>(a, st, q);
    }
    { const int r = hconv_fc_dispatch(a, st, q); if (r != -1) return r; }      // the pipelined flat-tile form (32 .. 256 channels)
    if ((a.W & 15) != 0 && a.W + 2 <= HFLAT_WPMAX) return hconv_dispatch3<true>(a, st, q);
    return hconv_dispatch3<false>(a, st, q);
  }
  if (taps == 1) {
    if (!q) { const int r = hconv1x1_stream_dispatch(a, st); if (r != -1) return r; }
    if (a.Npad <= 16) return launch_hconv<1, 256, 16, 4, 1, 1, false>(a, st, q);
    if (a.Npad <= 32) return launch_hconv<1, 128, 32, 4, 1, 1, false>(a, st, q);
    if (a.M * (long)a.Npad <= 4096l * 1024) {
      const long tiles64 = ((a.M + 63) / 64) * ((a.Npad + 63) / 64);
      if (tiles64 < 192 && a.stat_groups <= 1) return launch_hconv<1, 32, 64, 2, 2, 1, false>(a, st, q);
      return launch_hconv<1, 64, 64, 2, 2, 1, false>(a, st, q);
    }
    return launch_hconv<1, 128, 128, 2, 2, 1, false>(a, st, q);
  }
  return ARCO_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
struct HWgradArgs {
  const _Float16* dZ; long ldz; int Cout;
  const _Float16* X; long ldx; int Cin;
  int taps, NB, H, W, D3; long M;
  float* partial;      // [chunks][taps][CoutPad][CinPad]
  int CoutPad, CinPad, n_tiles;
};

constexpr int h_row_dw(int chans) { return chans == 16 ? 8 : (chans == 32 ? 24 : 40); }   // 8 x odd dwords

// two transposing reads -> the 8 k (= pixel) values of one channel per lane: elements 0-3 from the block at `p`, 4-7 from `p + off2`
__device__ __forceinline__ h8 tr_pair(const unsigned* p, int off2_dw) {
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(p));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(p + off2_dw));
  return __builtin_bit_cast(h8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// NT = 9: the nine taps of one plane pair (3-D: depth tap = blockIdx.z, input plane x + dd - 1), tiles of 8 x 16 pixels (ragged
// at the plane's edges); NT = 1: GEMM form, tiles of 128 consecutive rows.
template <int CO_T, int CI_T, int NT>
__global__ __launch_bounds__(256) void hwgrad_kernel(HWgradArgs a) {
  constexpr int CO_B = 16 * CO_T, CI_B = 16 * CI_T;
  constexpr int RSZ = h_row_dw(CO_B), RSX = h_row_dw(CI_B);
  constexpr int XR = NT == 9 ? 180 : 128;
  constexpr int PZ = CO_B / 8, PX = CI_B / 8;                     // 16-byte pieces per row
  constexpr int NZ = (128 * PZ + 255) / 256, NX = (XR * PX + 255) / 256;
  constexpr int NSUB = CO_T * CI_T;
  extern __shared__ __attribute__((aligned(16))) unsigned smem_w[];
  unsigned* const Zs = smem_w;                   // [128][RSZ]
  unsigned* const Xs = smem_w + 128 * RSZ;       // [XR][RSX]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int dd = NT == 9 ? blockIdx.z : 0;
  const int dpl = (NT == 9 && a.taps == 27) ? dd - 1 : 0;
  const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
  const int ci_tiles = a.CinPad / CI_B;
  const int co0 = (blockIdx.y / ci_tiles) * CO_B, ci0 = (blockIdx.y % ci_tiles) * CI_B;
  const bool vz = (a.Cout & 7) == 0 && (a.ldz & 7) == 0, vx = (a.Cin & 7) == 0 && (a.ldx & 7) == 0;

  f32x4 acc[NT][CO_T][CI_T];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < CO_T; ++i)
#pragma unroll
      for (int j = 0; j < CI_T; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};

  // 32-channel rows (4 pieces, 24-dword stride): a ds_write_b128 is served 8 lanes at a time over 32 banks - rows r and r + 1 of a
  // lane octet overlap in 8 banks, rows r and r + 2 do not: the two low bits of the row index are swapped in the thread -> piece map
  auto rperm = [](int p, int pieces) { return pieces == 4 ? ((p & ~3) | ((p & 1) << 1) | ((p >> 1) & 1)) : p; };
  u32x4 pz[NZ], px_[NX];
  auto load8 = [&](const _Float16* src, int c, int C, bool vec) -> u32x4 {
    if (vec) return c < C ? *reinterpret_cast<const u32x4*>(src) : u32x4{0, 0, 0, 0};
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) if (c + e < C) v[e] = src[e];
    return __builtin_bit_cast(u32x4, v);
  };
  auto fetch = [&](int t) {
    int img = 0, y0 = 0, x0 = 0; long m0 = 0; bool plane_ok = true;
    if (NT == 9) {
      int tt = t; const int tx = tt % tiles_x; tt /= tiles_x; const int ty = tt % tiles_y; img = tt / tiles_y;
      y0 = ty * 8; x0 = tx * 16;
      const int pl = a.taps == 27 ? img % a.D3 + dpl : 0;
      plane_ok = pl >= 0 && pl < a.D3;
    } else {
      m0 = (long)t * 128;
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int idx = tid + i * 256, p = rperm(idx / PZ, PZ), q = idx % PZ, c = co0 + 8 * q;
      u32x4 v = u32x4{0, 0, 0, 0};
      if (idx < 128 * PZ && plane_ok) {
        long pix = -1;
        if (NT == 9) { const int y = y0 + (p >> 4), x = x0 + (p & 15); if (y < a.H && x < a.W) pix = ((long)img * a.H + y) * a.W + x; }
        else if (m0 + p < a.M) pix = m0 + p;
        if (pix >= 0) v = load8(a.dZ + pix * a.ldz + c, c, a.Cout, vz);
      }
      pz[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + i * 256, r = rperm(idx / PX, PX), q = idx % PX, c = ci0 + 8 * q;
      u32x4 v = u32x4{0, 0, 0, 0};
      if (idx < XR * PX && plane_ok) {
        long pix = -1;
        if (NT == 9) {
          const int y = y0 + r / 18 - 1, x = x0 + r % 18 - 1;
          if (y >= 0 && y < a.H && x >= 0 && x < a.W) pix = ((long)(img + dpl) * a.H + y) * a.W + x;
        } else if (m0 + r < a.M) pix = m0 + r;
        if (pix >= 0) v = load8(a.X + pix * a.ldx + c, c, a.Cin, vx);
      }
      px_[i] = v;
    }
  };

  // lane geometry of the transposing reads: lane 4q+p of a 16-lane group supplies row q, column quad p of its 4 x 16 block.
  // K step = wave: pixel (h, G = g, q) of the step = tile row 2 wid + h, column 4 g + q  (GEMM form: row 32 wid + 16 h + 4 g + q)
  const int tq = li >> 2, tp = li & 3;
  const int zrow = NT == 9 ? (2 * wid) * 16 + 4 * g + tq : 32 * wid + 4 * g + tq;
  const int xrow = NT == 9 ? (2 * wid) * 18 + 4 * g + tq : zrow;
  const unsigned* const zb = Zs + zrow * RSZ + 2 * tp;
  const unsigned* const xb = Xs + xrow * RSX + 2 * tp;

  int t = blockIdx.x;
  if (t < a.n_tiles) fetch(t);
  while (t < a.n_tiles) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) { const int idx = tid + i * 256; if (idx < 128 * PZ) *reinterpret_cast<u32x4_ma*>(Zs + rperm(idx / PZ, PZ) * RSZ + 4 * (idx % PZ)) = pz[i]; }
#pragma unroll
    for (int i = 0; i < NX; ++i) { const int idx = tid + i * 256; if (idx < XR * PX) *reinterpret_cast<u32x4_ma*>(Xs + rperm(idx / PX, PX) * RSX + 4 * (idx % PX)) = px_[i]; }
    __syncthreads();
    const int next = t + gridDim.x;
    if (next < a.n_tiles) fetch(next);
    h8 zf[CO_T];
#pragma unroll
    for (int i = 0; i < CO_T; ++i) zf[i] = tr_pair(zb + 8 * i, 16 * RSZ);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int toff = NT == 9 ? ((tap / 3) * 18 + tap % 3) * RSX : 0;
#pragma unroll
      for (int j = 0; j < CI_T; ++j) {
        const h8 xf = tr_pair(xb + toff + 8 * j, (NT == 9 ? 18 : 16) * RSX);
#pragma unroll
        for (int i = 0; i < CO_T; ++i)
          acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf[i], xf, acc[tap][i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    t = next;
  }

  // sum the four K-step waves (once per launch, tap by tap through LDS), then one slab per workgroup
  float* const red = reinterpret_cast<float*>(smem_w);     // [4 waves][NSUB][4][64]
#pragma unroll
  for (int tap = 0; tap < NT; ++tap) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CO_T; ++i)
#pragma unroll
      for (int j = 0; j < CI_T; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wid * NSUB + i * CI_T + j) * 4 + r) * 64 + lane] = acc[tap][i][j][r];
    __syncthreads();
    float* out = a.partial + (((long)blockIdx.x * a.taps + dd * NT + tap) * a.CoutPad) * a.CinPad;
    const int r = wid;                                        // this thread sums element (r, lane) of every sub-tile
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const float v = (red[((0 * NSUB + s) * 4 + r) * 64 + lane] + red[((1 * NSUB + s) * 4 + r) * 64 + lane]) +
                      (red[((2 * NSUB + s) * 4 + r) * 64 + lane] + red[((3 * NSUB + s) * 4 + r) * 64 + lane]);
      out[(long)(co0 + (s / CI_T) * 16 + 4 * g + r) * a.CinPad + ci0 + (s % CI_T) * 16 + li] = v;
    }
  }
}

template <int CO_T, int CI_T, int NT>
static void launch_hwgrad(const HWgradArgs& a, dim3 grid, hipStream_t st) {
  constexpr int RSZ = h_row_dw(16 * CO_T), RSX = h_row_dw(16 * CI_T), XR = NT == 9 ? 180 : 128;
  size_t sh = (size_t)(128 * RSZ + XR * RSX) * 4;
  const size_t rd = (size_t)4 * CO_T * CI_T * 4 * 64 * 4;
  if (sh < rd) sh = rd;
  auto kern = hwgrad_kernel<CO_T, CI_T, NT>;
  static unsigned long long attr_set = 0;
  if (sh > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); }
  hipLaunchKernelGGL(kern, grid, dim3(256), sh, st, a);
}

// dW (+)= weight gradient from f16 dZ / X; ws sized by arco_wgrad_ws_floats (same slab reservation as the fp32 kernels)
int hwgrad_dispatch(const void* dZ, long ld_dz, int Cout, const void* in, long ld_in, int Cin, int taps, int NB, int D3, int H, int W,
                    float* ws, float* dW, int accumulate, hipStream_t st) {
  HWgradArgs a{};
  a.dZ = reinterpret_cast<const _Float16*>(dZ); a.ldz = ld_dz; a.Cout = Cout;
  a.X = reinterpret_cast<const _Float16*>(in); a.ldx = ld_in; a.Cin = Cin;
  a.taps = taps; a.NB = NB; a.H = H; a.W = W; a.D3 = D3; a.M = (long)NB * H * W; a.partial = ws;
  if ((Cin & 7) != 0 || (ld_in & 7) != 0 || (ld_dz & 1) != 0) return ARCO_ERR_UNSUPPORTED;
  long chunks;
  if (taps >= 9) {
    const int hco = Cout > 16 ? 32 : 16, hci = Cin > 16 ? 32 : 16;
    a.CoutPad = (Cout + hco - 1) / hco * hco; a.CinPad = (Cin + hci - 1) / hci * hci;
    a.n_tiles = NB * ((H + 7) / 8) * ((W + 15) / 16);
    const int zdim = taps / 9, ydim = (a.CoutPad / hco) * (a.CinPad / hci);
    static const long target = getenv("ARCO_HWGRAD_TARGET") ? atol(getenv("ARCO_HWGRAD_TARGET")) : 512;
    chunks = target / ((long)zdim * ydim); if (chunks > a.n_tiles) chunks = a.n_tiles; if (chunks < 1) chunks = 1;
    const dim3 grid((unsigned)chunks, ydim, zdim);
    if (hco == 32 && hci == 32) launch_hwgrad<2, 2, 9>(a, grid, st);
    else if (hco == 32) launch_hwgrad<2, 1, 9>(a, grid, st);
    else if (hci == 32) launch_hwgrad<1, 2, 9>(a, grid, st);
    else launch_hwgrad<1, 1, 9>(a, grid, st);
  } else {
    const int co_b = Cout >= 64 ? 64 : (Cout > 16 ? 32 : 16), ci_b = Cin >= 64 ? 64 : (Cin > 16 ? 32 : 16);
    a.CoutPad = (Cout + co_b - 1) / co_b * co_b; a.CinPad = (Cin + ci_b - 1) / ci_b * ci_b;
    a.n_tiles = (int)((a.M + 127) / 128);
    const long ydim = (long)(a.CoutPad / co_b) * (a.CinPad / ci_b);
    static const long target1 = getenv("ARCO_HWGRAD1_TARGET") ? atol(getenv("ARCO_HWGRAD1_TARGET")) : 512;
    chunks = target1 / ydim; if (chunks < 1) chunks = 1; if (chunks > a.n_tiles) chunks = a.n_tiles;
    const dim3 grid((unsigned)chunks, (unsigned)ydim, 1);
#define HW1(CO, CI) if (co_b == 16 * CO && ci_b == 16 * CI) launch_hwgrad<CO, CI, 1>(a, grid, st)
    HW1(4, 4); else HW1(4, 2); else HW1(4, 1); else HW1(2, 4); else HW1(2, 2); else HW1(2, 1); else HW1(1, 4); else HW1(1, 2); else HW1(1, 1);
#undef HW1
  }
  launch_wgrad_reduce(st, ws, (int)chunks, taps, a.CoutPad, a.CinPad, Cout, Cin, dW, accumulate);
  return arco_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// the fp32 <-> f16 boundary
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_h2f_kernel(const _Float16* __restrict__ x, long n4, long n, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const h4 v = *reinterpret_cast<const h4*>(x + 4 * i);
    *reinterpret_cast<f32x4*>(y + 4 * i) = __builtin_convertvector(v, f32x4);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[4 * n4 + threadIdx.x] = (float)x[4 * n4 + threadIdx.x];
}
__global__ __launch_bounds__(256) void cast_f2h_kernel(const float* __restrict__ x, long n4, long n, float scale, _Float16* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    // saturate at the largest finite f16: a scaled gradient above 65504 becomes +-65504 (a clipped value) instead of an inf that
    // the f16 backward would spread through every gradient below it; NaNs pass (fminf / fmaxf would hide them: selects keep them)
    f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i) * scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 65504.f ? 65504.f : (v[j] < -65504.f ? -65504.f : v[j]);
    *reinterpret_cast<h4*>(y + 4 * i) = __builtin_convertvector(v, h4);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float t = x[4 * n4 + threadIdx.x] * scale;
    t = t > 65504.f ? 65504.f : (t < -65504.f ? -65504.f : t);
    y[4 * n4 + threadIdx.x] = (_Float16)t;
  }
}

extern "C" {

// y[i] = (float) x[i]: a dense f16 activation -> fp32 (the V-Net's outputs leave the f16 region)
int arco_cast_h2f(const void* x, long n, float* y, void* stream) {
  ARCO_CHECK_ARG(x && y && n > 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
  long gb = (n / 4 + 255) / 256; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
  hipLaunchKernelGGL(cast_h2f_kernel, dim3((unsigned)gb), dim3(256), 0, as_stream(stream), reinterpret_cast<const _Float16*>(x), n / 4, n, y);
  return arco_launch_status();
}
// y[i] = (f16)(scale * x[i]): an fp32 gradient enters the f16 region multiplied by the loss scale
int arco_cast_f2h(const float* x, long n, float scale, void* y, void* stream) {
  ARCO_CHECK_ARG(x && y && n > 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
  long gb = (n / 4 + 255) / 256; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
  hipLaunchKernelGGL(cast_f2h_kernel, dim3((unsigned)gb), dim3(256), 0, as_stream(stream), x, n / 4, n, scale, reinterpret_cast<_Float16*>(y));
  return arco_launch_status();
}

}  // extern "C"
