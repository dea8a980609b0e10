// Implicit-GEMM convolution on the CDNA4 fp32 matrix cores (v_mfma_f32_16x16x4_f32):
//   out[pix][n] = sum_{tap, k} in[pix + tap][k] * Wp[tap][n][k]  (+bias, +residual)
// channels-last activations (rows of `ld` floats), packed weights [TAPS][Npad][Kpad].
// One kernel family serves the 1x1 convs (FeatureExtractor / q_representation /
// UpBlock.conv1x1: model_2D.py:20-55, train_arco_2d.py:231-234, unetWithArgs.py:72-83)
// and the 3x3 convs of ConvBlock (unetWithArgs.py:31-47), forward and dgrad
// (dgrad = same kernel with flipped+transposed packed weights).
//
// Tiling: 4 waves / workgroup; a wave owns A_T x C_T MFMA tiles of 16 pixels x 16
// channels.  Per K-chunk (KC channels) the input patch (with halo for 3x3) and the
// weight slab of every tap are staged in LDS with rows padded to KC+4 floats, so the
// ds_read_b128 fragment reads (4 consecutive k per lane) are bank-conflict free.
// A b128 fragment feeds 4 MFMAs: lane (i, g) holds k = 4g+j for j = 0..3.
#include "igemm_args.h"
#include <stdlib.h>


// FLAT (3x3 only): the M-tile is BM consecutive positions of the plane stored with a padded row stride Wp = W + 2
// (one halo column each side) instead of a TH x 16 pixel rectangle.  A tap is then a uniform shift dy*Wp + dx of the
// position, rows need no tiling, and the padding waste is 2 / (W + 2) instead of rounding BOTH plane dimensions up
// to the tile: 7x7 planes waste 23 % instead of 62 %, 28x28 planes 7 % instead of 23 % (the V-Net's deep levels).
constexpr int IGEMM_FLAT_WPMAX = 64;
// MMA = 1 / 2 (BASELINE.json configs[4], "fp16 MFMA conv"): tensors stay fp32 in HBM and LDS; the four consecutive K
// values a lane reads per fragment are rounded to f16 / bf16 in registers and fed to ONE v_mfma_f32_16x16x16_{f16,bf16}
// (same K-to-lane mapping as four 16x16x4 fp32 MFMAs), fp32 accumulation: 1/16 of the matrix-core time, so the 3x3x3
// levels become LDS/HBM-bound.  Opt-in (--conv_mma), tolerance 1e-2; never the default or the benchmark.
// MMA = 3 ("3 x bf16", the DEFAULT of the trainers): fp32-accurate products on the bf16 matrix cores.  An fp32 value is
// EXACTLY x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (3 x 8 significand bits, RNE
// remainders are signed), so x*y = x0y0 + (x0y1 + x1y0) + (x0y2 + x1y1 + x2y0) + O(2^-24 |xy|): six
// v_mfma_f32_16x16x32_bf16 (each product exact, fp32 accumulate) replace eight v_mfma_f32_16x16x4_f32 per 32 k:
// 96 matrix-core cycles instead of 256 (measured issue rates, tools/micro/mfma_rate.hip: 16 vs 32 cycles), error per
// product <= 2^-23 relative - the size of one fp32 rounding (tests/test_split_mma_gpu.py compares both modes with fp64).
// The split is done ONCE: weights are packed pre-split (pack kernels, format "split": [tap][Npad][Kpad32/16][3][16] bf16),
// activations are split when a tile is staged into LDS (3 bf16 planes per row, 24 dwords per 16 k - with this stride the
// ds_read_b128 fragment reads are bank-conflict free without padding).  3x3: KC = 16, a K = 32 MFMA step covers the 16
// channels of TWO taps (lanes g = 0,1 read tap 2s, g = 2,3 tap 2s+1; the 10th half-step multiplies a zero weight row);
// 1x1: KC = 32.
template <int TAPS, int BM, int BN, int WAVES_M, int WAVES_N, int KC, bool DB, bool VEC, int DEPTH, bool FLAT = false, int MMA = 0>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs a) {
  constexpr int TH = BM / 16;
  constexpr bool X3 = MMA == 3;
  constexpr bool SWP = X3 && TAPS == 1 && DEPTH == 1;      // 1x1 split-bf16 GEMMs accumulate D^T (16-byte epilogue accesses)
  static_assert(!X3 || (TAPS == 9 ? KC == 16 : KC == 32), "split-bf16 mode: KC = 16 (3x3, tap pairs) or 32 (1x1)");
  constexpr int AROWS = TAPS == 9 ? (FLAT ? BM + 2 * IGEMM_FLAT_WPMAX + 2 : (TH + 2) * 18) : BM;
  constexpr int BROWS = TAPS * BN;
  constexpr int LDK = X3 ? (KC / 16) * 24 + (KC == 32 ? 8 : 0) : KC + 4;     // dwords per LDS row
  constexpr int A_T = BM / 16 / WAVES_M;
  constexpr int C_T = BN / 16 / WAVES_N;
  constexpr int Q4 = KC / 4;
  constexpr int PB = X3 ? (KC / 16) * 6 : Q4;          // 16-byte pieces per staged weight row
  constexpr int NA_IT = (AROWS * Q4 + 255) / 256;
  constexpr int NB_IT = (BROWS * PB + 255) / 256;
  constexpr int BUF = (AROWS + BROWS + (X3 ? 1 : 0)) * LDK;          // floats per LDS buffer (X3: + one zero weight row)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (TAPS == 1 && a.batch > 1) {       // batched GEMM (grouped InfoNCE: one problem per class)
    const long z = blockIdx.z;
    a.A += z * a.batchA; a.Wp += z * a.batchW; a.C += z * a.batchC;
  }

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int li = lane & 15, g = lane >> 4;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so the
  // N-tiles of one M-tile (which share the input rows) are given consecutive slots of ONE XCD.
  int mblk, nblk;
  {
    const int T = a.n_mblocks * a.n_nblocks, L = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = L & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    mblk = v / a.n_nblocks; nblk = v - mblk * a.n_nblocks;
  }
  const int n0 = nblk * BN;

  // tile origin
  long m0 = 0; int img = 0, y0 = 0, x0 = 0;
  const int Wp = a.W + 2;                 // FLAT: padded row stride
  int f0 = 0;                             // FLAT: first padded-plane position of this block
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
  // rows of this block end at m_lim: the tensor's end, or (GEMM form with grouped BN statistics) the end of the
  // block's BN group - M-blocks are laid out per group so that no stat slab mixes two groups
  long m_lim = a.M;
  if (TAPS == 1 && a.stat_groups > 1) {
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

  // chunk-invariant staging geometry: source row pointers (null = zero fill) and LDS offsets
  const float* srcA[NA_IT]; int ldsA[NA_IT], kA[NA_IT];
#pragma unroll
  for (int it = 0; it < NA_IT; ++it) {
    const int idx = tid + it * 256;
    srcA[it] = nullptr; ldsA[it] = -1; kA[it] = 0;
    if (idx < AROWS * Q4) {
      // split-bf16 GEMM tiles (Q4 = 8 pieces per row, 56-dword rows): the 16 lanes of a ds_write_b64 group would hold rows r, r + 1,
      // which overlap in 8 of the 32 banks (PMC: 19-37 % of these kernels' LDS cycles were conflicts); rows r, r + 2 do not
      const int q = idx % Q4, r_ = idx / Q4;
      const int row = (X3 && TAPS == 1 && (AROWS & 3) == 0) ? ((r_ & ~3) | ((r_ & 1) << 1) | ((r_ >> 1) & 1)) : r_;
      long pix = -1;
      if (TAPS == 9 && FLAT) {
        const int pidx = f0 + row - 1;               // position in the plane padded by one halo row / column
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
      ldsA[it] = X3 ? row * LDK + (q / 4) * 24 + (q % 4) * 2 : row * LDK + 4 * q; kA[it] = 4 * q;
      if (pix >= 0) srcA[it] = a.A + pix * a.lda + 4 * q;
    }
  }
  const float* srcB[NB_IT]; int ldsB[NB_IT], kB[NB_IT];
#pragma unroll
  for (int it = 0; it < NB_IT; ++it) {
    const int idx = tid + it * 256;
    srcB[it] = nullptr; ldsB[it] = -1; kB[it] = 0;
    if (idx < BROWS * PB) {
      const int row = idx / PB, q = idx - row * PB;
      const int tap = row / BN, n = row - tap * BN;
      ldsB[it] = AROWS * LDK + row * LDK + 4 * q; kB[it] = X3 ? (q / 6) * 16 : 4 * q;
      if (n0 + n < a.Npad)      // X3: packed split rows, 24 dwords per 16 k
        srcB[it] = X3 ? a.Wp + ((long)tap * a.Npad + n0 + n) * a.Kg * 24 + 4 * q
                      : a.Wp + ((long)tap * a.Npad + n0 + n) * a.Kpad + 4 * q;
    }
  }

  // 3-D (DEPTH==3): a 3x3x3 conv is the sum over the depth tap dd of a 3x3 conv of input plane x+dd-1
  // with weight slice Wp[dd*9 .. dd*9+8]; the K loop simply runs DEPTH times over shifted planes.
  const int nchunks = (a.Kpad + KC - 1) / KC;        // (X3: Kpad = ceil16(K) for 3x3, ceil32(K) for 1x1, set by the entry point)
  const int plane = DEPTH == 3 ? img % a.D3 : 0;
  const long plane_elems = (long)a.H * a.W * a.lda;
  const long wslice = X3 ? (long)9 * a.Npad * a.Kg * 24 : (long)9 * a.Npad * a.Kpad;
  f32x4 ra[NA_IT], rb[NB_IT];
  auto load_chunk = [&](int itc) {
    const int dd = DEPTH == 3 ? itc / nchunks : 0;
    const int kc0 = (DEPTH == 3 ? itc - dd * nchunks : itc) * KC;
    const bool plane_ok = DEPTH == 3 ? (plane + dd - 1 >= 0 && plane + dd - 1 < a.D3) : true;
    const long aoff = DEPTH == 3 ? (long)(dd - 1) * plane_elems + kc0 : kc0;
    const long bk0 = X3 ? (long)(kc0 / 16) * 24 : kc0;
    const long boff = DEPTH == 3 ? (long)dd * wslice + bk0 : bk0;
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) {
      f32x4 v = f32x4{0, 0, 0, 0};
      const int k = kc0 + kA[it];
      if constexpr (VEC) {
        // (the scalar variant lives in its own instantiation: sharing registers between the two load
        //  forms makes hipcc wait vmcnt(0) before every vector load and serialises the prefetch)
        if (srcA[it] && plane_ok && k < a.K) v = *reinterpret_cast<const f32x4*>(srcA[it] + aoff);
      } else {
        if (srcA[it] && plane_ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < a.K) v[e] = srcA[it][aoff + e];
        }
      }
      ra[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      f32x4 v = f32x4{0, 0, 0, 0};
      if (srcB[it] && kc0 + kB[it] < a.Kpad) v = *reinterpret_cast<const f32x4*>(srcB[it] + boff);
      rb[it] = v;
    }
  };
  auto store_chunk = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) if (ldsA[it] >= 0) {
      if constexpr (X3) {      // activations: split into the three bf16 planes of the row's 16-k group
        u32x2 p0, p1, p2;
        split3_bf16x4(ra[it], p0, p1, p2);
        unsigned int* d = reinterpret_cast<unsigned int*>(buf) + ldsA[it];
        *reinterpret_cast<u32x2_ma*>(d) = p0; *reinterpret_cast<u32x2_ma*>(d + 8) = p1; *reinterpret_cast<u32x2_ma*>(d + 16) = p2;
      } else {
        *reinterpret_cast<f32x4*>(&buf[ldsA[it]]) = ra[it];
      }
    }
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) if (ldsB[it] >= 0) *reinterpret_cast<f32x4*>(&buf[ldsB[it]]) = rb[it];
  };
  auto compute = [&](const float* buf) {
    const float* As = buf; const float* Bs = buf + AROWS * LDK;
    if constexpr (X3) {
      constexpr int NSTEP = TAPS == 9 ? 5 : 1;
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int tapL = TAPS == 9 ? 2 * st + (g >> 1) : 0;
        const bool zt = tapL > 8;                       // the 10th half-step: zero weight row
        const int tapA = zt ? 8 : tapL;
        const int dy = TAPS == 9 ? tapA / 3 : 0, dx = TAPS == 9 ? tapA % 3 : 0;
        const int koff = TAPS == 9 ? (g & 1) * 4 : (g >> 1) * 24 + (g & 1) * 4;
        bf16x8 fa[A_T][3];
#pragma unroll
        for (int at = 0; at < A_T; ++at) {
          const int s = wm * A_T + at;
          const int row = TAPS == 9 ? (FLAT ? s * 16 + li + dy * Wp + dx : (s + dy) * 18 + li + dx) : s * 16 + li;
          const float* p = &As[row * LDK + koff];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[at][pl] = lds_bf16x8(p + 8 * pl);
        }
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) {
          const int row = zt ? BROWS : tapA * BN + (wn * C_T + ct) * 16 + li;
          const float* p = &Bs[row * LDK + koff];
          bf16x8 fb[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fb[pl] = lds_bf16x8(p + 8 * pl);
#pragma unroll
          for (int at = 0; at < A_T; ++at) {       // small terms first
            f32x4 c = acc[at][ct];
            if constexpr (SWP) {                   // D = W . X^T: a lane ends up with 4 consecutive channels of one pixel
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[at][2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[2], fa[at][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[at][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[at][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[1], fa[at][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0], fa[at][0], c, 0, 0, 0);
            } else {
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][2], fb[0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][0], fb[2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][1], fb[1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][1], fb[0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][0], fb[1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][0], fb[0], c, 0, 0, 0);
            }
            acc[at][ct] = c;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
#pragma unroll
      for (int kk = 0; kk < KC / 16; ++kk) {
        f32x4 af[A_T], bf[C_T];
#pragma unroll
        for (int at = 0; at < A_T; ++at) {
          const int s = wm * A_T + at;
          const int row = TAPS == 9 ? (FLAT ? s * 16 + li + dy * Wp + dx : (s + dy) * 18 + li + dx) : s * 16 + li;
          af[at] = *reinterpret_cast<const f32x4*>(&As[row * LDK + kk * 16 + 4 * g]);
        }
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) {
          const int row = tap * BN + (wn * C_T + ct) * 16 + li;
          bf[ct] = *reinterpret_cast<const f32x4*>(&Bs[row * LDK + kk * 16 + 4 * g]);
        }
        if constexpr (MMA == 1) {
          f16x4 ah[A_T], bh[C_T];
#pragma unroll
          for (int at = 0; at < A_T; ++at) ah[at] = to_f16x4(af[at]);
#pragma unroll
          for (int ct = 0; ct < C_T; ++ct) bh[ct] = to_f16x4(bf[ct]);
#pragma unroll
          for (int at = 0; at < A_T; ++at)
#pragma unroll
            for (int ct = 0; ct < C_T; ++ct)
              acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[at], bh[ct], acc[at][ct], 0, 0, 0);
        } else if constexpr (MMA == 2) {
          s16x4 ah[A_T], bh[C_T];
#pragma unroll
          for (int at = 0; at < A_T; ++at) ah[at] = to_bf16x4(af[at]);
#pragma unroll
          for (int ct = 0; ct < C_T; ++ct) bh[ct] = to_bf16x4(bf[ct]);
#pragma unroll
          for (int at = 0; at < A_T; ++at)
#pragma unroll
            for (int ct = 0; ct < C_T; ++ct)
              acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[at], bh[ct], acc[at][ct], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int at = 0; at < A_T; ++at)
#pragma unroll
              for (int ct = 0; ct < C_T; ++ct)
                acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[at][j], bf[ct][j], acc[at][ct], 0, 0, 0);
        }
      }
    }
  };

  // main loop: the global loads of chunk k+1 are in flight while chunk k is on the matrix cores
  // (split-K launches: this block covers the chunk range of slab blockIdx.y and writes its own output slab)
  int c_first = 0, niter = DEPTH * nchunks;
  if (TAPS == 1 && a.ksplit > 1) {
    const int per = (nchunks + a.ksplit - 1) / a.ksplit;
    c_first = blockIdx.y * per;
    niter = min(nchunks, c_first + per);
    a.C += (long)blockIdx.y * a.slab_stride;
  }
  if constexpr (X3) {       // the zero weight row behind the staged rows (both buffers)
    for (int i = tid; i < LDK * (DB ? 2 : 1); i += 256) smem[(i / LDK) * BUF + (AROWS + BROWS) * LDK + i % LDK] = 0.f;
  }
  if (c_first < niter) {
    load_chunk(c_first);
    store_chunk(DB ? smem + (c_first & 1) * BUF : smem);
  }
  __syncthreads();
  for (int c = c_first; c < niter; ++c) {
    float* cur = DB ? smem + (c & 1) * BUF : smem;
    float* nxt = DB ? smem + ((c + 1) & 1) * BUF : smem;
    const bool more = c + 1 < niter;
    if (more) load_chunk(c + 1);
    compute(cur);
    if (more) {
      if (!DB) __syncthreads();          // single buffer: everyone done reading before it is overwritten
      store_chunk(nxt);
    }
    __syncthreads();
  }

  // ---- epilogue: bias, residual, store, BN partial statistics
  if constexpr (SWP) {
    // split-bf16 GEMMs (1x1 convs): the accumulators are D^T tiles - lane (li, g) holds channels 4g .. 4g+3 of pixel li:
    // one 16-byte store (and residual load) per tile instead of four 4-byte ones
    float t1[C_T][4], t2[C_T][4];
    const bool v4 = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0) &&
                    (!a.R || (((a.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.R) & 15) == 0)));
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct) {
      const int nb = n0 + (wn * C_T + ct) * 16 + 4 * g;
      f32x4 bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { bv[r] = (a.bias && nb + r < a.N) ? a.bias[nb + r] : 0.f; t1[ct][r] = 0.f; t2[ct][r] = 0.f; }
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
        const long m = m0 + (wm * A_T + at) * 16 + li;
        if (m >= m_lim) continue;
        f32x4 v = acc[at][ct] + bv;
        if (v4 && nb + 3 < a.N) {
          if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + m * a.ldr + nb);
          *reinterpret_cast<f32x4*>(a.C + m * a.ldc + nb) = v;
#pragma unroll
          for (int r = 0; r < 4; ++r) { t1[ct][r] += v[r]; t2[ct][r] += v[r] * v[r]; }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < a.N) {
              float x = v[r];
              if (a.R) x += a.R[m * a.ldr + nb + r];
              a.C[m * a.ldc + nb + r] = x;
              t1[ct][r] += x; t2[ct][r] += x * x;
            }
        }
      }
    }
    if (a.stat_sum) {
      float* red = smem;   // reuse: [2][WAVES_M][BN]
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
            red[(0 * WAVES_M + wm) * BN + nl] = v1;
            red[(1 * WAVES_M + wm) * BN + nl] = v2;
          }
        }
      __syncthreads();
      for (int nl = tid; nl < BN; nl += 256) {
        const int n = n0 + nl;
        if (n < a.N) {
          float v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES_M; ++w) { v1 += red[(0 * WAVES_M + w) * BN + nl]; v2 += red[(1 * WAVES_M + w) * BN + nl]; }
          a.stat_sum[(long)n * a.n_mblocks + mblk] = v1;
          a.stat_sq[(long)n * a.n_mblocks + mblk] = v2;
        }
      }
    }
    return;
  }
  float s1[C_T], s2[C_T];
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct) { s1[ct] = 0.f; s2[ct] = 0.f; }
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct) {
    const int n = n0 + (wn * C_T + ct) * 16 + li;
    const bool nok = n < a.N;
    const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int at = 0; at < A_T; ++at) {
      const int s = wm * A_T + at;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        long pix = -1;
        if (TAPS == 9 && FLAT) {
          const int f = f0 + s * 16 + i, y = f / Wp, xq = f - y * Wp;
          if (y < a.H && xq >= 1 && xq <= a.W) pix = ((long)img * a.H + y) * a.W + xq - 1;
        } else if (TAPS == 9) {
          const int y = y0 + s, x = x0 + i;
          if (y < a.H && x < a.W) pix = ((long)img * a.H + y) * a.W + x;
        } else {
          const long m = m0 + s * 16 + i;
          if (m < m_lim) pix = m;
        }
        if (pix >= 0 && nok) {
          float v = acc[at][ct][r] + bv;
          if (a.R) v += a.R[pix * a.ldr + n];
          a.C[pix * a.ldc + n] = v;
          s1[ct] += v; s2[ct] += v * v;
        }
      }
    }
  }
  if (a.stat_sum) {
    float* red = smem;   // reuse: [2][WAVES_M][BN]
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct) {
      float v1 = s1[ct], v2 = s2[ct];
      v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
      v2 += __shfl_xor(v2, 16, 64); v2 += __shfl_xor(v2, 32, 64);
      if (g == 0) {
        const int nl = (wn * C_T + ct) * 16 + li;
        red[(0 * WAVES_M + wm) * BN + nl] = v1;
        red[(1 * WAVES_M + wm) * BN + nl] = v2;
      }
    }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += 256) {
      const int n = n0 + nl;
      if (n < a.N) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) { v1 += red[(0 * WAVES_M + w) * BN + nl]; v2 += red[(1 * WAVES_M + w) * BN + nl]; }
        a.stat_sum[(long)n * a.n_mblocks + mblk] = v1;
        a.stat_sq[(long)n * a.n_mblocks + mblk] = v2;
      }
    }
  }
}

template <int TAPS, int BM, int BN, int WAVES_M, int WAVES_N, int KC, bool DB, bool VEC, int DEPTH, bool FLAT = false, int MMA = 0>
static int launch_igemm_v(const IgemmArgs& a, hipStream_t st, int* n_mblocks_out) {
  constexpr int TH = BM / 16;
  constexpr int AROWS = TAPS == 9 ? (FLAT ? BM + 2 * IGEMM_FLAT_WPMAX + 2 : (TH + 2) * 18) : BM;
  int mblocks;
  if (TAPS == 9 && FLAT) mblocks = a.NB * ((a.H * (a.W + 2) + BM - 1) / BM);
  else if (TAPS == 9) mblocks = a.NB * ((a.H + TH - 1) / TH) * ((a.W + 15) / 16);
  else if (a.stat_groups > 1) mblocks = a.stat_groups * (int)((a.M / a.stat_groups + BM - 1) / BM);   // per-group M-blocks
  else mblocks = (int)((a.M + BM - 1) / BM);
  if (n_mblocks_out) {
    n_mblocks_out[0] = mblocks;
    n_mblocks_out[1] = TAPS * 1000000 + BM * 1000 + BN + (FLAT ? 500000 : 0);   // instantiation id (query only)
    n_mblocks_out[2] = KC * 100 + DEPTH * 10 + (DB ? 1 : 0);
    return ARCO_OK;
  }
  constexpr int LDKL = MMA == 3 ? (KC / 16) * 24 + (KC == 32 ? 8 : 0) : KC + 4;
  size_t sh = (size_t)(DB ? 2 : 1) * (AROWS + TAPS * BN + (MMA == 3 ? 1 : 0)) * LDKL * sizeof(float);
  const size_t red = (size_t)2 * WAVES_M * BN * sizeof(float);
  if (sh < red) sh = red;
  auto kern = igemm_kernel<TAPS, BM, BN, WAVES_M, WAVES_N, KC, DB, VEC, DEPTH, FLAT, MMA>;
  static unsigned long long attr_set = 0;      // once per instantiation (never inside a stream capture after warm-up)
  if (sh > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); }
  IgemmArgs b = a;
  b.n_mblocks = mblocks;
  b.n_nblocks = (a.Npad + BN - 1) / BN;
  dim3 grid((unsigned)(mblocks * b.n_nblocks), (unsigned)(TAPS == 1 && a.ksplit > 1 ? a.ksplit : 1),
            (unsigned)(TAPS == 1 && a.batch > 1 ? a.batch : 1));
  hipLaunchKernelGGL(kern, grid, dim3(256), sh, st, b);
  return arco_launch_status();
}

template <int TAPS, int BM, int BN, int WAVES_M, int WAVES_N, int KC, bool DB, int DEPTH = 1, bool FLAT = false>
static int launch_igemm(const IgemmArgs& a, hipStream_t st, int* n_mblocks_out) {
  const bool vec = ((a.K & 3) == 0) && ((a.lda & 3) == 0);
  if constexpr ((TAPS == 9 && KC == 16) || (TAPS == 1 && KC == 32)) {   // split-bf16 (fp32-accurate) mode
    if (vec && a.mma == 3) return launch_igemm_v<TAPS, BM, BN, WAVES_M, WAVES_N, KC, (TAPS == 1 && BM * BN > 64 * 64) ? false : DB, true, DEPTH, FLAT, 3>(a, st, n_mblocks_out);
  }
  if (a.mma == 3) return ARCO_ERR_UNSUPPORTED;   // the caller asks arco_conv_split_ok() first
  if constexpr (DEPTH == 3 || TAPS == 1) {   // reduced-precision MFMA operands: the 3x3x3 kernels and the 1x1x1 GEMMs (vector loads)
    if (vec && a.mma == 1) return launch_igemm_v<TAPS, BM, BN, WAVES_M, WAVES_N, KC, DB, true, DEPTH, FLAT, 1>(a, st, n_mblocks_out);
    if (vec && a.mma == 2) return launch_igemm_v<TAPS, BM, BN, WAVES_M, WAVES_N, KC, DB, true, DEPTH, FLAT, 2>(a, st, n_mblocks_out);
  }
  if (vec) return launch_igemm_v<TAPS, BM, BN, WAVES_M, WAVES_N, KC, DB, true, DEPTH, FLAT>(a, st, n_mblocks_out);
  return launch_igemm_v<TAPS, BM, BN, WAVES_M, WAVES_N, KC, DB, false, DEPTH, FLAT>(a, st, n_mblocks_out);
}

// config choice shared by the launch and the "how many M-blocks" query.
// Spatial convs: prefer the largest tile that still gives >= 2 workgroups per CU (512 blocks); the mid/deep
// U-Net / V-Net levels have few pixels, and one 4-wave block per CU leaves the matrix pipe half idle.
static long conv_blocks(const IgemmArgs& a, int bm, int bn) {
  const int th = bm / 16;
  return (long)a.NB * ((a.H + th - 1) / th) * ((a.W + 15) / 16) * ((a.Npad + bn - 1) / bn);
}
// flat-position tiles for narrow planes whose width is not a multiple of 16 (the V-Net's 56 / 28 / 14 / 7)
static long flat_blocks(const IgemmArgs& a, int bm, int bn) {
  return (long)a.NB * ((a.H * (a.W + 2) + bm - 1) / bm) * ((a.Npad + bn - 1) / bn);
}
static long want_blocks(const IgemmArgs& a) {      // A/B knob: ARCO_IGEMM_WANT / ARCO_IGEMM_WANT3 (split-bf16 launches)
  static const long w0 = getenv("ARCO_IGEMM_WANT") ? atol(getenv("ARCO_IGEMM_WANT")) : 512;
  static const long w3 = getenv("ARCO_IGEMM_WANT3") ? atol(getenv("ARCO_IGEMM_WANT3")) : 512;
  return a.mma == 3 ? w3 : w0;
}
static int dispatch_flat3(const IgemmArgs& a, hipStream_t st, int* nmb) {
  const long want = want_blocks(a);
  if (a.Npad <= 16) return launch_igemm<9, 128, 16, 4, 1, 16, false, 3, true>(a, st, nmb);
  if (a.Npad <= 32) {
    static const int big = getenv("ARCO_IGEMM_256") ? atoi(getenv("ARCO_IGEMM_256")) : 0;      // A/B: 256-position tiles (4 x 2 MFMA tiles per wave)
    if (big && a.mma == 3 && flat_blocks(a, 256, 32) >= want) return launch_igemm<9, 256, 32, 4, 1, 16, false, 3, true>(a, st, nmb);
    if (flat_blocks(a, 128, 32) >= want) return launch_igemm<9, 128, 32, 4, 1, 16, false, 3, true>(a, st, nmb);
    return launch_igemm<9, 64, 32, 2, 2, 16, false, 3, true>(a, st, nmb);
  }
  if (flat_blocks(a, 128, 64) >= want) return launch_igemm<9, 128, 64, 4, 1, 16, false, 3, true>(a, st, nmb);
  if (flat_blocks(a, 64, 64) >= want) return launch_igemm<9, 64, 64, 2, 2, 16, false, 3, true>(a, st, nmb);
  if (flat_blocks(a, 64, 32) >= want) return launch_igemm<9, 64, 32, 2, 2, 16, false, 3, true>(a, st, nmb);
  return launch_igemm<9, 32, 32, 2, 2, 16, false, 3, true>(a, st, nmb);
}
template <int DEPTH>
static int dispatch_spatial(const IgemmArgs& a, hipStream_t st, int* nmb) {
  const long want = want_blocks(a);
  if (DEPTH == 3 && (a.W & 15) != 0 && a.W + 2 <= IGEMM_FLAT_WPMAX) return dispatch_flat3(a, st, nmb);
  if (a.Npad <= 16) {
    if (DEPTH == 1 && conv_blocks(a, 256, 16) >= want) return launch_igemm<9, 256, 16, 4, 1, 16, false, DEPTH>(a, st, nmb);
    if (conv_blocks(a, 128, 16) >= want || DEPTH == 3) return launch_igemm<9, 128, 16, 4, 1, 16, false, DEPTH>(a, st, nmb);
    return launch_igemm<9, 64, 16, 4, 1, 16, false, DEPTH>(a, st, nmb);
  }
  if (a.Npad <= 32) {
    if (conv_blocks(a, 128, 32) >= want) return launch_igemm<9, 128, 32, 4, 1, 16, false, DEPTH>(a, st, nmb);
    return launch_igemm<9, 64, 32, 2, 2, 16, false, DEPTH>(a, st, nmb);
  }
  if (conv_blocks(a, 128, 64) >= want) return launch_igemm<9, 128, 64, 4, 1, 16, false, DEPTH>(a, st, nmb);
  if (conv_blocks(a, 64, 64) >= want) return launch_igemm<9, 64, 64, 2, 2, 16, false, DEPTH>(a, st, nmb);
  if (conv_blocks(a, 64, 32) >= want) return launch_igemm<9, 64, 32, 2, 2, 16, false, DEPTH>(a, st, nmb);
  return launch_igemm<9, 32, 32, 2, 2, 16, false, DEPTH>(a, st, nmb);
}

// ---------------------------------------------------------------------------
// 3x3 convolution for the shallow levels (Cin in {16, 32}, Cout <= 32): persistent workgroups,
// weights resident in LDS for the whole launch, one input HALO tile staged per output tile
// (every input pixel is fetched once per tile instead of once per tap), all 9 taps fed to the
// MFMAs from LDS with no barrier in between, next tile's halo prefetched into registers while
// the current one is computed.  D = W (16 cout x 4 k) * X^T (4 k x 16 px): a lane ends up with
// 4 consecutive output channels of one pixel -> 16-byte coalesced stores.
//   tile = 8 rows x TW cols; wave w owns rows 2w, 2w+1.
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int TW>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(IgemmArgs a) {
  constexpr int TH = 8, HW_ = TW + 2, NP = (TH + 2) * HW_;
  constexpr int LDC = CIN + 4, Q4 = CIN / 4, KJ = CIN / 16, NT = COUT / 16, MT = 2 * (TW / 16);
  constexpr int UNITS = NP * Q4, NLD = (UNITS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                        // [9][COUT][LDC]
  float* As = smem + 9 * COUT * LDC;       // [NP][LDC]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long n_tiles = (long)a.NB * tiles_y * tiles_x;

  for (int u = tid; u < 9 * COUT * Q4; u += 256) {          // weights: Wp[tap][Npad][Kpad], Kpad == CIN
    const int q = u % Q4, r = u / Q4, n = r % COUT, tap = r / COUT;
    f32x4 v = {0, 0, 0, 0};
    if (n < a.Npad) v = *reinterpret_cast<const f32x4*>(a.Wp + ((long)tap * a.Npad + n) * a.Kpad + 4 * q);
    *reinterpret_cast<f32x4*>(&Ws[(tap * COUT + n) * LDC + 4 * q]) = v;
  }

  f32x4 pre[NLD];
  auto fetch = [&](long tile) {
    const int tx = tile % tiles_x; const long r = tile / tiles_x; const int ty = r % tiles_y; const int nb = r / tiles_y;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int u = tid + i * 256;
      const int hp = u / Q4, q = u % Q4, hy = hp / HW_, hx = hp % HW_;
      const int gy = y0 + hy, gx = x0 + hx;
      f32x4 v = {0, 0, 0, 0};
      if (u < UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        v = *reinterpret_cast<const f32x4*>(a.A + (((long)nb * a.H + gy) * a.W + gx) * a.lda + 4 * q);
      pre[i] = v;
    }
  };

  f32x4 s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = f32x4{0, 0, 0, 0}; s2[nt] = f32x4{0, 0, 0, 0}; }
  f32x4 bias4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int n = nt * 16 + 4 * g + e; bias4[nt][e] = (a.bias && n < a.N) ? a.bias[n] : 0.f; }

  // a workgroup stays inside ONE BN group of images (its stat slab belongs to that group): group g owns the
  // blocks [g*bpg, (g+1)*bpg) and the tiles [g*tpg, (g+1)*tpg)
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1, bpg = gridDim.x / n_grp;
  const long tpg = n_tiles / n_grp, t_end = (blockIdx.x / bpg + 1) * tpg;
  long tile = (blockIdx.x / bpg) * tpg + blockIdx.x % bpg;
  if (tile < t_end) fetch(tile);
  while (tile < t_end) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int u = tid + i * 256;
      if (u < UNITS) *reinterpret_cast<f32x4*>(&As[(u / Q4) * LDC + 4 * (u % Q4)]) = pre[i];
    }
    __syncthreads();
    const long next = tile + bpg;
    if (next < t_end) fetch(next);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        f32x4 bw[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[nt] = *reinterpret_cast<const f32x4*>(&Ws[(tap * COUT + nt * 16 + li) * LDC + 16 * j + 4 * g]);
        f32x4 ax[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int prow = 2 * w + mt / (TW / 16) + dy, pcol = (mt % (TW / 16)) * 16 + li + dx;
          ax[mt] = *reinterpret_cast<const f32x4*>(&As[(prow * HW_ + pcol) * LDC + 16 * j + 4 * g]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)                       // independent accumulators back to back
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[nt][e], ax[mt][e], acc[mt][nt], 0, 0, 0);
      }
    }
    {
      const int tx = tile % tiles_x; const long r = tile / tiles_x; const int ty = r % tiles_y; const int nb = r / tiles_y;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int gy = ty * TH + 2 * w + mt / (TW / 16), gx = tx * TW + (mt % (TW / 16)) * 16 + li;
        if (gy < a.H && gx < a.W) {
          const long pix = ((long)nb * a.H + gy) * a.W + gx;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + 4 * g;
            if (n < a.N) {
              f32x4 v = acc[mt][nt] + bias4[nt];
              if (a.R) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += a.R[pix * a.ldr + n + e];
              }
              if ((a.ldc & 3) == 0) *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + n) = v;
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) a.C[pix * a.ldc + n + e] = v[e];
              }
              s1[nt] += v; s2[nt] += v * v;
            }
          }
        }
      }
    }
    __syncthreads();
    tile = next;
  }

  if (a.stat_sum) {
    float* red = As;                       // [2][4 waves][COUT]; As is idle (barrier above / no tile at all)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v1 = s1[nt][e], v2 = s2[nt][e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
        if (li == 0) { red[(0 * 4 + w) * COUT + nt * 16 + 4 * g + e] = v1; red[(1 * 4 + w) * COUT + nt * 16 + 4 * g + e] = v2; }
      }
    __syncthreads();
    if (tid < COUT && tid < a.N) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) { v1 += red[(0 * 4 + ww) * COUT + tid]; v2 += red[(1 * 4 + ww) * COUT + tid]; }
      a.stat_sum[(long)tid * a.n_mblocks + blockIdx.x] = v1;
      a.stat_sq[(long)tid * a.n_mblocks + blockIdx.x] = v2;
    }
  }
}

// eligibility + launch; the block count (= BN-stat slabs per channel) is a pure function of the geometry
static bool halo_eligible(const IgemmArgs& a) {
  static const bool off = getenv("ARCO_CONV_HALO") && atoi(getenv("ARCO_CONV_HALO")) == 0;   // A/B switch
  if (off) return false;
  return (a.K == 16 || a.K == 32) && a.Kpad == a.K && a.Npad <= 32 && (a.N & 3) == 0 && (a.lda & 3) == 0;
}
template <int CIN, int COUT, int TW>
static int launch_halo(const IgemmArgs& a, hipStream_t st, int* q) {
  const long n_tiles = (long)a.NB * ((a.H + 7) / 8) * ((a.W + TW - 1) / TW);
  constexpr size_t sh = (size_t)(9 * COUT * (CIN + 4) + 10 * (TW + 2) * (CIN + 4)) * sizeof(float);
  constexpr int per_cu = sh * 4 <= 160 * 1024 ? 4 : (sh * 3 <= 160 * 1024 ? 3 : (sh * 2 <= 160 * 1024 ? 2 : 1));
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1;
  if (a.NB % n_grp != 0) return ARCO_ERR_ARG;
  long bpg = 256l * (per_cu > 3 ? 3 : per_cu) / n_grp; if (bpg > n_tiles / n_grp) bpg = n_tiles / n_grp;   // <= 3 waves/SIMD by VGPRs
  if (bpg < 1) bpg = 1;
  const long blocks = bpg * n_grp;
  if (q) { q[0] = (int)blocks; q[1] = 9 * 1000000 + 900000 + CIN * 1000 + COUT; q[2] = CIN * 100 + 10; return ARCO_OK; }
  auto kern = conv3x3_halo_kernel<CIN, COUT, TW>;
  static unsigned long long attr_set = 0;
  if (sh > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); }
  IgemmArgs b = a; b.n_mblocks = (int)blocks; b.n_nblocks = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), sh, st, b);
  return arco_launch_status();
}
static int dispatch_halo(const IgemmArgs& a, hipStream_t st, int* q) {
  if (a.K == 16) return a.Npad <= 16 ? launch_halo<16, 16, 32>(a, st, q) : launch_halo<16, 32, 32>(a, st, q);
  return a.Npad <= 16 ? launch_halo<32, 16, 16>(a, st, q) : launch_halo<32, 32, 16>(a, st, q);
}
// ---------------------------------------------------------------------------
// 3x3 convolution of an IMAGE (Cin <= 4: the network's first layer, 1 or 3 input channels) to 16 channels.
// 9*Cin <= 36 multiply-adds per output value: far too little K for the matrix pipe (the MFMA path pads K to 16
// and spends 16x the work) - this is a streaming kernel: 2 MB in, 33.5 MB out at 8x256x256.  A lane owns 4
// output channels of one pixel (16-byte coalesced stores), the 6x18 input halo and the weights sit in LDS,
// workgroups are persistent so the BN partials stay <= 1024 slabs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_image_kernel(IgemmArgs a) {
  constexpr int TH = 16, TW = 16, HW_ = TW + 2, NP = (TH + 2) * HW_;
  __shared__ float Xs[4][NP];            // [k][halo pixel]
  __shared__ __attribute__((aligned(16))) float Ws[9 * 4 * 16];   // [tap][k][n]
  __shared__ float red[2][4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int pl = tid >> 2, qd = tid & 3;                         // pixel of the tile (0..63), channel quad
  const int K = a.K;
  for (int u = tid; u < 9 * 4 * 16; u += 256) {
    const int n = u & 15, k = (u >> 4) & 3, tap = u >> 6;
    Ws[u] = (k < K && n < a.Npad) ? a.Wp[((long)tap * a.Npad + n) * a.Kpad + k] : 0.f;
  }
  f32x4 bias4 = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) if (a.bias && 4 * qd + e < a.N) bias4[e] = a.bias[4 * qd + e];
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long n_tiles = (long)a.NB * tiles_y * tiles_x;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1, bpg = gridDim.x / n_grp;      // one BN group per workgroup
  const long tpg = n_tiles / n_grp, t_end = (blockIdx.x / bpg + 1) * tpg;
  for (long t = (blockIdx.x / bpg) * tpg + blockIdx.x % bpg; t < t_end; t += bpg) {
    const int tx = t % tiles_x; const long r = t / tiles_x; const int ty = r % tiles_y; const int nb = r / tiles_y;
    __syncthreads();
    for (int u = tid; u < NP * K; u += 256) {
      const int k = u % K, hp = u / K, hy = hp / HW_, hx = hp % HW_;
      const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
      Xs[k][hp] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? a.A[(((long)nb * a.H + gy) * a.W + gx) * a.lda + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TH * TW / 64; ++i) {                     // 4 pixels per lane: rows py, py + 4, ...
      const int p = pl + 64 * i, py = p / TW, px = p % TW;
      f32x4 acc = bias4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int hp = (py + tap / 3) * HW_ + px + tap % 3;
        for (int k = 0; k < K; ++k) {
          const float x = Xs[k][hp];
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[(tap * 4 + k) * 16 + 4 * qd]);
          acc += wv * x;
        }
      }
      const int gy = ty * TH + py, gx = tx * TW + px;
      if (gy < a.H && gx < a.W && 4 * qd < a.N) {
        const long pix = ((long)nb * a.H + gy) * a.W + gx;
        *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + 4 * qd) = acc;
        s1 += acc; s2 += acc * acc;
      }
    }
  }
  if (a.stat_sum) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v1 = s1[e], v2 = s2[e];
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (lane < 4) { red[0][w][4 * lane + e] = v1; red[1][w][4 * lane + e] = v2; }
    }
    __syncthreads();
    if (tid < 16 && tid < a.N) {
      a.stat_sum[(long)tid * a.n_mblocks + blockIdx.x] = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
      a.stat_sq[(long)tid * a.n_mblocks + blockIdx.x] = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
    }
  }
}
// ---------------------------------------------------------------------------
// 3x3x3 convolution of a ONE-channel volume (the V-Net's first layer, 1 -> 16): the generic kernel pads Cin to 16
// and spends the MFMA time of a 16 -> 16 layer on 15/16 zeros.  Here the 27 taps ARE the reduction dimension
// (K = 27 -> 28): D[cout][voxel] = W[cout][tap] * im2col[tap][voxel], the im2col operand read straight from a
// three-plane halo tile in LDS (one float per lane and MFMA), the weights live in 7 registers per lane.
// Persistent workgroups over 16 x 16 tiles of a plane; a lane ends up with 4 consecutive output channels of one
// voxel (16-byte stores); per-channel (sum, sum of squares) partials for the BatchNorm that follows.
// ---------------------------------------------------------------------------
// DEPTH = 1: the same for a one-channel IMAGE (3 x 3 taps, K = 9 -> 12; the U-Net's first layer).
template <int DEPTH>
__global__ __launch_bounds__(256) void conv3d_image_kernel(IgemmArgs a) {
  constexpr int TH = 16, TW = 16, HW_ = TW + 2, NP = (TH + 2) * HW_, NT = 9 * DEPTH, KS = (NT + 3) / 4;
  __shared__ float Xs[DEPTH * NP];       // [depth tap][halo position]
  __shared__ float red[2][4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
  float wk[KS]; int off[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int t = 4 * s + g;                                     // tap = dz*9 + dy*3 + dx
    wk[s] = (t < NT && li < a.N) ? a.Wp[((long)t * a.Npad + li) * a.Kpad] : 0.f;
    off[s] = t < NT ? (t / 9) * NP + ((t % 9) / 3) * HW_ + t % 3 : 0;
  }
  f32x4 bias4 = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) if (a.bias && 4 * g + e < a.N) bias4[e] = a.bias[4 * g + e];
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long n_tiles = (long)a.NB * tiles_y * tiles_x;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1, bpg = gridDim.x / n_grp;      // one BN group per workgroup
  const long tpg = n_tiles / n_grp, t_end = (blockIdx.x / bpg + 1) * tpg;
  for (long t = (blockIdx.x / bpg) * tpg + blockIdx.x % bpg; t < t_end; t += bpg) {
    const int tx = t % tiles_x; const long r = t / tiles_x; const int ty = r % tiles_y; const int pl = r / tiles_y;
    const int xd = pl % a.D3;                                    // plane index inside its volume
    __syncthreads();
    for (int u = tid; u < DEPTH * NP; u += 256) {
      const int k = u / NP, hp = u - k * NP, hy = hp / HW_, hx = hp - hy * HW_;
      const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1, dk = k - DEPTH / 2, pz = xd + dk;
      Xs[u] = (pz >= 0 && pz < a.D3 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                  ? a.A[(((long)(pl + dk) * a.H + gy) * a.W + gx) * a.lda] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {                             // wave w: tile rows 4w .. 4w+3, lane li = column
      const int py = 4 * w + rg;
      const float* xb = Xs + py * HW_ + li;
      f32x4 acc = bias4;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[s], xb[off[s]], acc, 0, 0, 0);
      const int gy = ty * TH + py, gx = tx * TW + li;
      if (gy < a.H && gx < a.W && 4 * g < a.N) {
        const long pix = ((long)pl * a.H + gy) * a.W + gx;
        if (a.mma == 4) {      // f16 activation storage: round, store 8 bytes, statistics of the rounded values
          const f16x4 hv = to_f16x4(acc);
          *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.C) + pix * a.ldc + 4 * g) = hv;
          acc = __builtin_convertvector(hv, f32x4);
        } else {
          *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + 4 * g) = acc;
        }
        s1 += acc; s2 += acc * acc;
      }
    }
  }
  if (a.stat_sum) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v1 = s1[e], v2 = s2[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (li == 0) { red[0][w][4 * g + e] = v1; red[1][w][4 * g + e] = v2; }
    }
    __syncthreads();
    if (tid < 16 && tid < a.N) {
      a.stat_sum[(long)tid * a.n_mblocks + blockIdx.x] = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
      a.stat_sq[(long)tid * a.n_mblocks + blockIdx.x] = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
    }
  }
}
static bool image_conv3d_eligible(const IgemmArgs& a) {
  return a.K == 1 && a.Npad == 16 && (a.N & 3) == 0 && (a.ldc & 3) == 0 && a.R == nullptr && a.D3 >= 1;
}
template <int DEPTH>
static int launch_image_conv3d(const IgemmArgs& a, hipStream_t st, int* q) {
  const long n_tiles = (long)a.NB * ((a.H + 15) / 16) * ((a.W + 15) / 16);
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1;
  if (a.NB % n_grp != 0) return ARCO_ERR_ARG;
  long bpg = 2048 / n_grp; if (bpg > n_tiles / n_grp) bpg = n_tiles / n_grp; if (bpg < 1) bpg = 1;
  const long blocks = bpg * n_grp;
  if (q) { q[0] = (int)blocks; q[1] = 9 * DEPTH * 1000000 + 700000 + 1000 + 16; q[2] = 10; return ARCO_OK; }
  IgemmArgs b = a; b.n_mblocks = (int)blocks; b.n_nblocks = 1;
  if (DEPTH == 1) b.D3 = 1;
  hipLaunchKernelGGL(conv3d_image_kernel<DEPTH>, dim3((unsigned)blocks), dim3(256), 0, st, b);
  return arco_launch_status();
}
static bool image_conv_eligible(const IgemmArgs& a) {
  return a.K <= 4 && a.Npad == 16 && (a.N & 3) == 0 && (a.ldc & 3) == 0 && a.R == nullptr;
}
static int launch_image_conv(const IgemmArgs& a, hipStream_t st, int* q) {
  const long n_tiles = (long)a.NB * ((a.H + 15) / 16) * ((a.W + 15) / 16);
  const int n_grp = a.stat_groups > 1 ? a.stat_groups : 1;
  if (a.NB % n_grp != 0) return ARCO_ERR_ARG;
  long bpg = 1024 / n_grp; if (bpg > n_tiles / n_grp) bpg = n_tiles / n_grp; if (bpg < 1) bpg = 1;
  const long blocks = bpg * n_grp;
  if (q) { q[0] = (int)blocks; q[1] = 9 * 1000000 + 800000 + a.K * 1000 + 16; q[2] = 10; return ARCO_OK; }
  IgemmArgs b = a; b.n_mblocks = (int)blocks; b.n_nblocks = 1;
  hipLaunchKernelGGL(conv3x3_image_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
  return arco_launch_status();
}
// ---------------------------------------------------------------------------------------------------------------------------
// Streaming 1x1(x1) convolutions with a handful of channels on one side (round 6).  The V-Net's out_conv (16 -> classes,
// vnetWithArgs.py:182) and its data gradient (classes -> 16) run over the FULL-resolution map: 128 MB (LA) per launch for
// 0.06 GFLOP.  As GEMMs on 256 x 16 tiles (N padded 2 -> 16, K padded 16 -> 32, operands staged and split through LDS) they ran
// 85 us / 204 us - 1.5 / 0.6 TB/s; they are streams.  fp32 FMAs in a fixed order (deterministic; values differ from the MFMA
// route by fp32 rounding of a 16-term sum).
//   narrow-out (Q = K / 4 lanes per row, N <= 4): lane (row, quad) multiplies its input quad with its 4 N weights, the Q partial
//     sums meet in an xor butterfly, lane quad 0 writes the row's N outputs.  Weights: the split-bf16 pack (mma 3), summed back
//     to the exact fp32 value (the three terms are an exact decomposition), or the plain fp32 pack.
//   narrow-in (K <= 4, N = 4 Q outputs, plain fp32 pack): lane (row, quad) reads the row's K inputs and writes one 16-byte quad.
// ---------------------------------------------------------------------------------------------------------------------------
template <int Q>
__global__ __launch_bounds__(256) void conv1x1_narrow_out_kernel(IgemmArgs a) {
  const int lq = threadIdx.x % Q;
  float w[4][4], bs[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    bs[n] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * lq + e;
      float v = 0.f;
      if (n < a.N && k < a.K) {
        if (a.mma == 3) {
          const unsigned short* wp = reinterpret_cast<const unsigned short*>(a.Wp) + ((long)n * a.Kg + (k >> 4)) * 48 + (k & 15);
          const float b0 = __uint_as_float((unsigned)wp[0] << 16), b1 = __uint_as_float((unsigned)wp[16] << 16), b2 = __uint_as_float((unsigned)wp[32] << 16);
          v = (b0 + b1) + b2;
        } else {
          v = a.Wp[(long)n * a.Kpad + k];
        }
      }
      w[n][e] = v;
    }
  }
  const long rstride = (long)gridDim.x * (256 / Q);
  for (long row = (long)blockIdx.x * (256 / Q) + threadIdx.x / Q; row < a.M; row += rstride) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.A + row * a.lda + 4 * lq);
    float p[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) p[n] = fmaf(x[3], w[n][3], fmaf(x[2], w[n][2], fmaf(x[1], w[n][1], x[0] * w[n][0])));
#pragma unroll
    for (int off = 1; off < Q; off <<= 1)
#pragma unroll
      for (int n = 0; n < 4; ++n) p[n] += __shfl_xor(p[n], off);
    if (lq == 0) {
      float* o = a.C + row * a.ldc;
#pragma unroll
      for (int n = 0; n < 4; ++n) if (n < a.N) o[n] = p[n] + bs[n];
    }
  }
}
__global__ __launch_bounds__(256) void conv1x1_narrow_in_kernel(IgemmArgs a) {
  const int Q = a.N >> 2, lq = threadIdx.x % Q;
  f32x4 w[4], bq = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) w[k][e] = k < a.K ? a.Wp[(long)(4 * lq + e) * a.Kpad + k] : 0.f;
  if (a.bias) bq = *reinterpret_cast<const f32x4*>(a.bias + 4 * lq);
  const long rstride = (long)gridDim.x * (256 / Q);
  for (long row = (long)blockIdx.x * (256 / Q) + threadIdx.x / Q; row < a.M; row += rstride) {
    const float* x = a.A + row * a.lda;
    f32x4 y = bq;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < a.K) { const float xv = x[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaf(xv, w[k][e], y[e]); }
    if (a.R) y += *reinterpret_cast<const f32x4*>(a.R + row * a.ldr + 4 * lq);
    *reinterpret_cast<f32x4*>(a.C + row * a.ldc + 4 * lq) = y;
  }
}
// -1: not taken.  ARCO_CONV1X1_STREAM=0: off (A/B)
static int conv1x1_stream_dispatch(const IgemmArgs& a, hipStream_t st) {
  static const int on = getenv("ARCO_CONV1X1_STREAM") ? atoi(getenv("ARCO_CONV1X1_STREAM")) : 1;
  if (!on || a.stat_sum || a.pro.mean || a.Rup || a.ksplit > 1 || a.batch > 1 || a.M < 65536 || (a.mma != 3 && a.mma != 0)) return -1;
  const long rows_per_block_min = 8;
  if (!a.R && a.N >= 1 && a.N <= 4 && (a.K == 4 || a.K == 8 || a.K == 16 || a.K == 32) && (a.lda & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(a.A) & 15) == 0) {
    const int Q = a.K / 4;
    long blocks = (a.M + (256 / Q) * rows_per_block_min - 1) / ((256 / Q) * rows_per_block_min);
    if (blocks > 16384) blocks = 16384;
    switch (Q) {
      case 1: hipLaunchKernelGGL(conv1x1_narrow_out_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL(conv1x1_narrow_out_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      case 4: hipLaunchKernelGGL(conv1x1_narrow_out_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL(conv1x1_narrow_out_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a); break;
    }
    return arco_launch_status();
  }
  if (a.mma == 0 && a.K >= 1 && a.K <= 4 && (a.N == 4 || a.N == 8 || a.N == 16 || a.N == 32) && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) &&
      (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && (!a.R || (reinterpret_cast<uintptr_t>(a.R) & 15) == 0) &&
      (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0)) {
    const int Q = a.N / 4;
    long blocks = (a.M + (256 / Q) * rows_per_block_min - 1) / ((256 / Q) * rows_per_block_min);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(conv1x1_narrow_in_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return arco_launch_status();
  }
  return -1;
}

static int dispatch_igemm(const IgemmArgs& a, int taps, hipStream_t st, int* nmb) {
  if (taps == 1 && !nmb) { const int r = conv1x1_stream_dispatch(a, st); if (r != -1) return r; }
  if (taps == 1 && a.mma == 3) {      // split-bf16 GEMMs: K chunks of 32
    if (!nmb) {                       // the wide many-tile GEMMs: the software-pipelined kernel of gemm_sp.hip (launches only: the
      const int r = gemm_sp_dispatch(a, st, nullptr);      // tile queries describe igemm_kernel, which every statistics launch uses)
      if (r != -1) return r;
    }
    if (a.Npad <= 16) return launch_igemm<1, 256, 16, 4, 1, 32, false>(a, st, nmb);
    if (a.Npad <= 32) return launch_igemm<1, 128, 32, 4, 1, 32, false>(a, st, nmb);
    if (a.M * (long)a.Npad <= 4096l * 1024) {
      const long tiles64 = ((a.M + 63) / 64) * ((a.Npad + 63) / 64);
      if (tiles64 < 192 && a.stat_groups <= 1) return launch_igemm<1, 32, 64, 2, 2, 32, true>(a, st, nmb);
      return launch_igemm<1, 64, 64, 2, 2, 32, true>(a, st, nmb);
    }
    if ((a.Npad + 223) / 224 * 224 < (a.Npad + 127) / 128 * 128) {
      // N = 448 / 192 / 224 ...: 224-wide tiles pad N least, but the 64 x 224 split tile is single-buffered (50 KB of B
      // planes per chunk); A/B knob ARCO_GEMM224: 0 = 64 x 224 (round 2), 1 = 64 x 64, 2 = 128 x 64, 3 = 128 x 128
      static const int g224 = getenv("ARCO_GEMM224") ? atoi(getenv("ARCO_GEMM224")) : 0;
      if (g224 == 1 && (a.Npad & 63) == 0) return launch_igemm<1, 64, 64, 2, 2, 32, true>(a, st, nmb);
      if (g224 == 2 && (a.Npad & 63) == 0) return launch_igemm<1, 128, 64, 2, 2, 32, true>(a, st, nmb);
      if (g224 == 3) return launch_igemm<1, 128, 128, 2, 2, 32, true>(a, st, nmb);
      return launch_igemm<1, 64, 224, 2, 2, 32, false>(a, st, nmb);
    }
    return launch_igemm<1, 128, 128, 2, 2, 32, true>(a, st, nmb);
  }
  if (taps == 1) {
    if (a.Npad <= 16) return launch_igemm<1, 256, 16, 4, 1, 16, true>(a, st, nmb);
    if (a.Npad <= 32) return launch_igemm<1, 128, 32, 4, 1, 16, true>(a, st, nmb);
    if (a.M * (long)a.Npad <= 4096l * 1024) {
      // few-tile GEMMs (<= 1024 anchor rows of the row-sparse head, the 16 x 16 level): 32-row tiles double the workgroups
      const long tiles64 = ((a.M + 63) / 64) * ((a.Npad + 63) / 64);
      if (tiles64 < 192 && a.stat_groups <= 1) return launch_igemm<1, 32, 64, 2, 2, 32, true>(a, st, nmb);
      return launch_igemm<1, 64, 64, 2, 2, 32, true>(a, st, nmb);
    }
    // N-tile 224 (14 sub-tiles, 7 per wave column) when it pads N less than 128 does: 448 = 2 x 224 exactly
    // (64 x 224: 56 accumulator registers, 3 workgroups per CU; 128 x 224 needs 276 registers -> 1 wave per SIMD)
    if ((a.Npad + 223) / 224 * 224 < (a.Npad + 127) / 128 * 128) return launch_igemm<1, 64, 224, 2, 2, 16, true>(a, st, nmb);
    return launch_igemm<1, 128, 128, 2, 2, 32, true>(a, st, nmb);
  }
  if (taps == 27 && image_conv3d_eligible(a)) return launch_image_conv3d<3>(a, st, nmb);   // one-channel volume (first layer)
  if (taps == 9 && image_conv3d_eligible(a)) return launch_image_conv3d<1>(a, st, nmb);     // one-channel image
  if (taps == 27 && a.mma == 3) {     // the 16 -> 16 full-resolution level: resident weights, ring of three input planes (conv_sp.hip)
    const int r = conv3d_rw_dispatch(a, st, nmb);
    if (r != -1) return r;
  }
  if (taps == 27 && a.mma == 3) {     // the 32 .. 256-channel levels: pipelined flat-tile kernel over 3 K virtual channels (conv3d_fl.hip)
    const int r = conv3d_fl_dispatch(a, st, nmb);
    if (r != -1) return r;
  }
  if (taps == 27) return dispatch_spatial<3>(a, st, nmb);   // 3x3x3: planes of H x W, depth taps looped in the kernel
  if (taps == 9 && a.mma == 3) {     // 2-D levels with enough tiles: the software-pipelined kernels of conv_sp.hip
    const int r = conv_sp_dispatch(a, st, nmb);
    if (r != -1) return r;
  }
  if (taps == 9 && image_conv_eligible(a)) return a.mma == 3 ? ARCO_ERR_UNSUPPORTED : launch_image_conv(a, st, nmb);
  // split-bf16 launches (the caller asked arco_conv_split_ok) never take the fp32-only halo kernel
  if (taps == 9) return (a.mma != 3 && halo_eligible(a)) ? dispatch_halo(a, st, nmb) : dispatch_spatial<1>(a, st, nmb);
  return ARCO_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// weight packing: torch [Cout][Cin][kh][kw] -> Wp[tap][Npad][Kpad]
//   mode 0 (forward):  Wp[tap][co][ci] = W[co][ci][tap]
//   mode 1 (dgrad):    Wp[tap][ci][co] = W[co][ci][TAPS-1-tap]   (flipped, transposed)
// ---------------------------------------------------------------------------
//   mode | 2 ("split", the operand format of the MMA = 3 kernels): element (tap, n, k) is stored as its three bf16 terms
//   x0 + x1 + x2 at ((tap*Npad + n) * Kpad/16 + k/16) * 48 + p*16 + k%16 (bf16 units), Kpad = ceil32(K)
__device__ __forceinline__ void store_split3(float v, unsigned short* __restrict__ dst, long r, int k, int Kpad) {
  const __bf16 b0 = (__bf16)v; const float r1 = v - (float)b0;
  const __bf16 b1 = (__bf16)r1; const float r2 = r1 - (float)b1;
  const __bf16 b2 = (__bf16)r2;
  unsigned short* d = dst + (r * (Kpad / 16) + k / 16) * 48 + (k & 15);
  d[0] = __builtin_bit_cast(unsigned short, b0); d[16] = __builtin_bit_cast(unsigned short, b1);
  d[32] = __builtin_bit_cast(unsigned short, b2);
}
__global__ void pack_weight_kernel(const float* __restrict__ W, int Cout, int Cin, int taps, int mode, int Npad,
                                   int Kpad, float* __restrict__ Wp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)taps * Npad * Kpad;
  if (i >= tot) return;
  const int k = i % Kpad; const long r = i / Kpad; const int n = r % Npad; const int tap = r / Npad;
  float v = 0.f;
  if ((mode & 1) == 0) { if (n < Cout && k < Cin) v = W[((long)n * Cin + k) * taps + tap]; }
  else { if (n < Cin && k < Cout) v = W[((long)k * Cin + n) * taps + (taps - 1 - tap)]; }
  if (mode & 4) reinterpret_cast<_Float16*>(Wp)[i] = (_Float16)v;           // f16 pack (conv_h.hip): [tap][Npad][Kpad = ceil32(K)]
  else if (mode & 2) store_split3(v, reinterpret_cast<unsigned short*>(Wp), r, k, Kpad);
  else Wp[i] = v;
}

// out = sum over slabs (fixed order), float4 lanes
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ ws, int splits, long n4, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
  for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(ws)[(long)k * n4 + i];
  reinterpret_cast<f32x4*>(out)[i] = s;
}

// one launch packs every conv weight of a model: desc[i] = {src, dst, Cout, Cin, taps, mode, Npad, Kpad, first}
struct PackDesc { const float* src; float* dst; int Cout, Cin, taps, mode, Npad, Kpad; long first; };
// One launch for every weight of a network.  grid = (chunks, descriptors): block (x, y) works on descriptor y only - no search.
//  * taps > 1 (the 3x3 / 3x3x3 weights, ~95 % of the elements): 16 (n) x 16 (k) x taps tiles through LDS.  The torch layout has the
//    tap index fastest, the packed layout the k index: element-wise each wave load touched 64 different 128-byte lines for 256 useful
//    bytes (every weight is packed up to four times per step: plain / split, forward / transposed) - 127-132 us per launch for the
//    V-Net's 38 MB at 0.7 TB/s (profiles/r06_notes.md section 18).  A tile reads sixteen contiguous runs of 16 x taps floats and
//    writes, per tap, sixteen runs of 16 consecutive k.
//  * taps == 1 and the GEMM forms of the k2 s2 convolutions (mode >> 3): element-wise, 4096 elements per block step.
__global__ __launch_bounds__(256) void pack_many_kernel(const PackDesc* __restrict__ desc, int n_desc, long total) {
  __shared__ float tile[16 * (16 * 27 + 1)];
  const PackDesc d = desc[blockIdx.y];
  const int gm = d.mode >> 3;
  if (gm == 0 && d.taps > 1 && d.taps <= 27 && (d.Npad & 15) == 0 && (d.Kpad & 15) == 0) {
    const int T = d.taps, run = 16 * T, rs = run + 1;          // LDS row stride (odd distance between the rows of a tile)
    const int kt = d.Kpad >> 4, n_tiles = (d.Npad >> 4) * kt;
    const bool tr = (d.mode & 1) != 0;
    // logical source matrix [A][B][T] (T fastest): forward A = n (< Cout), B = k (< Cin); transposed A = k (< Cout), B = n (< Cin)
    const int An = d.Cout, Bn = d.Cin;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
      const int n0 = (t / kt) * 16, k0 = (t - (t / kt) * kt) * 16;
      const int a0 = tr ? k0 : n0, b0 = tr ? n0 : k0;
      __syncthreads();
      for (int e = threadIdx.x; e < 16 * run; e += 256) {        // row = a - a0, off = (b - b0) * T + tap: contiguous in the source
        const int row = e / run, off = e - row * run;
        const int a = a0 + row, b = b0 + off / T;
        tile[row * rs + off] = (a < An && b < Bn) ? d.src[((long)a * Bn + b0) * T + off] : 0.f;
      }
      __syncthreads();
      const int nl = threadIdx.x >> 4, kl = threadIdx.x & 15;
      const int n = n0 + nl, k = k0 + kl;
      for (int tap = 0; tap < T; ++tap) {
        const float v = tr ? tile[kl * rs + nl * T + (T - 1 - tap)] : tile[nl * rs + kl * T + tap];
        const long r = (long)tap * d.Npad + n;
        if (d.mode & 4) reinterpret_cast<_Float16*>(d.dst)[r * d.Kpad + k] = (_Float16)v;
        else if (d.mode & 2) store_split3(v, reinterpret_cast<unsigned short*>(d.dst), r, k, d.Kpad);
        else d.dst[r * d.Kpad + k] = v;
      }
    }
    return;
  }
  const long n_el = (long)(gm ? 1 : d.taps) * d.Npad * d.Kpad;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n_el; j += (long)gridDim.x * 256) {
    const int k = j % d.Kpad; const long r = j / d.Kpad; const int n = r % d.Npad; const int tap = r / d.Npad;
    float v = 0.f;
    if (gm) {
      // GEMM form of a k2 s2 (transposed) convolution, gathered straight from the torch layout (round 4: the V-Net's
      // DownsamplingConvBlock / UpsamplingDeconvBlock rebuilt and re-packed their [N][K] weights at every forward: ~130
      // launches per volume step).  Logical matrix W2[a][b] (a < Cout, b < Cin; one "tap"); d.taps carries the inner dim g:
      //   gm 1 (Conv3d [co][ci][8]):           W2[co][t * ci + c] = W[co][c][t],  g = ci
      //   gm 2 (ConvTranspose3d [ci][co][8]):  W2[t * co + o][ci] = W[ci][o][t],  g = co
      //   gm 4 (bias repeated over the 8 taps): W2[0][t * co + o] = bias[o],       g = co
      int a_, b_; bool ok;
      if ((d.mode & 1) == 0) { a_ = n; b_ = k; ok = n < d.Cout && k < d.Cin; }
      else { a_ = k; b_ = n; ok = n < d.Cin && k < d.Cout; }
      if (ok) {
        const int g = d.taps;
        if (gm == 1) v = d.src[((long)a_ * g + b_ % g) * 8 + b_ / g];
        else if (gm == 2) v = d.src[((long)b_ * g + a_ % g) * 8 + a_ / g];
        else v = d.src[b_ % g];
      }
    }
    else if ((d.mode & 1) == 0) { if (n < d.Cout && k < d.Cin) v = d.src[((long)n * d.Cin + k) * d.taps + tap]; }
    else { if (n < d.Cin && k < d.Cout) v = d.src[((long)k * d.Cin + n) * d.taps + (d.taps - 1 - tap)]; }
    if (d.mode & 4) reinterpret_cast<_Float16*>(d.dst)[j] = (_Float16)v;
    else if (d.mode & 2) store_split3(v, reinterpret_cast<unsigned short*>(d.dst), r, k, d.Kpad);
    else d.dst[j] = v;
  }
}

// ---------------------------------------------------------------------------
// weight gradient:  dW[co][ci][tap] = sum_pix dZ[pix][co] * Ain[pix + tap][ci]
// M = co, N = ci, K = pixels.  grid.x = pixel chunk, grid.y = co tile,
// grid.z = tap * ci_tiles + ci tile.  Tiles of 128 pixels (8x16) are staged channels-last in
// LDS (row stride = 16 mod 32 dwords -> conflict-free ds_read_b32 of 16 channels x 4
// pixels); each wave reduces 32 pixels per tile; partial slabs are summed in fixed
// order by wgrad_reduce_kernel (deterministic).
// ---------------------------------------------------------------------------
// ablation bits of tools/micro/wgrad_abl.py / wgrad1_bench.py: compiled in only with -DARCO_WGRAD_ABLATION (as run-time
// branches they cost wgrad_split_kernel<16,16> 8 VGPRs and 56 % of its speed)
#ifdef ARCO_WGRAD_ABLATION
#define WGRAD_ABL(a) ((a).abl)
#else
#define WGRAD_ABL(a) 0
#endif
struct WgradArgs {
  const float* dZ; long ldz; int Cout;
  const float* Ain; long lda; int Cin;
  int taps, NB, H, W, D3; long M;
  float* partial;      // [chunks][taps][CoutPad][CinPad]
  int CoutPad, CinPad, n_tiles;
  int mma;             // 0 fp32 MFMA; 2: bf16 operands (halo kernels, 3x3x3 only), fp32 accumulate
  int abl;             // ablation bits for tools/micro/wgrad_abl.py (0 in the product): 1 no MFMA phase, 2 no staging, 4 no global loads
  ArcoActPro pro;      // pro.mean != nullptr: Ain holds the pre-activation z of the producing conv; the loader applies BN + LeakyReLU + dropout
                       // (wgrad_split_kernel, 3x3: the weight gradient of a block's second convolution reads z1, not a stored activation)
};

constexpr int WGRAD1_PAD = 8;
template <int CO_B, int CI_B>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  // row pad 8 (was 16): 64 x 64 blocks take 73.7 KB instead of 82 KB of LDS - TWO workgroups per CU, so one stages while the
  // other is on the matrix cores (the staging loads are not prefetched); the fragment reads become 2-way bank conflicts,
  // 8 reads against 16 MFMAs per K step
  constexpr int LDZ = (CO_B % 32 == 0) ? CO_B + WGRAD1_PAD : CO_B;
  constexpr int LDA = (CI_B % 32 == 0) ? CI_B + WGRAD1_PAD : CI_B;
  constexpr int CO_T = CO_B / 16, CI_T = CI_B / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zs = smem;                 // [128][LDZ]
  float* Xs = smem + 128 * LDZ;     // [128][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int ci_tiles = a.CinPad / CI_B;
  const int tap = blockIdx.z / ci_tiles, cit = blockIdx.z % ci_tiles;
  const int co0 = blockIdx.y * CO_B, ci0 = cit * CI_B;
  const bool sp = a.taps >= 9;                     // spatial taps (3x3 per plane, x3 planes when taps == 27)
  const int t9 = tap % 9, dpl = a.taps == 27 ? tap / 9 - 1 : 0;
  const int dy = sp ? t9 / 3 - 1 : 0, dx = sp ? t9 % 3 - 1 : 0;
  const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;

  f32x4 acc[CO_T][CI_T];
#pragma unroll
  for (int i = 0; i < CO_T; ++i)
#pragma unroll
    for (int j = 0; j < CI_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  // the next tile's operands travel global -> registers while this tile is on the matrix cores (fetch), registers -> LDS
  // behind the barrier (stage): unprefetched staging loads were 100 of the 180 us of the 16 384 x 448 x 448 head gradient
  constexpr int NZ = 128 * (CO_B / 4) / 256, NX = 128 * (CI_B / 4) / 256;
  f32x4 rz[NZ], rx[NX];
  auto fetch = [&](int t) {
    int img = 0, y0 = 0, x0 = 0; long m0 = 0;
    if (sp) {
      int tt = t; const int tx = tt % tiles_x; tt /= tiles_x; const int ty = tt % tiles_y; img = tt / tiles_y;
      y0 = ty * 8; x0 = tx * 16;
    } else {
      m0 = (long)t * 128;
    }
#pragma unroll
    for (int it = 0; it < NZ; ++it) {
      const int idx = tid + it * 256;
      const int p = idx / (CO_B / 4), q = idx % (CO_B / 4);
      long pix = -1;
      if (sp) { const int y = y0 + p / 16, x = x0 + p % 16; if (y < a.H && x < a.W) pix = ((long)img * a.H + y) * a.W + x; }
      else { const long m = m0 + p; if (m < a.M) pix = m; }
      f32x4 v = f32x4{0, 0, 0, 0};
      if (pix >= 0) {
        const int c = co0 + 4 * q;
        const float* src = a.dZ + pix * a.ldz + c;
        if (((a.Cout & 3) == 0) && ((a.ldz & 3) == 0)) { if (c < a.Cout) v = *reinterpret_cast<const f32x4*>(src); }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.Cout) v[e] = src[e];
        }
      }
      rz[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = tid + it * 256;
      const int p = idx / (CI_B / 4), q = idx % (CI_B / 4);
      long pix = -1;
      if (sp) {
        const int yo = y0 + p / 16, xo = x0 + p % 16;
        const int y = yo + dy, x = xo + dx;
        const int pl = a.taps == 27 ? img % a.D3 + dpl : 0;
        if (yo < a.H && xo < a.W && y >= 0 && y < a.H && x >= 0 && x < a.W && pl >= 0 && pl < a.D3)
          pix = ((long)(img + dpl) * a.H + y) * a.W + x;
      } else { const long m = m0 + p; if (m < a.M) pix = m; }
      f32x4 v = f32x4{0, 0, 0, 0};
      if (pix >= 0) {
        const int c = ci0 + 4 * q;
        const float* src = a.Ain + pix * a.lda + c;
        if (((a.Cin & 3) == 0) && ((a.lda & 3) == 0)) { if (c < a.Cin) v = *reinterpret_cast<const f32x4*>(src); }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.Cin) v[e] = src[e];
        }
      }
      rx[it] = v;
    }
  };
  int t = blockIdx.x;
  if (t < a.n_tiles) fetch(t);
  for (; t < a.n_tiles; t += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NZ; ++it) {
      const int idx = tid + it * 256;
      *reinterpret_cast<f32x4*>(&Zs[(idx / (CO_B / 4)) * LDZ + 4 * (idx % (CO_B / 4))]) = rz[it];
    }
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = tid + it * 256;
      *reinterpret_cast<f32x4*>(&Xs[(idx / (CI_B / 4)) * LDA + 4 * (idx % (CI_B / 4))]) = rx[it];
    }
    __syncthreads();
    if (t + (int)gridDim.x < a.n_tiles) fetch(t + gridDim.x);
    const int pw = wid * 32;
#pragma unroll 4
    for (int ks = 0; ks < 8; ++ks) {
      const int p = pw + ks * 4 + g;
      float zf[CO_T], xf[CI_T];
#pragma unroll
      for (int i = 0; i < CO_T; ++i) zf[i] = Zs[p * LDZ + i * 16 + li];
#pragma unroll
      for (int j = 0; j < CI_T; ++j) xf[j] = Xs[p * LDA + j * 16 + li];
#pragma unroll
      for (int i = 0; i < CO_T; ++i)
#pragma unroll
        for (int j = 0; j < CI_T; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  // cross-wave reduction through LDS, then slab store
  __syncthreads();
  float* red = smem;   // [4][CO_B][CI_B]
#pragma unroll
  for (int i = 0; i < CO_T; ++i)
#pragma unroll
    for (int j = 0; j < CI_T; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((wid * CO_B) + i * 16 + 4 * g + r) * CI_B + j * 16 + li] = acc[i][j][r];
  __syncthreads();
  float* out = a.partial + (((long)blockIdx.x * a.taps + tap) * a.CoutPad) * a.CinPad;
  for (int idx = tid; idx < CO_B * CI_B; idx += 256) {
    const int co = idx / CI_B, ci = idx % CI_B;
    const float v = (red[(0 * CO_B + co) * CI_B + ci] + red[(1 * CO_B + co) * CI_B + ci]) +
                    (red[(2 * CO_B + co) * CI_B + ci] + red[(3 * CO_B + co) * CI_B + ci]);
    out[(long)(co0 + co) * a.CinPad + ci0 + ci] = v;
  }
}

// ---------------------------------------------------------------------------
// 1x1 weight gradient of the WIDE heads (FeatureExtractor fea3 / fea4, q_representation: model_2D.py:46-53, train_arco_2d.py:231-234;
// dW [496 x 496] = dY^T [496 x M] . X [M x 496] at M = 5 x 10^5): 128 x 128 output blocks, each WAVE owns a 64 x 64 quadrant for
// all 128 pixels of a tile.  wgrad_kernel<64,64> gives every wave the whole 64 x 64 block for a quarter of the pixels: its
// 8 x 8 = 64 blocks read each operand eight times (16.6 GB at M = 524 288: 53 TFLOP/s, bound by L2 / HBM traffic); here 4 x 4
// blocks read each operand four times, no cross-wave reduction, one workgroup per CU (139 KB of LDS), the next tile's 128 KB
// travelling global -> registers under the current tile's 512 fp32 MFMAs per wave.  Same products and per-block summation
// order over pixels as wgrad_kernel within a slab; slabs are summed in fixed order by wgrad_reduce_kernel.
// ---------------------------------------------------------------------------
template <int TP>       // pixels per staged tile: 128 (139 KB of LDS, one workgroup per CU) or 64 (70 KB, two per CU)
__global__ __launch_bounds__(256) void wgrad_q_kernel(WgradArgs a) {
  constexpr int CO_B = 128, CI_B = 128, LDZ = CO_B + WGRAD1_PAD, LDA = CI_B + WGRAD1_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zs = smem;                 // [TP][LDZ]
  float* Xs = smem + TP * LDZ;      // [TP][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int wr = wid >> 1, wc = wid & 1;
  const int co0 = blockIdx.y * CO_B, ci0 = blockIdx.z * CI_B;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  constexpr int NP = TP * (128 / 4) / 256;           // 16-byte pieces per thread, tile and operand (16 / 8)
  f32x4 rz[NP], rx[NP];
  auto fetch = [&](int t) {
    const long m0 = (long)t * TP;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int idx = tid + it * 256, p = idx >> 5, q = idx & 31;
      const long m = m0 + p;
      const int cz = co0 + 4 * q, cx = ci0 + 4 * q;
      // clamped addresses + a select: branch-free loads (a conditional load puts a vmcnt(0) behind every piece)
      const bool okz = m < a.M && cz < a.Cout, okx = m < a.M && cx < a.Cin;
      const f32x4 vz = *reinterpret_cast<const f32x4*>(okz ? a.dZ + m * a.ldz + cz : a.dZ);
      const f32x4 vx = *reinterpret_cast<const f32x4*>(okx ? a.Ain + m * a.lda + cx : a.Ain);
      rz[it] = okz ? vz : f32x4{0, 0, 0, 0};
      rx[it] = okx ? vx : f32x4{0, 0, 0, 0};
    }
  };
  int t = blockIdx.x;
  if (t < a.n_tiles) fetch(t);
  for (; t < a.n_tiles; t += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int idx = tid + it * 256, p = idx >> 5, q = idx & 31;
      *reinterpret_cast<f32x4*>(&Zs[p * LDZ + 4 * q]) = rz[it];
      *reinterpret_cast<f32x4*>(&Xs[p * LDA + 4 * q]) = rx[it];
    }
    __syncthreads();
    if (t + (int)gridDim.x < a.n_tiles) fetch(t + gridDim.x);
#pragma unroll 4
    for (int ks = 0; ks < TP / 4; ++ks) {
      const int p = ks * 4 + g;
      float zf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) zf[i] = Zs[p * LDZ + (wr * 4 + i) * 16 + li];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = Xs[p * LDA + (wc * 4 + j) * 16 + li];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  float* out = a.partial + ((long)blockIdx.x * a.CoutPad) * a.CinPad;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(long)(co0 + (wr * 4 + i) * 16 + 4 * g + r) * a.CinPad + ci0 + (wc * 4 + j) * 16 + li] = acc[i][j][r];
}

// ---------------------------------------------------------------------------
// all-taps weight gradient for the shallow layers (Cout, Cin <= 32): one block owns a tile of 8x16 output
// pixels of one plane, stages the dZ tile and the input HALO patch once in LDS and accumulates all 9 taps of
// the plane (3-D: the depth tap dd = blockIdx.z, input plane x+dd-1) -> both operands are read from HBM/L2
// once instead of once per tap.  The A fragment (dZ^T) is shared by the 9 taps.
// partial layout identical to wgrad_kernel: [chunk][tap][CoutPad][CinPad].
// ---------------------------------------------------------------------------
// Persistent over pixel tiles with the NEXT tile's operands prefetched into
// registers during the MFMAs.  The block's CO_B x CI_B outputs are split into 16x16 sub-tiles; each sub-tile is
// owned by 4 / n_sub waves for all 9 taps (32x32: one wave per sub-tile walking all 128 pixels -> no cross-wave
// reduction, 36 accumulator registers instead of 144; 16x32 / 32x16: two waves per sub-tile, 64 pixels each;
// 16x16: four waves, 32 pixels each), partners are summed once per LAUNCH through LDS.
// FLAT: tiles are 128 consecutive positions of the plane stored with padded row stride W + 2 (see igemm_kernel FLAT)
// MMA = 2: 16 pixels per step instead of 4 - a lane converts 4 consecutive pixels of dZ and of each tap's input
// to bf16 and issues one v_mfma_f32_16x16x16_bf16 per tap (see igemm_kernel's MMA note; gradients use bf16, not f16:
// dZ values of 1e-6 would flush in f16).
template <int CO_B, int CI_B, bool FLAT = false, int MMA = 0>
__global__ __launch_bounds__(256) void wgrad_halo2_kernel(WgradArgs a) {
  constexpr int LDZ = (CO_B % 32 == 0) ? CO_B + 16 : CO_B;
  constexpr int LDA = (CI_B % 32 == 0) ? CI_B + 16 : CI_B;
  constexpr int QZ = CO_B / 4, QA = CI_B / 4, HROWS = FLAT ? 128 + 2 * IGEMM_FLAT_WPMAX + 2 : 10 * 18;
  constexpr int NZ = (128 * QZ + 255) / 256, NX = (HROWS * QA + 255) / 256;
  constexpr int CI_T = CI_B / 16, NSUB = (CO_B / 16) * CI_T, WPS = 4 / NSUB, KSTEPS = 32 / WPS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Zs = smem;                 // [128][LDZ]
  float* Xs = smem + 128 * LDZ;     // [180][LDA]   halo patch
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int sub = wid / WPS, part = wid % WPS, wi = sub / CI_T, wj = sub % CI_T;
  const int dd = blockIdx.z;                                   // depth tap (3-D) ; 0 for 2-D
  const int dpl = a.taps == 27 ? dd - 1 : 0;
  const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
  const int Wp = a.W + 2, flat_per_plane = (a.H * Wp + 127) / 128;
  const int ci_tiles = a.CinPad / CI_B;
  const int co0 = (blockIdx.y / ci_tiles) * CO_B, ci0 = (blockIdx.y % ci_tiles) * CI_B;
  const bool vz = ((a.Cout & 3) == 0) && ((a.ldz & 3) == 0), vx = ((a.Cin & 3) == 0) && ((a.lda & 3) == 0);

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  f32x4 pz[NZ], px_[NX];
  auto fetch = [&](int t) {
    int tt = t; const int tx = tt % tiles_x; tt /= tiles_x; const int ty = tt % tiles_y;
    const int img = FLAT ? t / flat_per_plane : tt / tiles_y;
    const int y0 = ty * 8, x0 = tx * 16, f0 = FLAT ? (t - img * flat_per_plane) * 128 : 0;
    const int pl = a.taps == 27 ? img % a.D3 + dpl : 0;
    const bool plane_ok = pl >= 0 && pl < a.D3;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int idx = tid + i * 256, p = idx / QZ, q = idx % QZ;
      const int fy = (f0 + p) / Wp, fx = (f0 + p) - fy * Wp;                  // FLAT: output position -> (row, padded column)
      const int y = FLAT ? fy : y0 + p / 16, x = FLAT ? fx - 1 : x0 + p % 16, c = co0 + 4 * q;
      f32x4 v = f32x4{0, 0, 0, 0};
      if (idx < 128 * QZ && y < a.H && x >= 0 && x < a.W && plane_ok) {
        const float* src = a.dZ + (((long)img * a.H + y) * a.W + x) * a.ldz + c;
        if (vz) { if (c < a.Cout) v = *reinterpret_cast<const f32x4*>(src); }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.Cout) v[e] = src[e];
        }
      }
      pz[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + i * 256, r = idx / QA, q = idx % QA;
      const int pidx = f0 + r - 1, py = pidx >= 0 ? pidx / Wp : -1, ppx = pidx - py * Wp;      // FLAT: padded-plane position
      const int y = FLAT ? py - 1 : y0 + r / 18 - 1, x = FLAT ? ppx - 1 : x0 + r % 18 - 1, c = ci0 + 4 * q;
      f32x4 v = f32x4{0, 0, 0, 0};
      if (idx < HROWS * QA && (!FLAT || (pidx >= 0 && r < 128 + 2 * Wp + 2)) && y >= 0 && y < a.H && x >= 0 && x < a.W && plane_ok) {
        const float* src = a.Ain + (((long)(img + dpl) * a.H + y) * a.W + x) * a.lda + c;
        if (vx) { if (c < a.Cin) v = *reinterpret_cast<const f32x4*>(src); }
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < a.Cin) v[e] = src[e];
        }
      }
      px_[i] = v;
    }
  };

  int t = blockIdx.x;
  if (t < a.n_tiles) fetch(t);
  while (t < a.n_tiles) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) { const int idx = tid + i * 256; if (idx < 128 * QZ) *reinterpret_cast<f32x4*>(&Zs[(idx / QZ) * LDZ + 4 * (idx % QZ)]) = pz[i]; }
#pragma unroll
    for (int i = 0; i < NX; ++i) { const int idx = tid + i * 256; if (idx < HROWS * QA) *reinterpret_cast<f32x4*>(&Xs[(idx / QA) * LDA + 4 * (idx % QA)]) = px_[i]; }
    __syncthreads();
    const int next = t + gridDim.x;
    if (next < a.n_tiles) fetch(next);
    if constexpr (MMA == 2) {
#pragma unroll 2
      for (int k4 = 0; k4 < KSTEPS / 4; ++k4) {
        const int p0 = (part * KSTEPS + 4 * k4) * 4 + 4 * g, py = p0 >> 4, pxx = p0 & 15;   // pixels p0 .. p0+3: one tile row
        f32x4 zv;
#pragma unroll
        for (int j = 0; j < 4; ++j) zv[j] = Zs[(p0 + j) * LDZ + wi * 16 + li];
        const s16x4 zh = to_bf16x4(zv);
        const float* xrow = Xs + (FLAT ? p0 : py * 18 + pxx) * LDA + wj * 16 + li;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int toff = (FLAT ? (tap / 3) * Wp + tap % 3 : (tap / 3) * 18 + tap % 3) * LDA;
          f32x4 xv;
#pragma unroll
          for (int j = 0; j < 4; ++j) xv[j] = xrow[toff + j * LDA];
          acc[tap] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(zh, to_bf16x4(xv), acc[tap], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 4
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int p = (part * KSTEPS + ks) * 4 + g, py = p >> 4, pxx = p & 15;
      const float zf = Zs[p * LDZ + wi * 16 + li];
      const float* xrow = Xs + (FLAT ? p : py * 18 + pxx) * LDA + wj * 16 + li;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(zf, xrow[(FLAT ? (tap / 3) * Wp + tap % 3 : (tap / 3) * 18 + tap % 3) * LDA], acc[tap], 0, 0, 0);
    }
    }
    __syncthreads();
    t = next;
  }
  if (WPS > 1) {                      // sum the pixel-split partner waves (once per launch)
    float* red = smem;                // [4 waves][9][256]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid * 9 + tap) * 256 + r * 64 + lane] = acc[tap][r];
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[tap][r];
#pragma unroll
          for (int w2 = 1; w2 < WPS; ++w2) v += red[((wid + w2) * 9 + tap) * 256 + r * 64 + lane];
          acc[tap][r] = v;
        }
    }
  }
  if (part == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float* out = a.partial + (((long)blockIdx.x * a.taps + dd * 9 + tap) * a.CoutPad) * a.CinPad;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(long)(co0 + wi * 16 + 4 * g + r) * a.CinPad + ci0 + wj * 16 + li] = acc[tap][r];
    }
  }
}

// ---------------------------------------------------------------------------
// Split-bf16 weight gradient of the 3x3 / 3x3x3 convolutions (the MMA = 3 counterpart of wgrad_halo2_kernel):
//   dW[tap][co][ci] = sum_pix dZ[pix][co] * X[pix + tap][ci],   M = co, N = ci, K = PIXELS.
// v_mfma_f32_16x16x32_bf16 wants 8 consecutive k (= pixels) per lane, so both operands are staged TRANSPOSED in LDS, as
// three bf16 planes [plane][channel][pixel]: a staging thread loads 4 consecutive pixels x 4 channels (four 16-byte global
// loads), splits every value into its three bf16 terms and writes, per channel and plane, the 4 pixels as one 8-byte word.
// One K step = 32 pixels = two rows of the 8 x 16 tile; lane (li, g) supplies channel li, row 2s + (g >> 1), columns
// 8 (g & 1) .. + 7.  The input operand of tap (dy, dx) is the halo tile shifted by dx columns: a lane reads the 12 columns
// 8h .. 8h + 11 of row r + dy once (ds_read_b128 + ds_read_b64) and forms the three dx windows in registers (dx = 1: four
// v_alignbit_b32, dx = 2: the next dwords) - 18 LDS reads feed the 54 MFMAs of a K step.  Persistent over tiles with the
// next tile's global loads in flight during the MFMAs; partial layout / reduction identical to wgrad_halo2_kernel.
// ---------------------------------------------------------------------------
// LDS channel rows of wgrad_split_kernel.  PMC (tools/pmc_lds_step.sh): with plain rows of 68 / 124 dwords TWO THIRDS of the kernel's
// LDS cycles were bank conflicts - the staging stores (8 lanes = 8 channel quads of one pixel group, channel stride 4 rows = 16 mod 32
// banks: 4-way) and the fragment reads (2-way).  The dword offset inside a channel's row is XOR-ed with 4 * (channel / 4): the eight
// quads of a store group land in eight different 4-dword blocks, and with rows of 72 / 136 dwords the ds_read_b128 groups are
// conflict-free as well (model: tools/micro/lds_bank_model.py - stores 4 -> 1.0 / 1.2, b128 reads 2 -> 1.0 / 1.5 cycles per group).
// ARCO_WG_SWZ=0 + ARCO_WG_CSZ=68 + ARCO_WG_CSX=124 rebuilds round 2's layout.
#ifndef ARCO_WG_SWZ
#define ARCO_WG_SWZ 1
#endif
#ifndef ARCO_WG_CSZ
#define ARCO_WG_CSZ (ARCO_WG_SWZ ? 72 : 68)
#endif
#ifndef ARCO_WG_CSX
#define ARCO_WG_CSX (ARCO_WG_SWZ ? 136 : 124)
#endif
template <int CO_B, int CI_B, bool PRO = false>
__global__ __launch_bounds__(256) void wgrad_split_kernel(WgradArgs a) {
  constexpr int QZ = CO_B / 4, QA = CI_B / 4;
  constexpr int ZU = 32 * QZ, XU = 50 * QA;                 // staging units (4 pixels x 4 channels)
  constexpr int NZU = (ZU + 255) / 256, NXU = (XU + 255) / 256;
  constexpr int CSZ = ARCO_WG_CSZ, CSX = ARCO_WG_CSX;      // dwords per (plane, channel) row: 128 px / 10 x 24 halo px (+ pad)
  constexpr int CI_T = CI_B / 16, NSUB = (CO_B / 16) * CI_T, WPS = 4 / NSUB, KS = 4 / WPS;
  extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
  unsigned int* Zs = smem_u;                                // [3][CO_B][CSZ]
  unsigned int* Xs = smem_u + 3 * CO_B * CSZ;               // [3][CI_B][CSX]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int sub = wid / WPS, part = wid % WPS, wi = sub / CI_T, wj = sub % CI_T;
  const int dd = blockIdx.z;
  const int dpl = a.taps == 27 ? dd - 1 : 0;
  const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 7) / 8;
  const int ci_tiles = a.CinPad / CI_B;
  const int co0 = (blockIdx.y / ci_tiles) * CO_B, ci0 = (blockIdx.y % ci_tiles) * CI_B;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  f32x4 pz[NZU][4], px_[NXU][4];
  // consumer-side activation of the input operand (PRO): pixel origin of the fetched tile, the parameter quads of
  // this thread's channels for the BatchNorm group of the tile being staged (reloaded when the group changes: at most once per launch
  // and workgroup - tiles are walked in image order)
  int f_img = 0, f_y0 = 0, f_x0 = 0, pgrp = -1;
  ProQuad pq[NXU];
  const int ipg = PRO ? a.NB / (a.pro.groups > 1 ? a.pro.groups : 1) : 1;
  const bool pdrop = PRO && a.pro.drop_mode == 1;
  uint32_t dkey = 0, dthr = 0; float keep_scale = 1.f;
  if (PRO && pdrop) {
    const unsigned long long sd = a.pro.seed_dev ? a.pro.seed ^ (a.pro.seed_dev[0] * 0x9E3779B97F4A7C15ull) : a.pro.seed;
    dkey = drop_key32(sd); dthr = drop_thr16(a.pro.p); keep_scale = 1.0f / (1.0f - a.pro.p);
  }
  auto fetch = [&](int t) {
    int tt = t; const int tx = tt % tiles_x; tt /= tiles_x; const int ty = tt % tiles_y; const int img = tt / tiles_y;
    const int y0 = ty * 8, x0 = tx * 16;
    if (PRO) { f_img = img; f_y0 = y0; f_x0 = x0; }
    const int pl = a.taps == 27 ? img % a.D3 + dpl : 0;
    const bool plane_ok = pl >= 0 && pl < a.D3;
#pragma unroll
    for (int i = 0; i < NZU; ++i) {
      const int u = tid + i * 256, pg = u / QZ, q = u % QZ, r = pg >> 2, cg = pg & 3;
      const int y = y0 + r, c = co0 + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + 4 * cg + j;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (u < ZU && y < a.H && x < a.W && plane_ok && c < a.Cout) v = *reinterpret_cast<const f32x4*>(a.dZ + (((long)img * a.H + y) * a.W + x) * a.ldz + c);
        pz[i][j] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
      const int u = tid + i * 256, pg = u / QA, q = u % QA, hr = pg / 5, hg = pg % 5;
      const int y = y0 + hr - 1, c = ci0 + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 - 1 + 4 * hg + j;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (u < XU && y >= 0 && y < a.H && x >= 0 && x < a.W && plane_ok && c < a.Cin) {
          v = *reinterpret_cast<const f32x4*>(a.Ain + (((long)(img + dpl) * a.H + y) * a.W + x) * a.lda + c);
        }
        px_[i][j] = v;
      }
    }
  };
  // 4 pixels x 4 channels -> per channel and plane one 8-byte word of 4 bf16 pixels
  auto stage = [&](const f32x4 (&v)[4], unsigned int* base, int nchan, int chan0, int cs, int off) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u32x2 p0, p1, p2;
      split3_bf16x4(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, p0, p1, p2);
      unsigned int* d = base + (long)(chan0 + e) * cs + (ARCO_WG_SWZ ? (off ^ (((chan0 >> 2) & 7) << 2)) : off);
      *reinterpret_cast<u32x2_ma*>(d) = p0;
      *reinterpret_cast<u32x2_ma*>(d + (long)nchan * cs) = p1;
      *reinterpret_cast<u32x2_ma*>(d + 2l * nchan * cs) = p2;
    }
  };

  const int swz_z = ARCO_WG_SWZ ? (((wi * 16 + li) >> 2) & 7) << 2 : 0, swz_x = ARCO_WG_SWZ ? (((wj * 16 + li) >> 2) & 7) << 2 : 0;
  int t = blockIdx.x;
  if (t < a.n_tiles && !(WGRAD_ABL(a) & 4)) fetch(t);
  while (t < a.n_tiles) {
    if (!(WGRAD_ABL(a) & 2)) {
#pragma unroll
    for (int i = 0; i < NZU; ++i) {
      const int u = tid + i * 256, pg = u / QZ, q = u % QZ, r = pg >> 2, cg = pg & 3;
      if (u < ZU) stage(pz[i], Zs, CO_B, 4 * q, CSZ, r * 8 + cg * 2);
    }
    if constexpr (PRO) {        // a = dropout(lrelu(BN(z))) on the fetched quads (zero padding stays zero), as bn_act_fwd_kernel computes it
      const int grp = f_img / ipg;
      if (grp != pgrp) {
        pgrp = grp;
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
          const int u = tid + i * 256, q = u % QA;
          int c = ci0 + 4 * q; if (c + 4 > a.Cin) c = 0;
          pq[i].mu = *reinterpret_cast<const f32x4*>(a.pro.mean + (long)grp * a.Cin + c);
          pq[i].is = *reinterpret_cast<const f32x4*>(a.pro.istd + (long)grp * a.Cin + c);
          pq[i].ga = *reinterpret_cast<const f32x4*>(a.pro.gamma + c);
          pq[i].be = *reinterpret_cast<const f32x4*>(a.pro.beta + c);
        }
      }
      // Tiles whose halo lies inside the image (all but the image's border ring) need no zeroing: the arithmetic alone costs next to
      // nothing, arithmetic + per-pixel zeroing 40 us on the 16 -> 16 layer at 256^2 (tools/micro/wgrad_pro_bench.py) - wave-uniform branch
      const bool interior = f_y0 > 0 && f_y0 + 9 <= a.H && f_x0 > 0 && f_x0 + 17 <= a.W && (a.Cin % CI_B) == 0;
      if (interior) {
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
          const int u = tid + i * 256, pg = u / QA, q = u % QA, hr = pg / 5, hg = pg % 5;
          const uint32_t e0 = (uint32_t)(((f_img * a.H + f_y0 + hr - 1) * a.W + f_x0 - 1 + 4 * hg) * a.Cin + ci0 + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 v = pro_bn_lrelu(px_[i][j], pq[i], a.pro.slope);
            if (pdrop) v = pro_dropout(v, dkey, e0 + (uint32_t)(j * a.Cin), dthr, keep_scale);
            px_[i][j] = v;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
          const int u = tid + i * 256, pg = u / QA, q = u % QA, hr = pg / 5, hg = pg % 5;
          const uint32_t e0 = (uint32_t)(((f_img * a.H + f_y0 + hr - 1) * a.W + f_x0 - 1 + 4 * hg) * a.Cin + ci0 + 4 * q);
          const int y = f_y0 + hr - 1;
          const bool rok = u < XU && y >= 0 && y < a.H && ci0 + 4 * q < a.Cin;      // (validity recomputed here, on border tiles only: no mask carried from fetch)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int x = f_x0 - 1 + 4 * hg + j;
            f32x4 v = pro_bn_lrelu(px_[i][j], pq[i], a.pro.slope);
            if (pdrop) v = pro_dropout(v, dkey, e0 + (uint32_t)(j * a.Cin), dthr, keep_scale);
            px_[i][j] = pro_mask(v, (rok && x >= 0 && x < a.W) ? 1u : 0u);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
      const int u = tid + i * 256, pg = u / QA, q = u % QA, hr = pg / 5, hg = pg % 5;
      if (u < XU) stage(px_[i], Xs, CI_B, 4 * q, CSX, hr * 12 + hg * 2);
    }
    }
    __syncthreads();
    const int next = t + gridDim.x;
    if (next < a.n_tiles && !(WGRAD_ABL(a) & 4)) fetch(next);
    if (!(WGRAD_ABL(a) & 1))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int s = part * KS + ks, rr = 2 * s + (g >> 1), h = g & 1;
      bf16x8 za[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) za[p] = lds_bf16x8(&Zs[(p * CO_B + wi * 16 + li) * CSZ + ((rr * 8 + h * 4) ^ swz_z)]);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        unsigned int w[3][6];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const unsigned int* row = &Xs[(p * CI_B + wj * 16 + li) * CSX];
          const int o = (rr + dy) * 12 + h * 4;
          const u32x4 lo = *reinterpret_cast<const u32x4_ma*>(row + (o ^ swz_x));
          const u32x2 hi = *reinterpret_cast<const u32x2_ma*>(row + ((o + 4) ^ swz_x));
          w[p][0] = lo[0]; w[p][1] = lo[1]; w[p][2] = lo[2]; w[p][3] = lo[3];
          w[p][4] = hi[0]; w[p][5] = hi[1];
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          bf16x8 xb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            unsigned int d4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              d4[i] = dx == 0 ? w[p][i] : (dx == 2 ? w[p][i + 1] : __builtin_amdgcn_alignbit(w[p][i + 1], w[p][i], 16));
            xb[p] = __builtin_bit_cast(bf16x8, u32x4{d4[0], d4[1], d4[2], d4[3]});
          }
          f32x4 c = acc[dy * 3 + dx];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[2], xb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0], xb[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[1], xb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[1], xb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0], xb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0], xb[0], c, 0, 0, 0);
          acc[dy * 3 + dx] = c;
        }
      }
    }
    __syncthreads();
    t = next;
  }
  if (WPS > 1) {                      // sum the pixel-split partner waves (once per launch)
    float* red = reinterpret_cast<float*>(smem_u);                // [4 waves][9][256]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid * 9 + tap) * 256 + r * 64 + lane] = acc[tap][r];
    __syncthreads();
    if (part == 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[tap][r];
#pragma unroll
          for (int w2 = 1; w2 < WPS; ++w2) v += red[((wid + w2) * 9 + tap) * 256 + r * 64 + lane];
          acc[tap][r] = v;
        }
    }
  }
  if (part == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float* out = a.partial + (((long)blockIdx.x * a.taps + dd * 9 + tap) * a.CoutPad) * a.CinPad;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(long)(co0 + wi * 16 + 4 * g + r) * a.CinPad + ci0 + wj * 16 + li] = acc[tap][r];
    }
  }
}

// Weight gradient of the one-channel 3x3x3 layer: dW[co][tap] = sum_vox dZ[vox][co] * X[vox + tap].
// D[tap][co] = im2col^T [tap][vox] * dZ [vox][co]: M = 27 taps (two 16-row MFMA tiles), N = 16, K = voxels; the im2col
// operand comes from a three-plane halo tile in LDS, dZ is read once straight from HBM (64 B per voxel, coalesced).
// Persistent over 16 x 16 tiles; the four waves split a tile's voxels; one slab [27][CoutPad][CinPad] per workgroup
// in the layout of the generic kernels (only ci = 0 is used) -> wgrad_reduce_kernel finishes.
// DEPTH = 1: one-channel image, 9 taps (one MFMA tile of taps).
template <int DEPTH>
__global__ __launch_bounds__(256) void wgrad_image3d_kernel(WgradArgs a) {
  constexpr int TH = 16, TW = 16, HW_ = TW + 2, NP = (TH + 2) * HW_, NT = 9 * DEPTH;
  __shared__ float Xs[DEPTH * NP];
  __shared__ __attribute__((aligned(16))) float Zs[256 * 16];   // dZ tile [voxel][co], zero outside the plane
  __shared__ float red[4][2][256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
  // Both MFMA operands come straight from LDS reads (as in every other kernel here).  Taps >= NT (rows of D nobody
  // reads) simply alias tap 0.  NB: a first version masked them with a VALU multiply and zero-initialised dZ in a
  // register before a predicated load; hipcc (ROCm 7.2, gfx950) then scheduled VALU writes of the operand registers
  // right around the MFMA and the single-accumulator (DEPTH = 1) chain produced wrong rows on the hardware.
  const int t0 = li < NT ? li : 0, t1 = li + 16 < NT ? li + 16 : 0;
  const int off0 = (t0 / 9) * NP + ((t0 % 9) / 3) * HW_ + t0 % 3;
  const int off1 = (t1 / 9) * NP + ((t1 % 9) / 3) * HW_ + t1 % 3;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const long n_tiles = (long)a.NB * tiles_y * tiles_x;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tx = t % tiles_x; const long r = t / tiles_x; const int ty = r % tiles_y; const int pl = r / tiles_y;
    const int xd = pl % a.D3;
    __syncthreads();
    for (int u = tid; u < DEPTH * NP; u += 256) {
      const int k = u / NP, hp = u - k * NP, hy = hp / HW_, hx = hp - hy * HW_;
      const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1, dk = k - DEPTH / 2, pz = xd + dk;
      Xs[u] = (pz >= 0 && pz < a.D3 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                  ? a.Ain[(((long)(pl + dk) * a.H + gy) * a.W + gx) * a.lda] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + 256 * i, p = u >> 2, q = u & 3;
      const int gy = ty * TH + (p >> 4), gx = tx * TW + (p & 15);
      f32x4 v = {0, 0, 0, 0};
      if (gy < a.H && gx < a.W && 4 * q < a.Cout) {
        const long zo = (((long)pl * a.H + gy) * a.W + gx) * a.ldz + 4 * q;
        if (a.mma == 4) v = __builtin_convertvector(*reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(a.dZ) + zo), f32x4);
        else v = *reinterpret_cast<const f32x4*>(a.dZ + zo);
      }
      *reinterpret_cast<f32x4*>(&Zs[p * 16 + 4 * q]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {                               // wave w: voxels 64w .. 64w+63 of the tile, 4 per step
      const int p = 64 * w + 4 * s + g;
      const float z = Zs[p * 16 + li];
      const float* xb = Xs + (p >> 4) * HW_ + (p & 15);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[off0], z, acc0, 0, 0, 0);
      if (DEPTH == 3) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[off1], z, acc1, 0, 0, 0);
    }
  }
  // lane holds D[tap = 16*h + 4g + r][co = li]; sum the four waves, then one slab per workgroup
#pragma unroll
  for (int r = 0; r < 4; ++r) { red[w][0][(4 * g + r) * 16 + li] = acc0[r]; red[w][1][(4 * g + r) * 16 + li] = acc1[r]; }
  __syncthreads();
  for (int u = tid; u < 512; u += 256) {
    const int h = u >> 8, e = u & 255, tap = 16 * h + (e >> 4), co = e & 15;
    if (tap < NT) {
      const float v = (red[0][h][e] + red[1][h][e]) + (red[2][h][e] + red[3][h][e]);
      a.partial[(((long)blockIdx.x * a.taps + tap) * a.CoutPad + co) * a.CinPad] = v;
    }
  }
}

// EL consecutive slab elements x GR slab groups per 512-thread block.  <16, 32> for launches with many slabs (a
// 9 x 32 x 32 block has only 144 workgroups at 64 elements each, streaming 19 MB: latency-bound at 11 us); <64, 8>
// when there are few slabs (most of 32 groups would idle).  Slab layout [tap][co][ci], dW torch layout.
template <int EL, int GR>
__global__ __launch_bounds__(512) void wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, int taps, int CoutPad, int CinPad,
                                    int Cout, int Cin, float* __restrict__ dW, int accumulate) {
  __shared__ float red[GR][EL];
  const int tx = threadIdx.x % EL, ty = threadIdx.x / EL;
  const long j = (long)blockIdx.x * EL + tx;
  const long tot = (long)taps * Cout * Cin;
  float s0 = 0.f, s1 = 0.f;
  int ci = 0, co = 0, tap = 0;
  if (j < tot) {
    ci = j % Cin; const long r = j / Cin; co = r % Cout; tap = r / Cout;
    const long off = ((long)tap * CoutPad + co) * CinPad + ci, stride = (long)taps * CoutPad * CinPad;
    int c = ty;
    for (; c + GR < chunks; c += 2 * GR) { s0 += partial[off + (long)c * stride]; s1 += partial[off + (long)(c + GR) * stride]; }
    for (; c < chunks; c += GR) s0 += partial[off + (long)c * stride];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && j < tot) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < GR; g += 4) s += (red[g][tx] + red[g + 1][tx]) + (red[g + 2][tx] + red[g + 3][tx]);
    const long o = ((long)co * Cin + ci) * taps + tap;
    dW[o] = accumulate ? dW[o] + s : s;
  }
}
// <16,32> with four consecutive elements (one 16-byte load per slab) per thread: 256-byte runs per slab instead of 64-byte
// ones, a quarter of the blocks.  Same per-element summation order as wgrad_reduce_kernel<16,32> (bit-identical).
__global__ __launch_bounds__(512) void wgrad_reduce4_kernel(const float* __restrict__ partial, int chunks, int taps, int CoutPad, int CinPad,
                                                           int Cout, int Cin, float* __restrict__ dW, int accumulate) {
  constexpr int EL4 = 16, GR = 32;
  __shared__ f32x4 red[GR][EL4];
  const int tx = threadIdx.x % EL4, ty = threadIdx.x / EL4;
  const long j = ((long)blockIdx.x * EL4 + tx) * 4;
  const long tot = (long)taps * Cout * Cin;
  f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
  int ci = 0, co = 0, tap = 0;
  if (j < tot) {
    ci = j % Cin; const long r = j / Cin; co = r % Cout; tap = r / Cout;
    const long off = ((long)tap * CoutPad + co) * CinPad + ci, stride = (long)taps * CoutPad * CinPad;
    int c = ty;
    for (; c + GR < chunks; c += 2 * GR) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(partial + off + (long)c * stride);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(partial + off + (long)(c + GR) * stride);
      s0 += v0; s1 += v1;
    }
    for (; c < chunks; c += GR) s0 += *reinterpret_cast<const f32x4*>(partial + off + (long)c * stride);
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0 && j < tot) {
    f32x4 sv = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < GR; g += 4) sv += (red[g][tx] + red[g + 1][tx]) + (red[g + 2][tx] + red[g + 3][tx]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long o = ((long)co * Cin + ci + e) * taps + tap;
      dW[o] = accumulate ? dW[o] + sv[e] : sv[e];
    }
  }
}
void launch_wgrad_reduce(hipStream_t st, const float* ws, int chunks, int taps, int CoutPad, int CinPad, int Cout, int Cin,
                                float* dW, int accumulate) {
  const long tot = (long)Cout * Cin * taps;
  static const bool v4 = !(getenv("ARCO_WGRAD_REDUCE4") && atoi(getenv("ARCO_WGRAD_REDUCE4")) == 0);
  if (v4 && chunks > 16 && (Cin & 3) == 0 && (CinPad & 3) == 0 && tot >= 64 * 64)
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((tot / 4 + 15) / 16)), dim3(512), 0, st, ws, chunks, taps, CoutPad, CinPad, Cout, Cin, dW, accumulate);
  else if (chunks > 16)
    hipLaunchKernelGGL((wgrad_reduce_kernel<16, 32>), dim3((tot + 15) / 16), dim3(512), 0, st, ws, chunks, taps, CoutPad, CinPad, Cout, Cin, dW, accumulate);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<64, 8>), dim3((tot + 63) / 64), dim3(512), 0, st, ws, chunks, taps, CoutPad, CinPad, Cout, Cin, dW, accumulate);
}

// column sums (bias gradient): out[c] = sum_pix X[pix][c].  float4 lanes along channels,
// rows strided over the block, per-block slabs + fp64 fixed-order finalize.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, long ldx, long M, int C,
                                                            float* __restrict__ partial) {
  const long rpb = (M + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
  extern __shared__ __attribute__((aligned(16))) float red[];   // [256][4]
  if ((C & 3) == 0 && (ldx & 3) == 0 && C <= 1024) {
    const int q4 = C / 4, tq = threadIdx.x % q4, tr = threadIdx.x / q4, rstep = 256 / q4;
    f32x4 s = {0, 0, 0, 0};
    if (tr < rstep)
      for (long r = r0 + tr; r < r1; r += rstep) s += ld4f(X + r * ldx + 4 * tq);
    *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int qq = c / 4, e = c % 4;
      float a = 0.f;
      for (int r = 0; r < rstep; ++r) a += red[(r * q4 + qq) * 4 + e];
      partial[(long)blockIdx.x * C + c] = a;
    }
  } else if (C <= 256) {
    // narrow / odd channel counts (the 2-class out_conv of the V-Net: 4 M rows x 2 channels): thread = (row lane, channel)
    const int rstep = 256 / C, c = threadIdx.x % C, tr = threadIdx.x / C;
    float s = 0.f;
    if (tr < rstep)
      for (long r = r0 + tr; r < r1; r += rstep) s += (float)X[r * ldx + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
      float a = 0.f;
      for (int r = 0; r < rstep; ++r) a += red[r * C + threadIdx.x];
      partial[(long)blockIdx.x * C + threadIdx.x] = a;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      for (long r = r0; r < r1; ++r) s += (float)X[r * ldx + c];
      partial[(long)blockIdx.x * C + c] = s;
    }
  }
}
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[(long)b * C + c];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

__global__ void transpose2d_kernel(const float* __restrict__ x, long ldx, int rows, int cols, float* __restrict__ y, long ldy) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int rr = by + r, cc = bx + threadIdx.x;
    tile[r][threadIdx.x] = (rr < rows && cc < cols) ? x[(long)rr * ldx + cc] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int cc = bx + r, rr = by + threadIdx.x;   // output row = input col
    if (cc < cols && rr < rows) y[(long)cc * ldy + rr] = tile[threadIdx.x][r];
  }
}

extern "C" {

// Query: number of M-blocks (= BN-stat partial slabs per channel) the conv launch will use.
int arco_conv_mblocks(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups) {
  IgemmArgs a{};
  a.stat_groups = stat_groups > 1 ? stat_groups : 1;
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in;
  int q[3] = {0, 0, 0};
  if (dispatch_igemm(a, taps, nullptr, q) != ARCO_OK) return ARCO_ERR_UNSUPPORTED;
  return q[0];
}
// the same for a launch in matrix-core mode `mma` (3: the split-bf16 kernels tile differently from the fp32 halo kernel)
int arco_conv_mblocks_mma(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups, int mma) {
  IgemmArgs a{};
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in;
  a.stat_groups = stat_groups > 1 ? stat_groups : 1;
  a.mma = mma;
  if (mma == 3) a.Kg = (Cin + 31) / 32 * 2;
  int q[3] = {0, 0, 0};
  if (mma == 4 && !(taps == 27 && image_conv3d_eligible(a))) {
    a.Kpad = (Cin + 31) / 32 * 32;
    return hconv_dispatch(a, taps, nullptr, q) == ARCO_OK ? q[0] : ARCO_ERR_UNSUPPORTED;
  }
  if (dispatch_igemm(a, taps, nullptr, q) != ARCO_OK) return ARCO_ERR_UNSUPPORTED;
  return q[0];
}

// ... for a launch through arco_conv3d_fwd_pro (consumer-side activation with pro_groups BatchNorm groups): the 3x3x3 kernels choose their
// tile shape among the forms that have the activation in their loaders, so the slab count can differ from the plain launch's
int arco_conv_mblocks_pro(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups, int mma, int pro_groups) {
  if (taps != 27) return arco_conv_mblocks_mma(taps, NB, H, W, Cin, Cout, ld_in, stat_groups, mma);
  if (mma != 3) return ARCO_ERR_UNSUPPORTED;
  IgemmArgs a{};
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in; a.ldc = Cout; a.mma = 3;
  a.Kg = (Cin + 31) / 32 * 2; a.stat_groups = stat_groups > 1 ? stat_groups : 1;
  static const float one = 1.f;
  a.pro.mean = a.pro.istd = a.pro.gamma = a.pro.beta = &one; a.pro.groups = pro_groups > 1 ? pro_groups : 1;
  int q[3] = {0, 0, 0};
  return conv3d_fl_dispatch(a, nullptr, q) == ARCO_OK ? q[0] : ARCO_ERR_UNSUPPORTED;
}

// which igemm_kernel<TAPS,BM,BN,..> instantiation a launch uses: returns TAPS*1e6 + BM*1e3 + BN (kernel-tap form:
// 9 for both 3x3 and 3x3x3); *kc_depth_db = KC*100 + DEPTH*10 + DB
int arco_conv_config(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int* kc_depth_db) {
  IgemmArgs a{};
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in;
  int q[3] = {0, 0, 0};
  if (dispatch_igemm(a, taps, nullptr, q) != ARCO_OK) return ARCO_ERR_UNSUPPORTED;
  if (kc_depth_db) *kc_depth_db = q[2];
  return q[1];
}
int arco_conv_config_mma(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int mma) {   // ... in matrix-core mode mma
  IgemmArgs a{};
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in; a.mma = mma;
  if (mma == 3) a.Kg = (Cin + 31) / 32 * 2;
  int q[3] = {0, 0, 0};
  if (mma == 4 && !(taps == 27 && image_conv3d_eligible(a))) {
    a.Kpad = (Cin + 31) / 32 * 32;
    return hconv_dispatch(a, taps, nullptr, q) == ARCO_OK ? q[1] : ARCO_ERR_UNSUPPORTED;
  }
  if (dispatch_igemm(a, taps, nullptr, q) != ARCO_OK) return ARCO_ERR_UNSUPPORTED;
  return q[1];
}

// 1 when a convolution of this shape runs on the split-bf16 (mma = 3) kernels - the caller then passes mma = 3 and the
// split-packed weights (mode | 2); 0: it runs on a kernel that only takes the fp32 format (one- / few-channel image
// kernels, the halo kernel of the shallow 2-D levels, scalar-load shapes)
int arco_conv_split_ok(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in) {
  IgemmArgs a{};
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in;
  if ((Cin & 3) != 0 || (ld_in & 3) != 0) return 0;
  if (taps == 1) return 1;
  if (taps == 27) return image_conv3d_eligible(a) ? 0 : 1;
  // of the shallow 2-D levels only 16 -> 16 stays on the fp32 halo kernel (HBM-bound either way, measured faster there);
  // 16 -> 32, 32 -> 16 and 32 -> 32 run 10-25 % faster on the split-bf16 implicit GEMM
  if (taps == 9) {
    if (image_conv3d_eligible(a)) return 0;
    if (image_conv_eligible(a) || (halo_eligible(a) && a.K == 16 && a.Npad <= 16)) {
      // ... unless the resident-weights pipelined kernel takes the launch (16 x 256^2 planes: HBM-bound there, the halo
      // kernel is not); A/B switch ARCO_CONV_RW16=0
      static const bool rw16 = !(getenv("ARCO_CONV_RW16") && atoi(getenv("ARCO_CONV_RW16")) == 0);
      int q[3];
      a.mma = 3;
      return rw16 && conv_sp_dispatch(a, nullptr, q) == ARCO_OK && q[1] == 9358016 ? 1 : 0;     // (conv3x3_rw_kernel<8,1> only)
    }
    return 1;
  }
  return 0;
}

int arco_pack_conv_weight(const float* W, int Cout, int Cin, int taps, int mode, float* Wp, void* stream) {
  ARCO_CHECK_ARG(Cout > 0 && Cin > 0 && (taps == 1 || taps == 9 || taps == 27) && mode >= 0 && mode <= 5 && (mode & 6) != 6);
  const int N = (mode & 1) == 0 ? Cout : Cin, K = (mode & 1) == 0 ? Cin : Cout;
  const int Npad = (N + 15) / 16 * 16, Kpad = (mode & 6) ? (K + 31) / 32 * 32 : (K + 15) / 16 * 16;   // split: Wp holds 3 bf16 per element; mode | 4: one f16
  const long tot = (long)taps * Npad * Kpad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((tot + 255) / 256), dim3(256), 0, as_stream(stream), W, Cout, Cin, taps,
                     mode, Npad, Kpad, Wp);
  return arco_launch_status();
}

int arco_conv3d_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                    const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                    int NV, int D3, int H, int W, int stat_groups, int mma, void* stream);
// desc: device array of n_desc PackDesc records (arco_pack_desc_bytes() each, see igemm.hip); total = sum of packed sizes
long arco_pack_desc_bytes() { return (long)sizeof(PackDesc); }
int arco_pack_many(const void* desc, int n_desc, long total, void* stream) {
  if (n_desc <= 0 || total <= 0) return ARCO_OK;
  // (the descriptors live in device memory: the x extent is a bound - 256 tiles cover a 256 x 256 weight, element-wise blocks stride)
  hipLaunchKernelGGL(pack_many_kernel, dim3(256, (unsigned)n_desc), dim3(256), 0, as_stream(stream), (const PackDesc*)desc, n_desc, total);
  return arco_launch_status();
}

// out[M][N] = in[M][K] . W[N][K]^T for a SMALL M x N and a LONG K (the InfoNCE anchor gradient: 256 x 496 x ~4600):
// K is cut into `splits` slabs (grid.y) so the launch has splits x as many workgroups; slabs land in ws
// (splits * M * ld_out floats) and are summed in fixed order.
int arco_gemm_splitk(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, long M,
                     int splits, float* ws, void* stream) {
  ARCO_CHECK_ARG(in && Wp && out && ws && K > 0 && N > 0 && M > 0 && splits >= 1 && (ld_out & 3) == 0);
  IgemmArgs a{};
  a.A = in; a.lda = ld_in; a.Wp = Wp; a.N = N; a.K = K;
  a.Npad = (N + 15) / 16 * 16; a.Kpad = (K + 15) / 16 * 16;
  a.C = ws; a.ldc = ld_out; a.NB = 1; a.H = 1; a.W = (int)M; a.M = M; a.D3 = 1;
  a.ksplit = splits; a.slab_stride = M * ld_out;
  const int rc = dispatch_igemm(a, 1, as_stream(stream), nullptr);
  if (rc != ARCO_OK) return rc;
  const long n4 = M * ld_out / 4;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, as_stream(stream), ws, splits, n4, out);
  return arco_launch_status();
}

// `batch` independent GEMMs out_z[M][N] = in_z[M][K] . W_z[N][K]^T in ONE launch (blockIdx.z = z): operand z lives at
// base + z * stride (floats).  splits > 1: split-K as arco_gemm_splitk - slab (z, y) lands in ws at
// (z * splits + y) * M * ld_out and the fixed-order slab sum writes out_z; ws may be NULL when splits == 1.
// The grouped InfoNCE (one problem per class: scores A_c . Bank_c^T and anchor gradients W_c . Bank_c) runs on these.
__global__ __launch_bounds__(256) void slab_sum_batched_kernel(const float* __restrict__ ws, int splits, long n4, long out_stride4,
                                                              float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long z = blockIdx.y;
  const f32x4* w = reinterpret_cast<const f32x4*>(ws) + z * splits * n4;
  f32x4 s = w[i];
  for (int k = 1; k < splits; ++k) s += w[(long)k * n4 + i];
  reinterpret_cast<f32x4*>(out)[z * out_stride4 + i] = s;
}
int arco_gemm_batched(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, long M,
                      int batch, long stride_in, long stride_w, long stride_out, int splits, float* ws, void* stream) {
  ARCO_CHECK_ARG(in && Wp && out && K > 0 && N > 0 && M > 0 && batch >= 1 && splits >= 1 && (ld_out & 3) == 0);
  ARCO_CHECK_ARG(splits == 1 || (ws && (stride_out & 3) == 0));
  IgemmArgs a{};
  a.A = in; a.lda = ld_in; a.Wp = Wp; a.N = N; a.K = K;
  a.Npad = (N + 15) / 16 * 16; a.Kpad = (K + 15) / 16 * 16;
  a.ldc = ld_out; a.NB = 1; a.H = 1; a.W = (int)M; a.M = M; a.D3 = 1;
  a.batch = batch; a.batchA = stride_in; a.batchW = stride_w;
  if (splits > 1) { a.C = ws; a.ksplit = splits; a.slab_stride = M * ld_out; a.batchC = (long)splits * M * ld_out; }
  else { a.C = out; a.batchC = stride_out; }
  const int rc = dispatch_igemm(a, 1, as_stream(stream), nullptr);
  if (rc != ARCO_OK || splits == 1) return rc;
  const long n4 = M * ld_out / 4;
  hipLaunchKernelGGL(slab_sum_batched_kernel, dim3((unsigned)((n4 + 255) / 256), (unsigned)batch), dim3(256), 0, as_stream(stream),
                     ws, splits, n4, stride_out / 4, out);
  return arco_launch_status();
}

// out[pix][0..N) = conv(in)[pix] (+bias) (+residual); channels-last; taps in {1, 9}
int arco_conv_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                  const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                  int NB, int H, int W, void* stream) {
  return arco_conv3d_fwd(in, ld_in, K, Wp, N, out, ld_out, bias, residual, ld_res, stat_sum, stat_sq, taps, NB, 1, H, W,
                         1, 0, stream);
}

// out[voxel][0..N) = W . in[voxel] + trilinear_align_corners(lo)[voxel]: FeatureExtractor_3d's fea_i over cat(up(x), f_i) with the
// wide block of the weights pushed under the upsample (model_3D.py:46-58) - the 1x1x1 GEMM over the high-resolution map f_i with the
// upsampled low-resolution product sampled in its epilogue instead of being written and read back (gemm_sp.hip).  Split-bf16 mode
// only; ARCO_ERR_UNSUPPORTED when the pipelined kernel does not take the shape (the caller then upsamples separately).
int arco_conv1x1_upres_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, const float* lo,
                           long ld_lo, int NV, int uD, int uH, int uW, int oD, int oH, int oW, void* stream) {
  ARCO_CHECK_ARG(in && Wp && out && lo && K > 0 && N > 0 && NV > 0 && uD > 0 && uH > 0 && uW > 0 && oD > 0 && oH > 0 && oW > 0);
  IgemmArgs a{};
  a.A = in; a.lda = ld_in; a.Wp = Wp; a.N = N; a.K = K;
  a.Npad = (N + 15) / 16 * 16;
  a.C = out; a.ldc = ld_out;
  a.NB = NV * oD; a.H = oH; a.W = oW; a.M = (long)NV * oD * oH * oW; a.D3 = oD;
  a.stat_groups = 1; a.mma = 3;
  a.Kg = (K + 31) / 32 * 2; a.Kpad = a.Kg * 16;
  a.Rup = lo; a.ldrup = ld_lo; a.uD = uD; a.uH = uH; a.uW = uW; a.oD = oD; a.oH = oH; a.oW = oW;
  const int r = gemm_sp_dispatch(a, as_stream(stream), nullptr);
  return r == -1 ? ARCO_ERR_UNSUPPORTED : r;
}

// 1 when arco_conv3d_fwd_pro / arco_conv3d_wgrad_pro take a convolution of this shape with a consumer-side activation on its input
// (the pipelined split-bf16 3x3 kernels of conv_sp.hip and wgrad_split_kernel); else the caller materialises the activation
int arco_conv_pro_ok(int taps, int NV, int D3, int H, int W, int Cin, int Cout, long ld_in, int mma, int groups) {
  if (taps == 27) {       // 3x3x3: conv3d_fc_kernel's loaders (conv3d_fl.hip; BatchNorm + ReLU / LeakyReLU, no dropout), gradient-free passes
    if (mma != 3 || groups < 1 || NV % groups != 0 || D3 < 1 || (Cin & 15) != 0 || (ld_in & 3) != 0) return 0;
    IgemmArgs a{};
    a.NB = NV * D3; a.H = H; a.W = W; a.M = (long)a.NB * H * W; a.D3 = D3;
    a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in; a.ldc = Cout; a.mma = 3;
    a.Kg = (Cin + 31) / 32 * 2; a.stat_groups = 1;
    static const float one = 1.f;
    a.pro.mean = a.pro.istd = a.pro.gamma = a.pro.beta = &one; a.pro.groups = groups;      // (query only: never dereferenced)
    int q[3] = {0, 0, 0};
    return conv3d_fl_dispatch(a, nullptr, q) == ARCO_OK ? 1 : 0;
  }
  if (taps != 9 || D3 != 1 || mma != 3 || groups < 1 || NV % groups != 0 || (Cin & 15) != 0 || (Cout & 3) != 0 || (ld_in & 3) != 0) return 0;
  if ((long)NV * H * W * Cin >= (1l << 31)) return 0;          // 32-bit element indices of the dropout mask
  IgemmArgs a{};
  a.NB = NV; a.H = H; a.W = W; a.M = (long)NV * H * W; a.D3 = 1;
  a.N = Cout; a.Npad = (Cout + 15) / 16 * 16; a.K = Cin; a.Kpad = (Cin + 15) / 16 * 16; a.lda = ld_in; a.mma = 3;
  a.Kg = (Cin + 31) / 32 * 2; a.stat_groups = 1;
  int q[3] = {0, 0, 0};
  return conv_sp_dispatch(a, nullptr, q) == ARCO_OK ? 1 : 0;
}

static int conv3d_fwd_impl(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                           const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                           int NV, int D3, int H, int W, int stat_groups, int mma, const ArcoActPro* pro, void* stream);
int arco_conv3d_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                    const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                    int NV, int D3, int H, int W, int stat_groups, int mma, void* stream) {
  return conv3d_fwd_impl(in, ld_in, K, Wp, N, out, ld_out, bias, residual, ld_res, stat_sum, stat_sq, taps, NV, D3, H, W, stat_groups, mma,
                         nullptr, stream);
}
// ... with a consumer-side activation: `in` is the pre-activation z of the producing convolution, pro its BatchNorm statistics /
// affine parameters / LeakyReLU slope / dropout (unetWithArgs.py:36-44: the Conv-BN-LeakyReLU-Dropout in front of a block's second conv)
int arco_conv3d_fwd_pro(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                        const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                        int NV, int D3, int H, int W, int stat_groups, int mma, const ArcoActPro* pro, void* stream) {
  ARCO_CHECK_ARG(pro && pro->mean && pro->istd && pro->gamma && pro->beta && pro->groups >= 1 && (pro->drop_mode == 0 || pro->drop_mode == 1) &&
                 pro->p >= 0.f && pro->p < 1.f);
  if (!arco_conv_pro_ok(taps, NV, D3, H, W, K, N, ld_in, mma, pro->groups)) return ARCO_ERR_UNSUPPORTED;
  return conv3d_fwd_impl(in, ld_in, K, Wp, N, out, ld_out, bias, residual, ld_res, stat_sum, stat_sq, taps, NV, D3, H, W, stat_groups, mma,
                         pro, stream);
}

// 3-D generalisation: NV volumes of D3 planes of H x W; taps in {1, 9 (per plane), 27 (3x3x3, pad 1)}
static int conv3d_fwd_impl(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                           const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                           int NV, int D3, int H, int W, int stat_groups, int mma, const ArcoActPro* pro, void* stream) {
  const int NB = NV * D3;
  ARCO_CHECK_ARG(in && Wp && out && K > 0 && N > 0 && NB > 0 && H > 0 && W > 0 && D3 > 0 && mma >= 0 && mma <= 4);
  IgemmArgs a{};
  a.A = in; a.lda = ld_in; a.Wp = Wp; a.N = N; a.K = K;
  a.Npad = (N + 15) / 16 * 16; a.Kpad = (K + 15) / 16 * 16;
  a.C = out; a.ldc = ld_out; a.bias = bias; a.R = residual; a.ldr = ld_res;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.D3 = D3;
  a.stat_groups = stat_groups > 1 ? stat_groups : 1;
  a.mma = mma;
  if (mma == 3) {       // split-bf16: Wp is the pre-split packed format (arco_pack_conv_weight_split / arco_pack_many fmt 1)
    a.Kg = (K + 31) / 32 * 2;
    a.Kpad = taps == 1 ? a.Kg * 16 : (K + 15) / 16 * 16;
  }
  ARCO_CHECK_ARG(NV % a.stat_groups == 0);
  if (pro && taps == 27) {      // (arco_conv_pro_ok has vouched for the shape: conv3d_fc_kernel<.., PRO>)
    a.pro = *pro;
    const int r = conv3d_fl_dispatch(a, as_stream(stream), nullptr);
    return r == -1 ? ARCO_ERR_UNSUPPORTED : r;
  }
  if (pro) {            // (arco_conv_pro_ok has vouched for the shape: the launch below lands in conv_sp_dispatch)
    a.pro = *pro;
    const int r = conv_sp_dispatch(a, as_stream(stream), nullptr);
    return r == -1 ? ARCO_ERR_UNSUPPORTED : r;
  }
  if (mma == 4) {       // f16 activation storage: `out` (and `in`, unless this is the one-channel fp32 volume of the first layer) are f16
    if (taps == 27 && image_conv3d_eligible(a)) return launch_image_conv3d<3>(a, as_stream(stream), nullptr);
    a.Kpad = (K + 31) / 32 * 32;
    return hconv_dispatch(a, taps, as_stream(stream), nullptr);
  }
  return dispatch_igemm(a, taps, as_stream(stream), nullptr);
}

int arco_conv3d_wgrad(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                      int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, void* stream);
long arco_wgrad_ws_floats(int Cout, int Cin, int taps, long M) {
  const int co_b = Cout >= 64 ? 64 : (Cout > 16 ? 32 : 16), ci_b = Cin >= 64 ? 64 : (Cin > 16 ? 32 : 16);
  const int CoutPad = (Cout + co_b - 1) / co_b * co_b, CinPad = (Cin + ci_b - 1) / ci_b * ci_b;
  const long n_tiles = (M + 127) / 128 * 4 + 64;   // upper bound incl. ragged spatial tiles
  const long yz = (long)(CoutPad / co_b) * (CinPad / ci_b) * taps;
  long chunks = 2048 / yz; if (chunks < 1) chunks = 1; if (chunks > n_tiles) chunks = n_tiles;
  if (taps >= 9) {
    const int hco = Cout > 16 ? 32 : 16, hci = Cin > 16 ? 32 : 16;
    const long cop = (Cout + hco - 1) / hco * hco, cip = (Cin + hci - 1) / hci * hci;
    chunks = 1536 / ((taps / 9) * (cop / hco) * (cip / hci)); if (chunks > n_tiles) chunks = n_tiles; if (chunks < 1) chunks = 1;
    return chunks * taps * cop * cip;
  }
  return chunks * taps * CoutPad * CinPad;
}

int arco_conv_wgrad(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NB,
                    int H, int W, float* ws, float* dW, int accumulate, void* stream) {
  return arco_conv3d_wgrad(dZ, ld_dz, Cout, in, ld_in, Cin, taps, NB, 1, H, W, ws, dW, accumulate, 0, stream);
}

static int conv3d_wgrad_impl(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                             int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, const ArcoActPro* pro, void* stream);
int arco_conv3d_wgrad(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                      int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, void* stream) {
  return conv3d_wgrad_impl(dZ, ld_dz, Cout, in, ld_in, Cin, taps, NV, D3, H, W, ws, dW, accumulate, mma, nullptr, stream);
}
// ... where `in` is the pre-activation z of the producing convolution (see arco_conv3d_fwd_pro): dW = dZ^T . act(z)
int arco_conv3d_wgrad_pro(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                          int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, const ArcoActPro* pro, void* stream) {
  ARCO_CHECK_ARG(pro && pro->mean && pro->istd && pro->gamma && pro->beta && pro->groups >= 1 && (pro->drop_mode == 0 || pro->drop_mode == 1) &&
                 pro->p >= 0.f && pro->p < 1.f);
  if (taps != 9) return ARCO_ERR_UNSUPPORTED;        // (the 3x3x3 consumer-side activation exists in the forward kernel only: gradient-free passes)
  if (!arco_conv_pro_ok(taps, NV, D3, H, W, Cin, Cout, ld_in, mma, pro->groups) || (ld_dz & 3) != 0) return ARCO_ERR_UNSUPPORTED;
  return conv3d_wgrad_impl(dZ, ld_dz, Cout, in, ld_in, Cin, taps, NV, D3, H, W, ws, dW, accumulate, mma, pro, stream);
}

static int conv3d_wgrad_impl(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                             int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, const ArcoActPro* pro, void* stream) {
  const int NB = NV * D3;
  ARCO_CHECK_ARG(dZ && in && ws && dW && (taps == 1 || taps == 9 || taps == 27) && mma >= 0 && mma <= 4);
  if (mma == 4 && !(taps >= 9 && Cin == 1))      // f16 activation storage (dZ and in are f16; the first layer's input volume is fp32)
    return hwgrad_dispatch(dZ, ld_dz, Cout, in, ld_in, Cin, taps, NB, D3, H, W, ws, dW, accumulate, as_stream(stream));
  WgradArgs a{};
  a.D3 = D3; a.mma = mma >= 3 ? mma : ((taps == 27 && mma) ? 2 : 0);   // 1 / 2: bf16 operands (gradients: range); 3: split-bf16 (fp32-accurate)
  a.dZ = dZ; a.ldz = ld_dz; a.Cout = Cout; a.Ain = in; a.lda = ld_in; a.Cin = Cin; a.taps = taps;
  a.NB = NB; a.H = H; a.W = W; a.M = (long)NB * H * W; a.partial = ws;
  static const int abl = getenv("ARCO_WGRAD_ABL") ? atoi(getenv("ARCO_WGRAD_ABL")) : 0;
  a.abl = abl;
  if (pro) a.pro = *pro;
  const int co_b = Cout >= 64 ? 64 : (Cout > 16 ? 32 : 16), ci_b = Cin >= 64 ? 64 : (Cin > 16 ? 32 : 16);
  a.CoutPad = (Cout + co_b - 1) / co_b * co_b; a.CinPad = (Cin + ci_b - 1) / ci_b * ci_b;
  a.n_tiles = taps >= 9 ? NB * ((H + 7) / 8) * ((W + 15) / 16) : (int)((a.M + 127) / 128);
  hipStream_t st = as_stream(stream);
  if (taps >= 9 && Cin == 1 && Cout <= 16 && (Cout & 3) == 0 && (ld_dz & 3) == 0) {   // one-channel input (first layer of both nets): taps as M
    a.CoutPad = 16; a.CinPad = 16;
    if (taps == 9) a.D3 = 1;
    const long tiles = (long)NB * ((H + 15) / 16) * ((W + 15) / 16);
    const long chunks = tiles < 512 ? tiles : 512;              // <= the slab count arco_wgrad_ws_floats reserves
    if (taps == 27) hipLaunchKernelGGL(wgrad_image3d_kernel<3>, dim3((unsigned)chunks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_image3d_kernel<1>, dim3((unsigned)chunks), dim3(256), 0, st, a);
    launch_wgrad_reduce(st, ws, (int)chunks, taps, a.CoutPad, a.CinPad, Cout, Cin, dW, accumulate);
    return arco_launch_status();
  }
  if (taps >= 9) {       // spatial kernels: all taps of a plane per block, operands staged once (halo in LDS)
    const int hco = Cout > 16 ? 32 : 16, hci = Cin > 16 ? 32 : 16;
    a.CoutPad = (Cout + hco - 1) / hco * hco; a.CinPad = (Cin + hci - 1) / hci * hci;
    // narrow planes (W not a multiple of 16): flat-position tiles for the fp32 / reduced-precision kernels.  In split-bf16 mode
    // the rectangular 8 x 16 tiles of wgrad_split_kernel are taken instead, ragged last column of tiles and all (W = 40: 17 %
    // of the MFMA rows idle, W = 20 / 10: 37 %) - six bf16 MFMAs per 32 pixels still beat eight fp32 MFMAs per 16 by more than
    // that (A/B knob ARCO_WGRAD_FLAT_SPLIT=0: round 2's choice, the fp32 flat-tile kernel)
    static const int flat_split = getenv("ARCO_WGRAD_FLAT_SPLIT") ? atoi(getenv("ARCO_WGRAD_FLAT_SPLIT")) : 1;
    const bool split_ok = a.mma == 3 && (Cout & 3) == 0 && (Cin & 3) == 0 && (ld_dz & 3) == 0 && (ld_in & 3) == 0;
    const bool flat = (W & 15) != 0 && W + 2 <= IGEMM_FLAT_WPMAX && !(split_ok && flat_split && W >= 10);
    if (flat) a.n_tiles = NB * ((H * (W + 2) + 127) / 128);
    const int zdim = taps / 9, ydim = (a.CoutPad / hco) * (a.CinPad / hci);
    // 32x32 blocks run persistent (2 workgroups per CU, several tiles each); the others one slab per ~tile
    // (16 x 32 blocks of the split kernel: 60.7 KB of LDS, two per CU as well - 768 left a third round at half occupancy)
    static const long target_env = getenv("ARCO_WGRAD_TARGET") ? atol(getenv("ARCO_WGRAD_TARGET")) : 0;     // A/B knob (<= the defaults: the slab reservation)
    const long target = target_env > 0 ? target_env : ((hci == 32 && (hco == 32 || a.mma == 3)) ? 512 : 768);
    long chunks = target / ((long)zdim * ydim); if (chunks > a.n_tiles) chunks = a.n_tiles; if (chunks < 1) chunks = 1;
    dim3 hgrid((unsigned)chunks, ydim, zdim);
#define WH(COB, CIB)                                                                              \
    do {                                                                                          \
      constexpr int LZ = (COB % 32 == 0) ? COB + 16 : COB, LA = (CIB % 32 == 0) ? CIB + 16 : CIB; \
      size_t sh = (size_t)(128 * LZ + (flat ? 128 + 2 * (W + 2) + 2 : 180) * LA) * 4; const size_t rd = (size_t)4 * 9 * 256 * 4;  \
      if (sh < rd && (COB / 16) * (CIB / 16) < 4) sh = rd;                                        \
      if (flat && a.mma == 2) {                                                                   \
        static unsigned long long attr_set2 = 0;                                                            \
        if (arco_first_on_device(attr_set2)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_halo2_kernel<COB, CIB, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (128 * LZ + (128 + 2 * IGEMM_FLAT_WPMAX + 2) * LA) * 4); } \
        hipLaunchKernelGGL((wgrad_halo2_kernel<COB, CIB, true, 2>), hgrid, dim3(256), sh, st, a); \
      } else if (flat) {                                                                          \
        static unsigned long long attr_set = 0;                                                             \
        if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_halo2_kernel<COB, CIB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (128 * LZ + (128 + 2 * IGEMM_FLAT_WPMAX + 2) * LA) * 4); } \
        hipLaunchKernelGGL((wgrad_halo2_kernel<COB, CIB, true>), hgrid, dim3(256), sh, st, a);    \
      } else if (a.mma == 2) hipLaunchKernelGGL((wgrad_halo2_kernel<COB, CIB, false, 2>), hgrid, dim3(256), sh, st, a); \
      else hipLaunchKernelGGL((wgrad_halo2_kernel<COB, CIB>), hgrid, dim3(256), sh, st, a);       \
    } while (0)
    const bool split = a.mma == 3 && !flat && (Cout & 3) == 0 && (Cin & 3) == 0 && (ld_dz & 3) == 0 && (ld_in & 3) == 0;
#define WS(COB, CIB)                                                                              \
    do {                                                                                          \
      size_t sh = (size_t)(3 * COB * ARCO_WG_CSZ + 3 * CIB * ARCO_WG_CSX) * 4; const size_t rd = (size_t)4 * 9 * 256 * 4; \
      if (sh < rd && (COB / 16) * (CIB / 16) < 4) sh = rd;                                        \
      static unsigned long long attr_s = 0;                                                                 \
      if (sh > 64 * 1024 && arco_first_on_device(attr_s)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<COB, CIB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); } \
      if (a.pro.mean) {                                                                           \
        static unsigned long long attr_p = 0;                                                               \
        if (sh > 64 * 1024 && arco_first_on_device(attr_p)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel<COB, CIB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); } \
        hipLaunchKernelGGL((wgrad_split_kernel<COB, CIB, true>), hgrid, dim3(256), sh, st, a);    \
      } else                                                                                      \
      hipLaunchKernelGGL((wgrad_split_kernel<COB, CIB>), hgrid, dim3(256), sh, st, a);            \
    } while (0)
    if (pro && !split) return ARCO_ERR_UNSUPPORTED;
    if (split) {
      if (hco == 32 && hci == 32) WS(32, 32);
      else if (hco == 32 && hci == 16) WS(32, 16);
      else if (hco == 16 && hci == 32) WS(16, 32);
      else WS(16, 16);
    }
    else if (hco == 32 && hci == 32) WH(32, 32);
    else if (hco == 32 && hci == 16) WH(32, 16);
    else if (hco == 16 && hci == 32) WH(16, 32);
    else WH(16, 16);
#undef WS
#undef WH
    launch_wgrad_reduce(st, ws, (int)chunks, taps, a.CoutPad, a.CinPad, Cout, Cin, dW, accumulate);
    return arco_launch_status();
  }
  {   // wide 1x1 gradients over many pixels: 128 x 128 blocks, one workgroup per CU (wgrad_q_kernel); A/B knob ARCO_WGRAD_Q=0
    static const int wq = getenv("ARCO_WGRAD_Q") ? atoi(getenv("ARCO_WGRAD_Q")) : 1;
    const int cop = (Cout + 127) / 128 * 128, cip = (Cin + 127) / 128 * 128;
    const long yzq = (long)(cop / 128) * (cip / 128);
    long chq = 256 / yzq; if (chq < 1) chq = 1; if (chq > a.n_tiles) chq = a.n_tiles;
    if (wq && taps == 1 && a.mma != 4 && Cout >= 192 && Cin >= 192 && a.M >= 32768 && (Cout & 3) == 0 && (Cin & 3) == 0 &&
        (ld_dz & 3) == 0 && (ld_in & 3) == 0 && (reinterpret_cast<uintptr_t>(dZ) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
        chq * cop * cip <= arco_wgrad_ws_floats(Cout, Cin, taps, a.M)) {
      a.CoutPad = cop; a.CinPad = cip;
      static const int tp = getenv("ARCO_WGRAD_Q_TP") ? atoi(getenv("ARCO_WGRAD_Q_TP")) : 64;
      if (tp == 64) {
        a.n_tiles = (int)((a.M + 63) / 64);
        long ch2 = 512 / yzq; if (ch2 < 1) ch2 = 1; if (ch2 > a.n_tiles) ch2 = a.n_tiles;
        if (ch2 * cop * cip <= arco_wgrad_ws_floats(Cout, Cin, taps, a.M)) chq = ch2;
        constexpr int shq = 64 * (2 * (128 + WGRAD1_PAD)) * 4;
        static unsigned long long attr_q = 0;
        if (arco_first_on_device(attr_q)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_q_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, shq); }
        hipLaunchKernelGGL(wgrad_q_kernel<64>, dim3((unsigned)chq, cop / 128, cip / 128), dim3(256), shq, st, a);
      } else {
        constexpr int shq = 128 * (2 * (128 + WGRAD1_PAD)) * 4;
        static unsigned long long attr_q = 0;
        if (arco_first_on_device(attr_q)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_q_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, shq); }
        hipLaunchKernelGGL(wgrad_q_kernel<128>, dim3((unsigned)chq, cop / 128, cip / 128), dim3(256), shq, st, a);
      }
      launch_wgrad_reduce(st, ws, (int)chq, taps, a.CoutPad, a.CinPad, Cout, Cin, dW, accumulate);
      return arco_launch_status();
    }
  }
  const long yz = (long)(a.CoutPad / co_b) * (a.CinPad / ci_b) * taps;
  static const long target1 = getenv("ARCO_WGRAD1_TARGET") ? atol(getenv("ARCO_WGRAD1_TARGET")) : 512;   // two resident workgroups per CU, 6-13 tiles each (2048: 3 tiles each, a third of them behind an exposed first fetch; 4x the slabs)
  long chunks = target1 / yz; if (chunks < 1) chunks = 1; if (chunks > a.n_tiles) chunks = a.n_tiles;
  dim3 grid((unsigned)chunks, a.CoutPad / co_b, (a.CinPad / ci_b) * taps);
#define WG(COB, CIB)                                                                              \
  do {                                                                                            \
    constexpr int LZ = (COB % 32 == 0) ? COB + WGRAD1_PAD : COB, LA = (CIB % 32 == 0) ? CIB + WGRAD1_PAD : CIB;   \
    size_t sh = (size_t)128 * (LZ + LA) * 4; const size_t rd = (size_t)4 * COB * CIB * 4;         \
    if (sh < rd) sh = rd;                                                                         \
    auto kern = wgrad_kernel<COB, CIB>;                                                           \
    static unsigned long long attr_set = 0;   /* once per instantiation: not legal inside a stream capture */ \
    if (sh > 64 * 1024 && arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); } \
    hipLaunchKernelGGL(kern, grid, dim3(256), sh, st, a);                                         \
  } while (0)
  if (co_b == 64 && ci_b == 64) WG(64, 64);
  else if (co_b == 64 && ci_b == 32) WG(64, 32);
  else if (co_b == 64 && ci_b == 16) WG(64, 16);
  else if (co_b == 32 && ci_b == 64) WG(32, 64);
  else if (co_b == 32 && ci_b == 32) WG(32, 32);
  else if (co_b == 32 && ci_b == 16) WG(32, 16);
  else if (co_b == 16 && ci_b == 64) WG(16, 64);
  else if (co_b == 16 && ci_b == 32) WG(16, 32);
  else WG(16, 16);
#undef WG
  launch_wgrad_reduce(st, ws, (int)chunks, taps, a.CoutPad, a.CinPad, Cout, Cin, dW, accumulate);
  return arco_launch_status();
}

// out[c] (+)= sum over pixels of X[pix][c];  ws holds 1024*C floats
int arco_colsum(const float* X, long ldx, long M, int C, float* ws, float* out, int accumulate, void* stream) {
  ARCO_CHECK_ARG(X && ws && out && M > 0 && C > 0);
  int nblk = (int)((M + 511) / 512); if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(nblk), dim3(256), 1024 * sizeof(float), as_stream(stream), X, ldx, M, C, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), ws, nblk, C, out, accumulate);
  return arco_launch_status();
}
// the same over an f16 tensor (f16 activation storage: bias gradients of the V-Net's convolutions)
int arco_colsum_h(const void* X, long ldx, long M, int C, float* ws, float* out, int accumulate, void* stream) {
  ARCO_CHECK_ARG(X && ws && out && M > 0 && C > 0);
  int nblk = (int)((M + 511) / 512); if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(colsum_partial_kernel<_Float16>, dim3(nblk), dim3(256), 1024 * sizeof(float), as_stream(stream),
                     reinterpret_cast<const _Float16*>(X), ldx, M, C, ws);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), ws, nblk, C, out, accumulate);
  return arco_launch_status();
}

int arco_transpose2d(const float* x, long ldx, int rows, int cols, float* y, long ldy, void* stream) {
  ARCO_CHECK_ARG(rows > 0 && cols > 0);
  hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, as_stream(stream),
                     x, ldx, rows, cols, y, ldy);
  return arco_launch_status();
}

}  // extern "C"
