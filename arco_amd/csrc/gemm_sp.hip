// Software-pipelined split-bf16 GEMM for the wide 1x1 / 1x1x1 convolutions (FeatureExtractor model_2D.py:20-55 / model_3D.py:20-63,
// q_representation train_arco_2d.py:231-234; forward and, with transposed packs, the data gradient): the arithmetic of
// igemm_kernel<1,...,MMA=3> (igemm.hip: six v_mfma_f32_16x16x32_bf16 per fp32-accurate product in a fixed order, D = W . X^T
// tiles, fp32 accumulate) - outputs bit-identical - restructured the way conv3x3_sp_kernel restructured the 3x3 kernel:
//
//   * persistent workgroups (one per CU, XCD-contiguous tile ranges: the N-tiles of one 64-pixel strip meet in one L2) walking a
//     flattened (tile, 32-k chunk) sequence: a tile's first loads run under the previous tile's last chunks, its output stores
//     drain under the next tile's first chunk;
//   * 8 waves, two roles: waves 0-3 read fragments and issue MFMAs (64 pixels x 64 channels each: 4 x 4 MFMA tiles, 96 MFMAs
//     per chunk), waves 4-7 load the next-but-one chunk (activations fp32, weights pre-split) into registers, split the next
//     chunk's activations into their three bf16 planes and write both operands to the OTHER LDS buffer.  igemm_kernel does both
//     on four waves with two barriers per chunk: its staging and its matrix phases do not overlap (140 TFLOP/s at M = 10^6,
//     N = K = 496);
//   * ONE raw s_barrier per chunk (no vmcnt drain: the loads of chunk c + 2 stay in flight across it).
//
// Tile 64 pixels x 256 channels: every staged activation element (split once per tile and chunk) feeds 16 MFMA columns - the
// split is VALU work on the MFMA waves' SIMDs and does NOT overlap with them (ablation, profiles/r05_notes.md section 2).
// LDS: 2 x (64 + 256) rows x 56 dwords = 143,360 bytes.  Shapes taken (gemm_sp_dispatch): split-bf16 mode, K % 4 == 0, aligned
// rows, no BN statistics / split-K / batching, enough tiles for every CU; everything else stays on igemm_kernel.
#include "igemm_args.h"
#include <stdlib.h>
#include <type_traits>

namespace {
constexpr int BM = 64, BN = 256, KC = 32, LDK = 56;
constexpr int A_DW = BM * LDK, B_DW = BN * LDK, BUF_DW = A_DW + B_DW;
constexpr int NA = BM * (KC / 4) / 256;           // 16-byte activation pieces per producer thread and chunk (2)
constexpr int NB = BN * 12 / 256;                 // 16-byte weight pieces (12): a staged row is 2 groups x 3 planes x 8 dwords
constexpr int A_T = 4, C_T = 4;

// a wave-uniform pointer moved into SGPRs: the scalar-base operand of a global load must not sit in VGPRs
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& x, const bf16x8& y) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}
}  // namespace

template <bool UPRES>      // UPRES: the residual is the trilinear upsample of a.Rup, sampled in the epilogue (its own instantiation:
                           // in one kernel the sampling code cost the plain GEMM 92 bytes of scratch in its main loop)
__global__ __launch_bounds__(512) void gemm_sp_kernel(IgemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const S = reinterpret_cast<unsigned*>(smem);
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const bool producer = threadIdx.x >= 256;
  const int nchunks = (a.Kpad + KC - 1) / KC;
  const int total_tiles = a.n_mblocks * a.n_nblocks;
  // XCD-aware tile order (as conv3x3_sp_kernel): XCD x gets the x-th contiguous eighth of the tiles
  const bool xcd_map = (gridDim.x & 7) == 0;
  const int G8 = xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int T8 = xcd_map ? (total_tiles + 7) >> 3 : total_tiles;
  const int tile0 = xcd_map ? ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int tile_end = xcd_map ? min(total_tiles, (((int)blockIdx.x & 7) + 1) * T8) : total_tiles;
  const int my_tiles = tile0 < tile_end ? (tile_end - tile0 + G8 - 1) / G8 : 0;
  const int total_gc = my_tiles * nchunks;
  if (my_tiles == 0) return;

  if (producer) {
    // ------------------------------------------------------------------ producer waves
    // launch-constant geometry of this thread's pieces: LDS offset, BYTE offset from the chunk's base (the loads take a scalar
    // base + a 32-bit vector offset: no per-piece address arithmetic in the loop - the loader waves share their SIMDs' issue
    // slots with the MFMA waves, and fp32-type VALU work barely overlaps with MFMAs: profiles/history_r01_r03.md)
    unsigned voffA[NA]; int ldsA[NA], rowA[NA], kA[NA];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = tid + it * 256, q = idx & 7, r_ = idx >> 3;
      const int row = (r_ & ~3) | ((r_ & 1) << 1) | ((r_ >> 1) & 1);      // rows r, r + 2 per ds_write_b64 group (igemm.hip: no bank overlap)
      rowA[it] = row; kA[it] = 4 * q;
      voffA[it] = (unsigned)((row * a.lda + 4 * q) * 4);
      ldsA[it] = row * LDK + (q >> 2) * 24 + (q & 3) * 2;
    }
    unsigned voffB[NB]; int ldsB[NB], rowB[NB], kB[NB];
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int idx = tid + it * 256, row = idx / 12, q = idx - row * 12;
      rowB[it] = row; kB[it] = (q / 6) * 16;
      voffB[it] = (unsigned)((row * a.Kg * 24 + 4 * q) * 4);
      ldsB[it] = A_DW + row * LDK + 4 * q;
    }
    struct Desc { int j, c; long m0; int n0; };
    auto decode = [&](Desc& d) {
      const int v = tile0 + d.j * G8;
      const int mblk = v / a.n_nblocks;
      d.m0 = (long)mblk * BM; d.n0 = (v - mblk * a.n_nblocks) * BN;
    };
    auto advance = [&](Desc& d) { if (++d.c == nchunks) { d.c = 0; ++d.j; decode(d); } };
    // Every wave issues NA + NB loads per chunk, real or not (asm: the compiler neither tracks nor waits for them - it had
    // reused their destination registers as temporaries at the loop head and put a vmcnt(5) there), so `s_waitcnt vmcnt(NA +
    // NB)` in front of a set's first use is exact: only the other set's loads are younger.  Interior chunks (the tile inside
    // the matrix in M, N and K: wave-uniform) load and store without masks; edge chunks load a clamped offset where a piece
    // lies outside and zero it at the split.
    // Rows past M (activations) or past Npad (weights) are never stored by the epilogue, so such pieces only need a valid
    // ADDRESS (offset 0), not zeros; only the K edge (k >= K in the last chunk) must read as zero.  The last N-tile's clamped
    // weight offsets are launch constants (N = 496: every second tile is that tile).
    unsigned voffBl[NB];
    const int n_last0 = (a.n_nblocks - 1) * BN;
#pragma unroll
    for (int it = 0; it < NB; ++it) voffBl[it] = n_last0 + rowB[it] < a.Npad ? voffB[it] : 0u;
    f32x4 ra[3][NA]; u32x4 rb[3][NB]; unsigned zA[3] = {0, 0, 0}; bool kedge[3] = {false, false, false};
    auto load = [&](auto SET_, const Desc& d, bool real) {
      constexpr int SET = decltype(SET_)::value;
      const int kc0 = d.c * KC;
      const float* Ab = uniform_ptr(real ? a.A + d.m0 * a.lda + kc0 : a.A);
      const float* Bb = uniform_ptr(real ? a.Wp + ((long)d.n0 * a.Kg + (kc0 >> 4)) * 24 : a.Wp);
      const bool mfull = d.m0 + BM <= a.M, kfull = kc0 + KC <= a.K, nlast = d.n0 == n_last0;
      kedge[SET] = real && !kfull;
#ifdef GSP_NO_LOAD
#pragma unroll
      for (int it = 0; it < NA + NB; ++it) asm volatile("s_nop 0" ::: "memory");
      return;
#endif
      if (real && mfull && kfull) {
#pragma unroll
        for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[SET][it]) : "v"(voffA[it]), "s"(Ab) : "memory");
      } else {
        unsigned z = 0;
#pragma unroll
        for (int it = 0; it < NA; ++it) {
          const bool kout = kc0 + kA[it] >= a.K;
          z |= kout ? (1u << it) : 0u;
          const unsigned o = (real && d.m0 + rowA[it] < a.M && !kout) ? voffA[it] : 0u;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[SET][it]) : "v"(o), "s"(Ab) : "memory");
        }
        zA[SET] = z;
      }
      if (!real) {
#pragma unroll
        for (int it = 0; it < NB; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[SET][it]) : "v"(0u), "s"(Bb) : "memory");
      } else if (!nlast) {
#pragma unroll
        for (int it = 0; it < NB; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[SET][it]) : "v"(voffB[it]), "s"(Bb) : "memory");
      } else {
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const unsigned o = voffBl[it];
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[SET][it]) : "v"(o), "s"(Bb) : "memory");
        }
      }
    };
    auto store = [&](auto SET_, unsigned* buf, auto YOUNGER_) {
      constexpr int SET = decltype(SET_)::value;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(YOUNGER_)::value) : "memory");
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[SET][it])::"memory");
#pragma unroll
      for (int it = 0; it < NB; ++it) asm volatile("" : "+v"(rb[SET][it])::"memory");
#ifdef GSP_NO_SPLIT
      return;
#endif
      if (kedge[SET]) {
#pragma unroll
        for (int it = 0; it < NA; ++it) if ((zA[SET] >> it) & 1u) ra[SET][it] = f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        u32x2 p0, p1, p2;
        split3_bf16x4(ra[SET][it], p0, p1, p2);
        unsigned* d = buf + ldsA[it];
        *reinterpret_cast<u32x2_ma*>(d) = p0; *reinterpret_cast<u32x2_ma*>(d + 8) = p1; *reinterpret_cast<u32x2_ma*>(d + 16) = p2;
      }
#pragma unroll
      for (int it = 0; it < NB; ++it) *reinterpret_cast<u32x4_ma*>(buf + ldsB[it]) = rb[SET][it];
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    Desc d{0, 0, 0, 0};
    decode(d);
    // THREE register sets: the loads of chunk gc + 3 are requested in iteration gc and used in iteration gc + 2 - two chunk
    // times (~2 us) between request and use.  With two sets (one chunk time) the producers waited for memory in every iteration:
    // the loader alone took 1.4 us per chunk, the MFMA waves alone 1.06 (ablation builds, profiles/r05_notes.md section 2).
    // prologue: chunk 0 -> LDS buffer 0, chunks 1, 2 -> register sets 1, 2
    using NONE = std::integral_constant<int, 0>; using TWOSETS = std::integral_constant<int, 2 * (NA + NB)>;
    load(S0{}, d, true);
    store(S0{}, S, NONE{});
    advance(d);
    load(S1{}, d, total_gc > 1);
    advance(d);
    load(S2{}, d, total_gc > 2);
    advance(d);
    step_barrier();
    // iteration gc: the consumers multiply chunk gc (buffer gc & 1); here chunk gc + 3 is requested into the set chunk gc has
    // left, and chunk gc + 1 (set (gc + 1) % 3, requested two iterations ago) is split and written to buffer (gc + 1) & 1,
    // which the consumers left at the last barrier
    auto iter = [&](auto SETN_, int gc) {
      constexpr int SETN = decltype(SETN_)::value;          // set of chunk gc + 1; chunk gc + 3 goes to set (gc + 3) % 3 = SETN + 2
      using SN = std::integral_constant<int, SETN>; using SF = std::integral_constant<int, (SETN + 2) % 3>;
      load(SF{}, d, gc + 3 < total_gc);
      advance(d);
      if (gc + 1 < total_gc) store(SN{}, S + ((gc + 1) & 1) * BUF_DW, TWOSETS{});
      step_barrier();
    };
    for (int gc = 0; gc < total_gc; gc += 3) {
      iter(S1{}, gc);
      if (gc + 1 < total_gc) iter(S2{}, gc + 1);
      if (gc + 2 < total_gc) iter(S0{}, gc + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  // Fragment reads are software-pipelined by hand (left alone hipcc sinks every ds_read to just above its first use: the LDS
  // latency - four waves hit the LDS at once behind each barrier - was exposed five times per chunk, 0.9 us of a 1.7 us chunk):
  //   * B fragments of column group ct + 1 are requested in front of group ct's 24 MFMAs (two register sets);
  //   * the chunk's barrier sits in front of its LAST column group: by then every fragment of the chunk is in registers, so
  //     the producers may refill this buffer, and the other buffer (chunk gc + 1) is complete - its A fragments and first B
  //     group are requested right behind the barrier, under the last 24 MFMAs (two A register sets, alternating per chunk).
  constexpr int wm = 0;                             // every MFMA wave takes all 64 rows and its own 64 of the 256 columns
  const int wn = wid;
  const int koff = (g >> 1) * 24 + (g & 1) * 4;
  const int laneA = (wm * A_T * 16 + li) * LDK + koff;
  const int laneB = A_DW + (wn * C_T * 16 + li) * LDK + koff;
  f32x4 acc[A_T][C_T];
  bf16x8 fa[2][A_T][3], fb[2][3];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_T; ++i)
#pragma unroll
      for (int k = 0; k < C_T; ++k) acc[i][k] = f32x4{0, 0, 0, 0};
  };
  auto read_A = [&](auto P_, const unsigned* buf) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
#ifdef GSP_NO_FRAG
    return;
#endif
#pragma unroll
    for (int at = 0; at < A_T; ++at)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[P][at][pl] = lds_bf16x8(buf + laneA + at * 16 * LDK + 8 * pl);
  };
  auto read_B = [&](auto F_, const unsigned* buf, int ct) __attribute__((always_inline)) {
    constexpr int F = decltype(F_)::value;
#ifdef GSP_NO_FRAG
    return;
#endif
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fb[F][pl] = lds_bf16x8(buf + laneB + ct * 16 * LDK + 8 * pl);
  };
  auto group = [&](auto P_, auto F_, auto CT_) __attribute__((always_inline)) {         // D = W . X^T; small terms first (the order of igemm_kernel's SWP path)
    constexpr int P = decltype(P_)::value, F = decltype(F_)::value, ct = decltype(CT_)::value;
#ifdef GSP_NO_MFMA      // timing-only ablation build: the fragments stay "used"
#pragma unroll
    for (int at = 0; at < A_T; ++at)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" ::"v"(fa[P][at][pl]), "v"(fb[F][pl]));
    return;
#endif
#pragma unroll
    for (int at = 0; at < A_T; ++at) {
      mfma_acc(acc[at][ct], fb[F][0], fa[P][at][2]);
      mfma_acc(acc[at][ct], fb[F][2], fa[P][at][0]);
      mfma_acc(acc[at][ct], fb[F][1], fa[P][at][1]);
      mfma_acc(acc[at][ct], fb[F][0], fa[P][at][1]);
      mfma_acc(acc[at][ct], fb[F][1], fa[P][at][0]);
      mfma_acc(acc[at][ct], fb[F][0], fa[P][at][0]);
    }
  };
  auto epilogue = [&](int j) __attribute__((always_inline)) {
    {
        // epilogue of tile j (registers only: the producers are already filling the buffers for tile j + 1)
        const int v = tile0 + j * G8;
        const int mblk = v / a.n_nblocks;
        const long m0 = (long)mblk * BM; const int n0 = (v - mblk * a.n_nblocks) * BN;
        if constexpr (UPRES) {
          // residual = trilinear(align_corners) sample of the low-resolution tensor at this lane's four voxels: the index math and
          // the lerp expression of trilinear_fwd_kernel (elementwise.hip), so that out == conv + trilinear(...) bit for bit
          const float sd = a.oD > 1 ? (float)(a.uD - 1) / (float)(a.oD - 1) : 0.f, sh = a.oH > 1 ? (float)(a.uH - 1) / (float)(a.oH - 1) : 0.f,
                      sw = a.oW > 1 ? (float)(a.uW - 1) / (float)(a.oW - 1) : 0.f;
#pragma unroll
          for (int at = 0; at < A_T; ++at) {
            const long m = m0 + (wm * A_T + at) * 16 + li;
            const bool mok = m < a.M;
            long r = mok ? m : 0;
            const int xo = (int)(r % a.oW); r /= a.oW; const int yo = (int)(r % a.oH); r /= a.oH; const int zo = (int)(r % a.oD); const long n = r / a.oD;
            int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
            ac_src(zo, sd, a.uD, z0, z1, lz); ac_src(yo, sh, a.uH, y0, y1, ly); ac_src(xo, sw, a.uW, x0, x1, lx);
            const float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
            const float* b = a.Rup + (n * a.uD) * (long)a.uH * a.uW * a.ldrup;
#define TLP(zz, yy, xx) (b + (((long)(zz) * a.uH + (yy)) * a.uW + (xx)) * a.ldrup)
            const float *p000 = TLP(z0, y0, x0), *p001 = TLP(z0, y0, x1), *p010 = TLP(z0, y1, x0), *p011 = TLP(z0, y1, x1);
            const float *p100 = TLP(z1, y0, x0), *p101 = TLP(z1, y0, x1), *p110 = TLP(z1, y1, x0), *p111 = TLP(z1, y1, x1);
#undef TLP
#pragma unroll
            for (int ct = 0; ct < C_T; ++ct) {
              const int nb = n0 + (wn * C_T + ct) * 16 + 4 * g;
              const int nc = nb < a.N ? nb : 0;
              const f32x4 v000 = *reinterpret_cast<const f32x4*>(p000 + nc), v001 = *reinterpret_cast<const f32x4*>(p001 + nc);
              const f32x4 v010 = *reinterpret_cast<const f32x4*>(p010 + nc), v011 = *reinterpret_cast<const f32x4*>(p011 + nc);
              const f32x4 v100 = *reinterpret_cast<const f32x4*>(p100 + nc), v101 = *reinterpret_cast<const f32x4*>(p101 + nc);
              const f32x4 v110 = *reinterpret_cast<const f32x4*>(p110 + nc), v111 = *reinterpret_cast<const f32x4*>(p111 + nc);
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                o[e] = tl_blend1(v000[e], v001[e], v010[e], v011[e], v100[e], v101[e], v110[e], v111[e], hx, lx, hy, ly, hz, lz);
              f32x4 bv = f32x4{0, 0, 0, 0};
              if (a.bias && nb < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + nb);
              const f32x4 out = (acc[at][ct] + bv) + o;
              if (mok && nb < a.N) *reinterpret_cast<f32x4*>(a.C + m * a.ldc + nb) = out;
            }
          }
        } else {      // (gemm_sp_dispatch only takes 16-byte-aligned outputs / residuals with N % 4 == 0: pieces are whole or absent)
          // 16-byte pieces, whole or absent: the residual loads of all 16 sub-tiles are issued back to back (clamped address
          // where the piece is outside the matrix), then added and stored - one memory round trip per tile, not sixteen
#pragma unroll
          for (int h = 0; h < 2; ++h) {             // two halves of the column groups: 8 residual pieces in flight per lane
            f32x4 res[A_T][2];
            if (a.R) {
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int at = 0; at < A_T; ++at) {
                  const int nb = n0 + (wn * C_T + 2 * h + c2) * 16 + 4 * g;
                  const long m = m0 + (wm * A_T + at) * 16 + li;
                  const bool ok = m < a.M && nb < a.N;
                  res[at][c2] = *reinterpret_cast<const f32x4*>(ok ? a.R + m * a.ldr + nb : a.R);
                }
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
              const int ct = 2 * h + c2;
              const int nb = n0 + (wn * C_T + ct) * 16 + 4 * g;
              f32x4 bv = f32x4{0, 0, 0, 0};
              if (a.bias && nb < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + nb);
#pragma unroll
              for (int at = 0; at < A_T; ++at) {
                const long m = m0 + (wm * A_T + at) * 16 + li;
                f32x4 o = acc[at][ct] + bv;
                if (a.R) o += res[at][c2];
                if (m < a.M && nb < a.N) *reinterpret_cast<f32x4*>(a.C + m * a.ldc + nb) = o;
              }
            }
          }
        }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  step_barrier();                                   // the producers' prologue barrier: chunk 0 is in buffer 0
  read_A(I0{}, S);
  read_B(I0{}, S, 0);
  zero_acc();
  int jt = 0, ct_in_tile = 0;                       // tile ordinal and chunk index inside it
  auto chunk = [&](auto P_, int gc) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    using PP = std::integral_constant<int, P>; using QQ = std::integral_constant<int, P ^ 1>;
    const unsigned* cur = S + (gc & 1) * BUF_DW;
    const unsigned* nxt = S + ((gc + 1) & 1) * BUF_DW;
    read_B(I1{}, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    group(PP{}, I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_B(I0{}, cur, 2);
    __builtin_amdgcn_sched_barrier(0);
    group(PP{}, I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    read_B(I1{}, cur, 3);
    __builtin_amdgcn_sched_barrier(0);
    group(PP{}, I0{}, I2{});
    __builtin_amdgcn_sched_barrier(0);
    // every fragment of this chunk has landed (the builtin wait: hipcc models it and does not re-wait for the reads above
    // behind the fifteen new ones below); the last chunk reads a stale buffer, harmlessly - no branch, no merged wait state
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_A(QQ{}, nxt);
    read_B(I0{}, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    group(PP{}, I1{}, I3{});
    __builtin_amdgcn_sched_barrier(0);
    if (++ct_in_tile == nchunks) {
      epilogue(jt);
      zero_acc();
      ct_in_tile = 0; ++jt;
    }
  };
  for (int gc = 0; gc < total_gc; gc += 2) {
    chunk(I0{}, gc);
    if (gc + 1 < total_gc) chunk(I1{}, gc + 1);
  }
}

// -1: the shape is not taken (the caller falls through to igemm_kernel); q != nullptr: query only
static int& gemm_sp_flag() { static int on = !(getenv("ARCO_GEMM_SP") && atoi(getenv("ARCO_GEMM_SP")) == 0); return on; }
static long& gemm_sp_min_tiles() { static long t = getenv("ARCO_GEMM_SP_TILES") ? atol(getenv("ARCO_GEMM_SP_TILES")) : 2048; return t; }
// A/B switch (tests, tools): on = 0 keeps every GEMM on igemm_kernel; min_tiles > 0 also sets the tile-count threshold.  Returns the previous `on`.
extern "C" int arco_gemm_sp_set(int on, long min_tiles) {
  const int prev = gemm_sp_flag();
  gemm_sp_flag() = on ? 1 : 0;
  if (min_tiles > 0) gemm_sp_min_tiles() = min_tiles;
  return prev;
}

int gemm_sp_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  const int on = gemm_sp_flag();
  const long min_tiles = gemm_sp_min_tiles();
  if (!on || a.mma != 3 || a.stat_sum || a.ksplit > 1 || a.batch > 1 || a.stat_groups > 1) return -1;
  if ((a.K & 3) != 0 || (a.lda & 3) != 0 || (reinterpret_cast<uintptr_t>(a.A) & 15) != 0) return -1;
  if ((a.N & 3) != 0 || (a.ldc & 3) != 0 || (reinterpret_cast<uintptr_t>(a.C) & 15) != 0) return -1;
  if (a.R && ((a.ldr & 3) != 0 || (reinterpret_cast<uintptr_t>(a.R) & 15) != 0)) return -1;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) return -1;
  if (a.Rup && (a.R || (a.ldrup & 3) != 0 || (reinterpret_cast<uintptr_t>(a.Rup) & 15) != 0 ||
                a.M % ((long)a.oD * a.oH * a.oW) != 0)) return -1;
  if (a.Npad < 64 || a.Kpad < 32) return -1;
  const int mblocks = (int)((a.M + BM - 1) / BM), nblocks = (a.Npad + BN - 1) / BN;
  const long tiles = (long)mblocks * nblocks;
  if (tiles < min_tiles) return -1;
  // N-tile padding: 128-wide tiles waste (nblocks * 128 - Npad) / (nblocks * 128) of the matrix work (N = 496: 3 %, 224: 12 %,
  // 192: 25 %, 240: 6 %); the 64 x 224 igemm tile keeps the narrow ones
  if (nblocks * BN - a.Npad > nblocks * BN / 5) return -1;
  if (q) { q[0] = mblocks; q[1] = 1000000 + BM * 1000 + BN + 400000; q[2] = KC * 100 + 10; return ARCO_OK; }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0; hipDeviceProp_t p;
    (void)hipGetDevice(&dev);
    n_cu = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256;
  }
  static unsigned long long attr_set = 0;       // (one bit per device: a process may drive several, common.h)
  if (arco_first_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_DW * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_DW * 4);
  }
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = nblocks;
  const int grid = (int)(tiles < n_cu ? tiles : n_cu);
  if (a.Rup) hipLaunchKernelGGL(gemm_sp_kernel<true>, dim3(grid), dim3(512), 2 * BUF_DW * 4, st, b);
  else hipLaunchKernelGGL(gemm_sp_kernel<false>, dim3(grid), dim3(512), 2 * BUF_DW * 4, st, b);
  return arco_launch_status();
}
