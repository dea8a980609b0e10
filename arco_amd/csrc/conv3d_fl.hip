// Software-pipelined split-bf16 3x3x3 convolution for the V-Net levels below full resolution (ConvBlock / the stages of the
// up path, vnetWithArgs.py:5-31, 222-262: 32 -> 32 on 56 x 40 planes, 64 -> 64 on 28 x 20, 128 -> 128 on 14 x 10, 256 -> 256 on
// 7 x 5 at the LA patch; forward and, with flipped + transposed packed weights, the data gradient).
// (-DARCO_FC_CLOCK: a DIAGNOSTIC build of conv3d_fc_kernel with s_memtime stamps into a buffer of their own - tools/micro/fc_clock.py; never in the
//  shipped library.)
//
// Same arithmetic as igemm_kernel<9,...,FLAT,DEPTH=3,MMA=3> (igemm.hip), operation for operation - the outputs are bit-identical,
// asserted in tests/test_conv3d_fl_gpu.py - in the execution structure of conv3x3_sp_kernel (conv_sp.hip):
//   * a 3x3x3 convolution over K channels IS a 3x3 convolution over 3 K "virtual" channels: virtual chunk vc = (dz, 16-channel
//     chunk c) reads its activation tile from plane p + dz - 1 (zeros outside the volume) and its weights from taps 9 dz .. 9 dz + 8
//     of the 27-tap pack.  A tile therefore runs 3 K / 16 chunks (6 at 32 channels, 48 at 256) through one accumulator set:
//     the per-tile prologue / epilogue the 2-D kernels pay every 2-16 chunks is paid every 6-48.
//   * FLAT tiles: the planes of these levels are 40 / 20 / 10 / 5 pixels wide - no multiple of the 16-pixel MFMA tile.  A tile is
//     BM = 64 A_T consecutive positions of the plane laid out with a padded row stride Wp = W + 2 (one zero column each side),
//     so a tap is the uniform shift dy Wp + dx of the position and the waste is 2 / (W + 2) + the plane's ragged last tile
//     (8 % at 56 x 40 with 128-position tiles; 16 x 16-pixel rectangles would waste 20 %, 60 % at 28 x 20).
//   * one persistent workgroup per CU, 8 waves in two roles: waves 0-3 read fragments and issue MFMAs, waves 4-7 produce - the
//     pre-split weights by LDS-DMA into a ring of five tap-pair slots, the activations by asm loads two chunks ahead, split into
//     their three bf16 planes on the way into the second A buffer; counted s_waitcnt vmcnt(N) + raw s_barrier per step.
//   * the geometry of a flat tile (which staged rows are pixels, where they live) depends on the tile's first position: the
//     producers recompute their piece offsets and the validity mask once per TILE (3 K / 16 chunks), not per chunk.
//   * D = W . X^T: a lane ends with 4 consecutive output channels of one position -> 16-byte stores; BatchNorm partial sums by
//     DPP row reductions, one slab per consumer wave.
// LDS: [2][BM + 2 * 63 + 2][24] A planes + [5][slot] B ring + bias (A_T = 4, C_T = 4: 73,728 + 61,440 + 1,024 bytes).
#include "sp_util.h"

__device__ __attribute__((aligned(64))) unsigned int conv3d_fl_zero_row[32];      // zero-initialised: the "tap 9" weights of a slice
#ifdef ARCO_FC_CLOCK
__device__ unsigned long long* arco_fc_clock_buf = nullptr;       // (diagnostic build: stamps go to a buffer of their own)
#endif

template <int A_T, int C_T>
struct FlGeom {
  static constexpr int BM = 64 * A_T, WPMAX = 63, AROWS = BM + 2 * WPMAX + 2, BN = C_T * 16;
  static constexpr int A_DW = AROWS * 24;                       // dwords per A buffer
  static constexpr int NBI = (2 * BN * 6 + 255) / 256;          // LDS-DMA instructions per thread and slot
  static constexpr int SLOT_DW = NBI * 256 * 4;                 // slot stride in dwords (whole wave-instructions)
  static constexpr int NA_IT = (AROWS * 4 + 255) / 256;         // 16-byte activation loads per thread and chunk
  static constexpr int BIAS_DW = 256;
  static constexpr size_t LDS_BYTES = (size_t)(2 * A_DW + 5 * SLOT_DW + BIAS_DW) * 4;
};

template <int A_T, int C_T>
__global__ __launch_bounds__(512) void conv3d_fl_kernel(IgemmArgs a) {
  using G = FlGeom<A_T, C_T>;
  constexpr int BM = G::BM, BN = G::BN, NA = G::NA_IT, NB = G::NBI;
  constexpr int NA4 = (NA + 3) / 4;                              // activation pieces split + written per staging step
  constexpr int NL = NA;                                         // VMEM loads per chunk and wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const As = reinterpret_cast<unsigned*>(smem);
  unsigned* const Bs = As + 2 * G::A_DW;
  float* const bias_s = reinterpret_cast<float*>(Bs + 5 * G::SLOT_DW);

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const bool producer = threadIdx.x >= 256;
  const int Wp = a.W + 2, npos = a.H * Wp, per_plane = (npos + BM - 1) / BM;
  const int arows = BM + 2 * Wp + 2;                             // staged rows of a tile: its positions + one padded row + 1 either side
  const int nk = a.K >> 4, nvc = 3 * nk;                         // virtual chunks per tile: (dz, 16-channel chunk)
  const int total_tiles = a.n_mblocks * a.n_nblocks;
  // XCD-aware tile order (see conv3x3_sp_kernel): XCD x gets the x-th contiguous eighth of the tiles
  const bool xcd_map = (gridDim.x & 7) == 0;
  const int G8 = xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int T8 = xcd_map ? (total_tiles + 7) >> 3 : total_tiles;
  const int tile0 = xcd_map ? ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int tile_end_x = xcd_map ? min(total_tiles, (((int)blockIdx.x & 7) + 1) * T8) : total_tiles;
  const int my_tiles = tile0 < tile_end_x ? (tile_end_x - tile0 + G8 - 1) / G8 : 0;
  const int total_gc = my_tiles * nvc;
  if (my_tiles == 0) return;
  const bool has_stats = a.stat_sum != nullptr;

  // chunk descriptors (tile ordinal j of this workgroup, virtual chunk vc = dz * nk + kc), advanced incrementally
  struct Desc { int j, vc, dz, kc, img, f0, nblk, mblk, pl; };
  auto decode = [&](Desc& d) {
    const int v = tile0 + d.j * G8;
    d.mblk = v / a.n_nblocks; d.nblk = v - d.mblk * a.n_nblocks;
    d.img = d.mblk / per_plane;                      // plane index over all volumes
    d.f0 = (d.mblk - d.img * per_plane) * BM;        // first padded-plane position of the tile
    d.pl = d.img % a.D3;                             // plane within its volume
  };
  auto advance = [&](Desc& d) {
    ++d.vc;
    if (++d.kc == nk) { d.kc = 0; ++d.dz; }
    if (d.vc == nvc) { d.vc = 0; d.dz = 0; d.kc = 0; ++d.j; decode(d); }
  };
  Desc d0{0, 0, 0, 0, 0, 0, 0, 0, 0};
  decode(d0);

  if (producer) {
    // ================================================================ producer waves
    Desc d1 = d0; advance(d1);
    Desc d2 = d1; advance(d2);
    Desc d3 = d2; advance(d3);
    const int qA = tid & 3;
    const float inv_wp = 1.0f / (float)Wp;
    const long plane_dw = (long)a.H * a.W * a.lda;   // floats per plane
    // per-thread LDS offsets of the NA activation pieces (fixed for the launch); their source offsets within a plane and the mask
    // of the pieces that are pixels (not padding, not outside the plane) belong to a TILE: tile_geom() at a tile's first load
    int ldsA[NA]; unsigned voff[NA]; unsigned okm_t = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int row = (tid + it * 256) >> 2;
      ldsA[it] = row < arows ? row * 24 + qA * 2 : -1;
      voff[it] = (unsigned)(qA * 16);
    }
    auto tile_geom = [&](const Desc& d) {
      okm_t = 0;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int row = (tid + it * 256) >> 2;
        const int pidx = d.f0 + row - 1;             // position in the plane padded by one row / column each side
        const int py = (int)(((float)pidx + 0.5f) * inv_wp), px = pidx - py * Wp;       // (exact: pidx < 2^16, Wp <= 63)
        const bool ok = row < arows && pidx >= 0 && py >= 1 && py <= a.H && px >= 1 && px <= a.W;
        voff[it] = ok ? (unsigned)((((py - 1) * a.W + px - 1) * (int)a.lda + qA * 4) * 4) : (unsigned)(qA * 16);
        okm_t |= ok ? (1u << it) : 0u;
      }
    };
    int woff[NB], wq[NB], wtap[NB];               // weight piece of LDS-DMA instruction i: dword offset, piece of the row, local tap (2 = padding)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = (i * 4 + wid) * 64 + lane;
      const int tl_ = p / (BN * 6), rem = p - tl_ * (BN * 6), n = rem / 6, q6 = rem - n * 6;
      wtap[i] = tl_ < 2 ? tl_ : 2;
      woff[i] = tl_ < 2 ? ((tl_ * a.Npad + n) * a.Kg) * 24 + q6 * 4 : 0;
      wq[i] = q6 * 4;
    }
    const long wslot2 = (long)2 * a.Npad * a.Kg * 24;        // two taps of packed weights (dwords)
    const long wslice = (long)9 * a.Npad * a.Kg * 24;        // one depth slice (9 taps)
    const float* const zrow = reinterpret_cast<const float*>(conv3d_fl_zero_row);

    // Activation loads of a chunk: asm statements (hipcc, which drains the LDS-DMA queue at the use of any load it knows of, does
    // not see them), two register sets, loaded TWO chunks ahead.  Every lane loads: pieces that are padding, and every piece of a
    // chunk whose plane lies outside the volume, load the plane's first pixel and are zeroed at the split.
    f32x4 ra[2][NA]; unsigned okm2[2] = {0, 0};
    auto load_A = [&](const Desc& d, bool real, int set) {
      if (d.vc == 0) tile_geom(d);                   // (wave-uniform; the loads of a tile's chunks are issued in order)
      const int pz = d.pl + d.dz - 1;
      const bool pok = real && pz >= 0 && pz < a.D3;
      const float* gbase = uniform_ptr(a.A + (long)(pok ? d.img + d.dz - 1 : 0) * plane_dw + d.kc * 16);
      okm2[set] = pok ? okm_t : 0u;
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(voff[it]), "s"(gbase) : "memory");
    };
    auto ra_fence = [&](int set) {
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[set][it])::"memory");
    };
    auto store_A = [&](unsigned* buf, int it, int set) {    // split piece `it` into its three bf16 planes (zero outside the plane)
      const f32x4 v = ((okm2[set] >> it) & 1u) ? ra[set][it] : f32x4{0, 0, 0, 0};
      u32x2 p0, p1, p2;
      split3_bf16x4(v, p0, p1, p2);
      if (ldsA[it] >= 0) {
        unsigned* d = buf + ldsA[it];
        *reinterpret_cast<u32x2_ma*>(d) = p0; *reinterpret_cast<u32x2_ma*>(d + 8) = p1; *reinterpret_cast<u32x2_ma*>(d + 16) = p2;
      }
    };
    auto refill = [&](int slot, const Desc& d) {   // LDS-DMA: weights of step `slot` of chunk d -> ring slot
      const float* base = a.Wp + d.dz * wslice + slot * wslot2 + ((long)d.nblk * BN * a.Kg + d.kc) * 24;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool zero = wtap[i] == 2 || (slot == 4 && wtap[i] == 1);
        const float* src = zero ? zrow + wq[i] : base + woff[i];
        unsigned* dst = Bs + slot * G::SLOT_DW + (i * 4 + wid) * 256;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
      }
    };

    // prologue.  Per-wave VMEM order: A0 | R0 R1 R2 | R3 | A1 A2  (the steady state's "... R3' A+3" tail)
    for (int i = tid; i < G::BIAS_DW; i += 256) bias_s[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    load_A(d0, true, 0);
    refill(0, d0); refill(1, d0); refill(2, d0);
    wait_vm<3 * NB>();                // chunk 0's activations (the DMA behind them stays in flight)
    ra_fence(0);
#pragma unroll
    for (int it = 0; it < NA; ++it) store_A(As, it, 0);
    refill(3, d0);
    load_A(d1, total_gc > 1, 1);      // (every wave issues every instruction, real or not: the counts below are exact)
    load_A(d2, total_gc > 2, 0);
    wait_vm<NL>();                    // chunk 1's activations (split from step 0 on) and with them slots 0-3
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    auto pchunk = [&](auto SET_, int gc) {
      constexpr int SET = decltype(SET_)::value;          // register set of chunk gc + 1 (= its parity)
      const bool more = gc + 1 < total_gc;
      unsigned* Anxt = As + ((gc + 1) & 1) * G::A_DW;
      auto step = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        // VMEM instructions younger than what this barrier needs; order per chunk:
        // s0 R4 | s1 R0' | s2 R1' | s3 R2' | s4 R3' A+3   (the last chunks issue the same instructions on dummy addresses)
        constexpr int NS = S <= 2 ? NL + 2 * NB : 2 * NB;
        wait_vm<NS>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        refill(S == 0 ? 4 : S - 1, S == 0 ? d0 : (more ? d1 : d0));     // the slot the previous step has finished with
        if (S == 0) ra_fence(SET);
        if (S <= 3) {                  // a quarter of the next chunk's activation tile per step, the loads in the fifth
#pragma unroll
          for (int it = S * NA4; it < (S + 1) * NA4 && it < NA; ++it) store_A(Anxt, it, SET);
        }
        if (S == 4) load_A(d3, gc + 3 < total_gc, SET);
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      d0 = d1; d1 = d2; d2 = d3; advance(d3);
    };
    for (int gc = 0; gc < total_gc; gc += 2) {
      pchunk(std::integral_constant<int, 1>{}, gc);
      if (gc + 1 < total_gc) pchunk(std::integral_constant<int, 0>{}, gc + 1);
    }
    wait_vm<0>();
    return;
  }

  // ================================================================== consumer waves
  // fragment addressing: lanes g = 0,1 take tap 2s, g = 2,3 tap 2s+1 (step 4: tap 8 and the zero tap); a tap is a row shift
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + tl > 8 ? 8 : 2 * s + tl;
    aoff[s] = ((tap / 3) * Wp + tap % 3) * 24;
  }
  const int laneA = (wid * A_T * 16 + li) * 24 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 24 + (g & 1) * 4;
  bf16x8 fa[2][A_T][3], fb[2][3];
  f32x4 acc[A_T][C_T];
#pragma unroll
  for (int i = 0; i < A_T; ++i)
#pragma unroll
    for (int j = 0; j < C_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

  __builtin_amdgcn_s_barrier();          // the producers' prologue barrier: chunk 0's activations, slots 0 and 1
#pragma unroll
  for (int at = 0; at < A_T; ++at)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[0][at][pl] = lds_bf16x8(As + laneA + aoff[0] + at * 16 * 24 + 8 * pl);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) fb[0][pl] = lds_bf16x8(Bs + laneB + 8 * pl);

  auto chunk = [&](auto CP_, int gc) {
    constexpr int CP = decltype(CP_)::value;             // parity of the chunk = A set holding step 0's fragments
    const unsigned* Acur = As + (gc & 1) * G::A_DW;
    const unsigned* Anxt = As + ((gc + 1) & 1) * G::A_DW;
    auto step = [&](auto S_) {
      constexpr int S = decltype(S_)::value;
      constexpr int P = (CP + S) & 1, Q = P ^ 1;
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      // LDS reads in the shadow of the MFMA chains (see conv3x3_sp_kernel): at the head of every group the three fragments of the
      // NEXT group, and behind the first chains the next step's A fragments (set Q), APC per chain
      constexpr int NCH = A_T * C_T;
      constexpr int APC = (A_T * 3 + (NCH - 2) - 1) / (NCH - 2 > 0 ? NCH - 2 : 1);
      const unsigned* An = S < 4 ? Acur : Anxt;
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct) {
        constexpr int GB = (C_T & 1) ? (CP + S) : 0;       // group parity base (an odd group count flips it per step)
        const int BPc = (GB + ct) & 1, BPn = BPc ^ 1;
        const bool last = ct + 1 == C_T;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fb[BPn][pl] = lds_bf16x8(Bs + (last ? (S + 1) % 5 : S) * G::SLOT_DW + laneB + (last ? 0 : ct + 1) * 16 * 24 + 8 * pl);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int at = 0; at < A_T; ++at) {          // D = W . X^T; small terms first
          mfma_acc(acc[at][ct], fb[BPc][0], fa[P][at][2]);
          mfma_acc(acc[at][ct], fb[BPc][2], fa[P][at][0]);
          mfma_acc(acc[at][ct], fb[BPc][1], fa[P][at][1]);
          mfma_acc(acc[at][ct], fb[BPc][0], fa[P][at][1]);
          mfma_acc(acc[at][ct], fb[BPc][1], fa[P][at][0]);
          mfma_acc(acc[at][ct], fb[BPc][0], fa[P][at][0]);
          const int chain = ct * A_T + at;
#pragma unroll
          for (int k = chain * APC; k < (chain + 1) * APC && k < A_T * 3; ++k)
            fa[Q][k / 3][k % 3] = lds_bf16x8(An + laneA + aoff[(S + 1) % 5] + (k / 3) * 16 * 24 + 8 * (k % 3));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
  };

  const float inv_wp = 1.0f / (float)Wp;
  auto tile_end = [&]() {              // ---- tile done: bias, store, BN partial statistics; accumulators back to zero
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results (asm: no hazard padding by hipcc)
    const int n0 = d0.nblk * BN;
    long pix[A_T];                     // output pixel of this lane's position per MFMA tile (-1: padding column / past the plane)
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
          f32x4 v = acc[at][ct] + bv;
          if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + pix[at] * a.ldr + n);
          *reinterpret_cast<f32x4*>(a.C + pix[at] * a.ldc + n) = v;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
        }
        acc[at][ct] = f32x4{0, 0, 0, 0};
      }
    }
    if (has_stats) {       // one partial per WAVE (4 slabs per tile): no LDS round trip, no barrier inside the pipeline
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

  // two chunks per trip: the A sets are back in their roles at the loop edge
  for (int gc = 0; gc < total_gc; gc += 2) {
    chunk(std::integral_constant<int, 0>{}, gc);
    if (d0.vc + 1 == nvc) tile_end();
    advance(d0);
    if (gc + 1 < total_gc) {
      chunk(std::integral_constant<int, 1>{}, gc + 1);
      if (d0.vc + 1 == nvc) tile_end();
      advance(d0);
    }
  }
}

template <int A_T, int C_T>
static int launch_fl(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = FlGeom<A_T, C_T>;
  const int mblocks = a.NB * ((a.H * (a.W + 2) + G::BM - 1) / G::BM);
  if (q) { q[0] = 4 * mblocks; q[1] = 9270000 + A_T * 1000 + G::BN; q[2] = 1630; return ARCO_OK; }      // 4 stat slabs per tile
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = a.Npad / G::BN;
  const int total = mblocks * b.n_nblocks, cus = conv_sp_cus();
  auto kern = conv3d_fl_kernel<A_T, C_T>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), G::LDS_BYTES, st, b);
  return arco_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv3d_fc_kernel<A_T, C_T>: the same tiles, virtual chunks and arithmetic with ONE rendezvous per 16-channel chunk instead of
// one per tap-pair step.  conv3d_fl_kernel's steps carry 6 A_T C_T MFMAs per wave (12 .. 96) between two workgroup barriers, and a
// rendezvous costs ~300 cycles whatever the step holds (measured: 0.22 us per step at 12 MFMAs, 0.30 at 24, ~0.5 at 48).  Here a
// chunk's weights - all five tap-pair steps, 10 x BN x 96 B - arrive as one LDS-DMA burst into one of TWO chunk buffers while the
// consumers stream the previous chunk's five steps back to back (conv3x3_rw_kernel's consumer: one rolling A fragment set, two
// sets of a step's B groups), so the producers and consumers meet once per 30 A_T C_T MFMAs.  The A buffers are sized by the
// launch's plane width (BM + 2 (W + 2) + 2 rows), which is what lets the 61 KB chunk buffers of 64-channel blocks fit twice.
// Outputs bit-identical to conv3d_fl_kernel and igemm_kernel (the same products in the same order).
// LDS: [2][arows][24] A planes + [2][10][BN][24] weights + bias.
// ---------------------------------------------------------------------------------------------------------------------------
// NCW = MFMA waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD, 12 waves with the loaders: one wave's epilogue, rendezvous waits and LDS
// latencies run under the other's MFMAs; 170 registers per wave)
template <int A_T, int C_T, int NCW = 4>
struct FcGeom {
  static constexpr int BM = NCW * 16 * A_T, WPMAX = 63, AROWS = BM + 2 * WPMAX + 2, BN = C_T * 16;
  static constexpr int WCH_DW = 10 * BN * 24;                   // one chunk of weights: 5 steps x [tapL][n][24]
  static constexpr int NW = (10 * BN * 6 + 255) / 256;          // LDS-DMA instructions per thread and chunk
  static constexpr int WBUF_DW = NW * 256 * 4;
  static constexpr int NA_IT = (AROWS * 4 + 255) / 256;
  static constexpr int BIAS_DW = 256;
  static int a_dw(int Wp) { return (BM + 2 * Wp + 2) * 24; }
  // pro_dw: the statistics / affine parameters of a consumer-side activation ([groups][K] mean and istd, [K] gamma and beta)
  static size_t lds_bytes(int Wp, int pro_dw = 0) { return (size_t)(2 * a_dw(Wp) + 2 * WBUF_DW + BIAS_DW + pro_dw) * 4; }
};

// PRO (consumer-side activation, igemm_args.h): the input is the PRE-activation z of the producing convolution; the loader waves apply
// ReLU(BatchNorm(z)) - bn_act_fwd_kernel's arithmetic, operation for operation - to every staged piece before the bf16 split, with the
// statistics and affine parameters read from LDS (copied there once per launch).  Positions that are padding stay zero (the mask is applied
// behind the activation).  The gradient-free passes of the V-Net run their stage -> stage links this way (ops.conv_block3d_nograd).
template <int A_T, int C_T, bool PRO = false, int NCW = 4>
__global__ __launch_bounds__((NCW + 4) * 64) void conv3d_fc_kernel(IgemmArgs a) {
  using G = FcGeom<A_T, C_T, NCW>;
  constexpr int BM = G::BM, BN = G::BN, NA = G::NA_IT, NW = G::NW;
  constexpr int NL = NA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Wp = a.W + 2, npos = a.H * Wp, per_plane = (npos + BM - 1) / BM;
  const int arows = BM + 2 * Wp + 2, A_DW = arows * 24;
  unsigned* const As = reinterpret_cast<unsigned*>(smem);
  unsigned* const Ws = As + 2 * A_DW;
  float* const bias_s = reinterpret_cast<float*>(Ws + 2 * G::WBUF_DW);
  float* const pro_s = bias_s + G::BIAS_DW;                      // PRO: [groups][K] mean, [groups][K] istd, [K] gamma, [K] beta
  const bool producer = threadIdx.x >= NCW * 64;              // waves 0 .. NCW-1: MFMA; the last four: loaders
  const int tid = producer ? (int)threadIdx.x - NCW * 64 : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
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
  const int pgroups = PRO ? (a.pro.groups > 1 ? a.pro.groups : 1) : 1;
  const int ppg = PRO ? a.NB / pgroups : 1;                       // planes per BatchNorm group of the producing layer
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
    // ================================================================ producer waves
    if constexpr (PRO) {               // the producing layer's statistics and parameters -> LDS, before any counted load is issued
      const int gk = pgroups * a.K;
      for (int i = tid; i < 2 * gk + 2 * a.K; i += 256)
        pro_s[i] = i < gk ? a.pro.mean[i] : (i < 2 * gk ? a.pro.istd[i - gk] : (i < 2 * gk + a.K ? a.pro.gamma[i - 2 * gk] : a.pro.beta[i - 2 * gk - a.K]));
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();    // (PRO only, both roles: every loader thread reads quads other threads have written)
    }
    Desc d1 = d0; advance(d1);
    Desc d2 = d1; advance(d2);
    Desc d3 = d2; advance(d3);
    const int qA = tid & 3;
    const float inv_wp = 1.0f / (float)Wp;
    const long plane_dw = (long)a.H * a.W * a.lda;
    int ldsA[NA]; unsigned voff[NA]; unsigned okm_t = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int row = (tid + it * 256) >> 2;
      ldsA[it] = row < arows ? row * 24 + qA * 2 : -1;
      voff[it] = (unsigned)(qA * 16);
    }
    auto tile_geom = [&](const Desc& d) {
      okm_t = 0;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const int row = (tid + it * 256) >> 2;
        const int pidx = d.f0 + row - 1;
        const int py = (int)(((float)pidx + 0.5f) * inv_wp), px = pidx - py * Wp;
        const bool ok = row < arows && pidx >= 0 && py >= 1 && py <= a.H && px >= 1 && px <= a.W;
        voff[it] = ok ? (unsigned)((((py - 1) * a.W + px - 1) * (int)a.lda + qA * 4) * 4) : (unsigned)(qA * 16);
        okm_t |= ok ? (1u << it) : 0u;
      }
    };
    int woff[NW], wq[NW]; bool wz[NW];               // weight piece of LDS-DMA instruction i: dword offset within a depth slice, piece, zero tap / padding
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int p = (i * 4 + wid) * 64 + lane;
      const int q6 = p % 6, n = (p / 6) % BN, t10 = p / (6 * BN);      // t10 = 2 s + tapL
      wz[i] = t10 >= 9;
      woff[i] = t10 < 9 ? ((t10 * a.Npad + n) * a.Kg) * 24 + q6 * 4 : 0;
      wq[i] = q6 * 4;
    }
    const long wslice = (long)9 * a.Npad * a.Kg * 24;
    const float* const zrow = reinterpret_cast<const float*>(conv3d_fl_zero_row);

    f32x4 ra[2][NA]; unsigned okm2[2] = {0, 0};
    int pch2[2] = {0, 0}, pgr2[2] = {0, 0};         // PRO: first channel of the thread's quad and BatchNorm group of the chunk in each register set
    auto load_A = [&](const Desc& d, bool real, int set) {
      if (d.vc == 0) tile_geom(d);
      const int pz = d.pl + d.dz - 1;
      const bool pok = real && pz >= 0 && pz < a.D3;
      const float* gbase = uniform_ptr(a.A + (long)(pok ? d.img + d.dz - 1 : 0) * plane_dw + d.kc * 16);
      okm2[set] = pok ? okm_t : 0u;
      if constexpr (PRO) { pch2[set] = d.kc * 16 + qA * 4; pgr2[set] = real ? d.img / ppg : 0; }
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(voff[it]), "s"(gbase) : "memory");
    };
    auto store_all = [&](unsigned* buf, int set) {
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[set][it])::"memory");
      ProQuad pq;
      if constexpr (PRO) {
        const int gk = pgroups * a.K;
        pq.mu = *reinterpret_cast<const f32x4*>(pro_s + pgr2[set] * a.K + pch2[set]);
        pq.is = *reinterpret_cast<const f32x4*>(pro_s + gk + pgr2[set] * a.K + pch2[set]);
        pq.ga = *reinterpret_cast<const f32x4*>(pro_s + 2 * gk + pch2[set]);
        pq.be = *reinterpret_cast<const f32x4*>(pro_s + 2 * gk + a.K + pch2[set]);
      }
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        f32x4 v = ra[set][it];
        if constexpr (PRO) v = pro_mask(pro_bn_lrelu(v, pq, a.pro.slope), okm2[set] >> it);
        else v = ((okm2[set] >> it) & 1u) ? v : f32x4{0, 0, 0, 0};
        u32x2 q0, q1, q2;
        split3_bf16x4(v, q0, q1, q2);
        if (ldsA[it] >= 0) {
          unsigned* d = buf + ldsA[it];
          *reinterpret_cast<u32x2_ma*>(d) = q0; *reinterpret_cast<u32x2_ma*>(d + 8) = q1; *reinterpret_cast<u32x2_ma*>(d + 16) = q2;
        }
      }
    };
    auto load_W = [&](const Desc& d, int buf) {      // LDS-DMA: the chunk's weights (9 taps of its depth slice + the zero tap) -> chunk buffer `buf`
      const float* base = a.Wp + d.dz * wslice + ((long)d.nblk * BN * a.Kg + d.kc) * 24;
      unsigned* const dst0 = Ws + buf * G::WBUF_DW;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const float* src = wz[i] ? zrow + wq[i] : base + woff[i];
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst0 + (i * 4 + wid) * 256), 16, 0, 0);
      }
    };
    // Prologue.  Per-wave VMEM order: W0 A0 A1 | W1 A2 | (barrier B0) | A3 | ...   Barrier k (the consumers pass it at the head of step 4
    // of chunk k): chunk k + 1 is complete in LDS (activations and weights), the buffers of chunk k are free.
    for (int i = tid; i < G::BIAS_DW; i += 256) bias_s[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    load_W(d0, 0);
    load_A(d0, true, 0);
    load_A(d1, total_gc > 1, 1);
    wait_vm<NL>();                     // chunk 0: weights and activations (only chunk 1's loads are younger)
    store_all(As, 0);
    load_W(d1, 1);                     // (dummy addresses past the last chunk: valid weights of some chunk, never read)
    load_A(d2, total_gc > 2, 0);
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();      // B0: chunk 0 staged, bias
    wait_vm<NL>();                     // chunk 1: activations and weights (only chunk 2's loads are younger)
    store_all(As + A_DW, 1);
    load_A(d3, total_gc > 3, 1);
    Desc dn = d3; advance(dn);         // chunk k + 4
#ifdef ARCO_FC_CLOCK
    unsigned long long pc_bar = 0, pc_dma = 0, pc_store = 0, pc_load = 0, pc_wait = 0;
#endif
    Desc dw_ = d2;                     // chunk k + 2 (its weights)
    for (int k = 0; k < total_gc; k += 2) {          // two chunks per trip: the register sets keep their roles at the loop edge
      wait_lgkm0();
#ifdef ARCO_FC_CLOCK
      const unsigned long long pb0 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();    // barrier k
#ifdef ARCO_FC_CLOCK
      const unsigned long long pb1 = __builtin_amdgcn_s_memtime();
#endif
      load_W(dw_, 0); advance(dw_);    // chunk k + 2's weights into the buffer chunk k has finished with
      wait_vm<NL + NW>();              // chunk k + 2's activations (loaded two chunks ago; after the first trip they have landed long before)
#ifdef ARCO_FC_CLOCK
      const unsigned long long pb2 = __builtin_amdgcn_s_memtime();
#endif
      store_all(As, 0);
#ifdef ARCO_FC_CLOCK
      const unsigned long long pb3 = __builtin_amdgcn_s_memtime();
#endif
      load_A(dn, k + 4 < total_gc, 0);
      advance(dn);
#ifdef ARCO_FC_CLOCK
      const unsigned long long pb4 = __builtin_amdgcn_s_memtime();
#endif
      wait_vm<NL>();                   // the weight burst has landed (and chunk k + 3's activations); only chunk k + 4's loads are younger
#ifdef ARCO_FC_CLOCK
      { const unsigned long long pb5 = __builtin_amdgcn_s_memtime();
        pc_bar += pb1 - pb0; pc_dma += pb2 - pb1; pc_store += pb3 - pb2; pc_load += pb4 - pb3; pc_wait += pb5 - pb4; }
#endif
      if (k + 1 < total_gc) {
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();  // barrier k + 1
        load_W(dw_, 1); advance(dw_);
        wait_vm<NL + NW>();
        store_all(As + A_DW, 1);
        load_A(dn, k + 5 < total_gc, 1);
        advance(dn);
        wait_vm<NL>();
      }
    }
    wait_vm<0>();
#ifdef ARCO_FC_CLOCK
    if (!PRO && threadIdx.x == NCW * 64 && arco_fc_clock_buf && blockIdx.x < 512) {      // (even chunks only: half of the intervals)
      unsigned long long* o = arco_fc_clock_buf + 8 * 512 + 8 * blockIdx.x;
      o[0] = pc_bar; o[1] = pc_dma; o[2] = pc_store; o[3] = pc_load; o[4] = pc_wait; o[5] = (unsigned long long)((total_gc + 1) / 2);
    }
#endif
    return;
  }

  // ================================================================== consumer waves
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s_ = 0; s_ < 5; ++s_) {
    const int tap = 2 * s_ + tl > 8 ? 8 : 2 * s_ + tl;
    aoff[s_] = ((tap / 3) * Wp + tap % 3) * 24;
  }
  const int laneA = (wid * A_T * 16 + li) * 24 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 24 + (g & 1) * 4;             // within a step's [tapL][n] block
  bf16x8 fa[A_T][3], fb[2][C_T][3];
  f32x4 acc[A_T][C_T];
#pragma unroll
  for (int i = 0; i < A_T; ++i)
#pragma unroll
    for (int j = 0; j < C_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  if constexpr (PRO) __builtin_amdgcn_s_barrier();      // the loaders' parameter copy
  __builtin_amdgcn_s_barrier();        // B0
#pragma unroll
  for (int at = 0; at < A_T; ++at)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[at][pl] = lds_bf16x8(As + laneA + aoff[0] + at * 16 * 24 + 8 * pl);
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fb[0][ct][pl] = lds_bf16x8(Ws + laneB + ct * 16 * 24 + 8 * pl);

#ifdef ARCO_FC_CLOCK
  unsigned long long clk_bar = 0, clk_epi = 0, clk_e1 = 0, clk_e2 = 0;
#endif
  auto chunk = [&](auto CP_, int gc, bool more) {
    constexpr int CP = decltype(CP_)::value;               // B set of step 0 (5 steps per chunk: alternates per chunk)
    const unsigned* Acur = As + (gc & 1) * A_DW;
    const unsigned* Anxt = As + ((gc + 1) & 1) * A_DW;
    const unsigned* Wc = Ws + (gc & 1) * G::WBUF_DW;
    const unsigned* Wn = Ws + ((gc + 1) & 1) * G::WBUF_DW;
    auto step = [&](auto S_) {
      constexpr int S = decltype(S_)::value;
      constexpr int P = (CP + S) & 1, Q = P ^ 1;
      if (S == 4) {                    // the one rendezvous of the chunk: the next chunk is complete, this one's buffers are free
        wait_lgkm0();
#ifdef ARCO_FC_CLOCK
        const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef ARCO_FC_CLOCK
        clk_bar += __builtin_amdgcn_s_memtime() - tb0;
#endif
      }
      constexpr int BPA = (C_T * 3 + A_T - 1) / A_T;        // next-step B reads per pixel tile
      const unsigned* An = S < 4 ? Acur : Anxt;
      const unsigned* Wb = (S < 4 ? Wc : Wn) + ((S + 1) % 5) * (2 * BN * 24);
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) {                  // D = W . X^T; small terms first
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][2]);
          mfma_acc(acc[at][ct], fb[P][ct][2], fa[at][0]);
          mfma_acc(acc[at][ct], fb[P][ct][1], fa[at][1]);
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][1]);
          mfma_acc(acc[at][ct], fb[P][ct][1], fa[at][0]);
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][0]);
        }
        if (S < 4 || more) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[at][pl] = lds_bf16x8(An + laneA + aoff[(S + 1) % 5] + at * 16 * 24 + 8 * pl);
        }
#pragma unroll
        for (int k = at * BPA; k < (at + 1) * BPA && k < C_T * 3; ++k)
          fb[Q][k / 3][k % 3] = lds_bf16x8(Wb + laneB + (k / 3) * 16 * 24 + 8 * (k % 3));
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
  auto tile_end = [&]() {
#ifdef ARCO_FC_CLOCK
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int n0 = d0.nblk * BN;
    long pix[A_T];
#pragma unroll
    for (int at = 0; at < A_T; ++at) {
      const int f = d0.f0 + (wid * A_T + at) * 16 + li;
      const int y = (int)(((float)f + 0.5f) * inv_wp), xq = f - y * Wp;
      pix[at] = (y < a.H && xq >= 1 && xq <= a.W) ? ((long)d0.img * a.H + y) * a.W + xq - 1 : -1;
    }
#ifdef ARCO_FC_CLOCK
    const unsigned long long te1 = __builtin_amdgcn_s_memtime();
    clk_e1 += te1 - te0;
#endif
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
          f32x4 v = acc[at][ct] + bv;
          if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + pix[at] * a.ldr + n);
          *reinterpret_cast<f32x4*>(a.C + pix[at] * a.ldc + n) = v;
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
        }
        acc[at][ct] = f32x4{0, 0, 0, 0};
      }
    }
#ifdef ARCO_FC_CLOCK
    const unsigned long long te2 = __builtin_amdgcn_s_memtime();
    clk_e2 += te2 - te1;
#endif
    if (has_stats) {
      const long slab = (long)d0.mblk * NCW + wid, nslab = (long)a.n_mblocks * NCW;
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
#ifdef ARCO_FC_CLOCK
    clk_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
  };
#ifdef ARCO_FC_CLOCK          // DIAGNOSTIC build only (tools/micro/fc_clock.py): the in-kernel clock under the kernel's own load
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
#ifdef ARCO_FC_CLOCK
  if (!PRO && threadIdx.x == 0 && arco_fc_clock_buf && blockIdx.x < 512) {
    unsigned long long* o = arco_fc_clock_buf + 8 * blockIdx.x;
    o[0] = __builtin_amdgcn_s_memtime() - clk_t0; o[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    o[2] = (unsigned long long)total_gc * 30 * A_T * C_T; o[3] = (unsigned long long)my_tiles;
    o[4] = clk_bar; o[5] = clk_epi; o[6] = (unsigned long long)total_gc; o[7] = (clk_e1 << 32) | (clk_e2 & 0xffffffffull);
  }
#endif
}

#ifdef ARCO_FC_CLOCK
extern "C" void arco_fc_clock_buffer(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(arco_fc_clock_buf), &p, sizeof(p)); }
#endif

static int fc_pro_dw(const IgemmArgs& a) { return a.pro.mean ? (2 * (a.pro.groups > 1 ? a.pro.groups : 1) + 2) * a.K : 0; }
template <int A_T, int C_T, int NCW = 4>
static int launch_fc(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = FcGeom<A_T, C_T, NCW>;
  const int mblocks = a.NB * ((a.H * (a.W + 2) + G::BM - 1) / G::BM);
  if (q) { q[0] = NCW * mblocks; q[1] = 9290000 + (A_T + (NCW == 8 ? 5 : 0)) * 1000 + G::BN; q[2] = 1630; return ARCO_OK; }      // one stat slab per MFMA wave and tile
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  const size_t lds = G::lds_bytes(a.W + 2, fc_pro_dw(a));
  if (lds > 160 * 1024) return ARCO_ERR_UNSUPPORTED;
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = a.Npad / G::BN;
  const int total = mblocks * b.n_nblocks, cus = conv_sp_cus();
  if (a.pro.mean) {           // consumer-side activation of the input
    if (NCW != 4) return ARCO_ERR_UNSUPPORTED;
    if (a.pro.drop_mode != 0 || a.NB % (a.pro.groups > 1 ? a.pro.groups : 1) != 0) return ARCO_ERR_UNSUPPORTED;
    auto kern = conv3d_fc_kernel<A_T, C_T, true, 4>;
    static unsigned long long attr_set = 0;
    if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), lds, st, b);
    return arco_launch_status();
  }
  auto kern = conv3d_fc_kernel<A_T, C_T, false, NCW>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3((NCW + 4) * 64), lds, st, b);
  return arco_launch_status();
}
template <int A_T, int C_T, int NCW = 4>
static bool fc_fits(const IgemmArgs& a) { return FcGeom<A_T, C_T, NCW>::lds_bytes(a.W + 2, fc_pro_dw(a)) <= 160 * 1024; }

// ---------------------------------------------------------------------------------------------------------------------------
// Depth-walking form for the 32-channel-wide output blocks (conv3d_dw_kernel<A_T>, 16 C_T = 32 output channels per workgroup).
// conv3d_fl_kernel treats the three depth taps as three times the input channels: every input plane tile is staged THREE times
// (once per output plane it feeds) and every A fragment read from LDS meets only C_T = 2 weight fragments - at 32 output channels
// the 18 ds_read_b128 per 48 MFMAs of a step keep the LDS pipe busier than the matrix cores (measured 142 TFLOP/s at 56 x 56 x 40,
// the same ceiling as conv3x3_sp_kernel<4,2> in 2-D).  Here a workgroup owns a COLUMN: one flat tile of BM positions x S
// consecutive output planes.  It walks the input planes p0 - 1 .. p0 + S once; input plane q feeds the three output planes
// q + 1 (dz = 0), q (dz = 1), q - 1 (dz = 2), each with its own accumulator set:
//   * an input plane tile is staged once (a third of the producers' loads, splits and LDS writes),
//   * an A fragment meets 3 x C_T = 6 weight fragments: 30 ds_read_b128 per 144 MFMAs of a step,
//   * a step is 144 MFMAs per wave between two rendezvous, and a ring slot holds the step's weights of all three depth taps
//     (3 x 2 taps x 32 channels x 96 B = 18 KB; four slots, refilled two steps ahead by LDS-DMA).
// When input plane q is done, output plane q - 1 is complete: stored (+ bias, BN partial sums: the same one-slab-per-wave layout
// as conv3d_fl_kernel), and the accumulator sets move up one role (64 v_mov per plane, against 1440 MFMAs).  The first and last
// input plane of a column feed one output plane only (their other depth taps are skipped, wave-uniform branches); planes
// outside the volume are skipped the same way.  An output element still receives its products in the order (input plane, 16-channel
// chunk, tap pair) = igemm_kernel's (dd, chunk, step): bit-identical results.
// One column per workgroup (not persistent): the column height S is chosen so that the columns about fill the CUs once.
// LDS (A_T = 4): [2][384][24] A planes 73,728 + [4][slot 20,480] + bias 1,024 = 156,672 bytes.
// ---------------------------------------------------------------------------------------------------------------------------
template <int A_T>
struct DwGeom {
  static constexpr int BM = 64 * A_T, WPMAX = 63, AROWS = BM + 2 * WPMAX + 2, BN = 32, C_T = 2;
  static constexpr int A_DW = AROWS * 24;
  static constexpr int SLOT_PIECES = 3 * 2 * BN * 6;            // 16-byte pieces of a step's weights: [dz][tapL][n][6]
  static constexpr int NBI = (SLOT_PIECES + 255) / 256;         // LDS-DMA instructions per thread and slot
  static constexpr int SLOT_DW = NBI * 256 * 4;
  static constexpr int NSLOT = 4;
  static constexpr int NA_IT = (AROWS * 4 + 255) / 256;
  static constexpr int BIAS_DW = 256;
  static constexpr size_t LDS_BYTES = (size_t)(2 * A_DW + NSLOT * SLOT_DW + BIAS_DW) * 4;
};

template <int A_T>
__global__ __launch_bounds__(512) void conv3d_dw_kernel(IgemmArgs a, int S, int nseg) {
  using G = DwGeom<A_T>;
  constexpr int BM = G::BM, BN = G::BN, C_T = G::C_T, NA = G::NA_IT, NB = G::NBI;
  constexpr int NA4 = (NA + 3) / 4;
  constexpr int NL = NA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const As = reinterpret_cast<unsigned*>(smem);
  unsigned* const Bs = As + 2 * G::A_DW;
  float* const bias_s = reinterpret_cast<float*>(Bs + G::NSLOT * G::SLOT_DW);

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const bool producer = threadIdx.x >= 256;
  const int Wp = a.W + 2, npos = a.H * Wp, per_plane = (npos + BM - 1) / BM;
  const int arows = BM + 2 * Wp + 2;
  const int nk = a.K >> 4;
  // the column of this workgroup: (volume, depth segment, flat tile, 32-channel block)
  int u = blockIdx.x;
  const int nblk = u % a.n_nblocks; u /= a.n_nblocks;
  const int tile = u % per_plane; u /= per_plane;
  const int seg = u % nseg; const int vol = u / nseg;
  const int p0 = seg * S, p1 = min(a.D3, p0 + S), f0 = tile * BM;
  const int nin = p1 - p0 + 2;                                   // input planes p0 - 1 .. p1
  const int total_gc = nin * nk;                                 // chunks: (input plane, 16-channel chunk)
  const long plane_dw = (long)a.H * a.W * a.lda;

  if (producer) {
    // ================================================================ producer waves
    const int qA = tid & 3;
    const float inv_wp = 1.0f / (float)Wp;
    int ldsA[NA]; unsigned voff[NA]; unsigned okm_t = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {              // the tile's geometry: fixed for the column
      const int row = (tid + it * 256) >> 2;
      const int pidx = f0 + row - 1;
      const int py = (int)(((float)pidx + 0.5f) * inv_wp), px = pidx - py * Wp;
      const bool ok = row < arows && pidx >= 0 && py >= 1 && py <= a.H && px >= 1 && px <= a.W;
      ldsA[it] = row < arows ? row * 24 + qA * 2 : -1;
      voff[it] = ok ? (unsigned)((((py - 1) * a.W + px - 1) * (int)a.lda + qA * 4) * 4) : (unsigned)(qA * 16);
      okm_t |= ok ? (1u << it) : 0u;
    }
    int woff[NB], wq[NB], wtl[NB];                 // weight piece of LDS-DMA instruction i: dword offset (without the step), piece, local tap (2 = padding)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = (i * 4 + wid) * 64 + lane;
      const int q6 = p % 6, n = (p / 6) % BN, tl_ = (p / (6 * BN)) & 1, dz = p / (12 * BN);
      wtl[i] = p < G::SLOT_PIECES ? tl_ : 2;
      woff[i] = p < G::SLOT_PIECES ? (((dz * 9 + tl_) * a.Npad + n) * a.Kg) * 24 + q6 * 4 : 0;
      wq[i] = q6 * 4;
    }
    const long wslot2 = (long)2 * a.Npad * a.Kg * 24;
    const float* const zrow = reinterpret_cast<const float*>(conv3d_fl_zero_row);
    const float* const wbase = a.Wp + (long)nblk * BN * a.Kg * 24;

    f32x4 ra[2][NA]; unsigned okm2[2] = {0, 0};
    auto load_A = [&](int t, int set) {            // chunk t = (input plane i, 16-channel chunk kc); t >= total_gc: dummy loads (exact vmcnt counts)
      const int i = t / nk, kc = t - i * nk;
      const int q = p0 - 1 + i;
      const bool pok = t < total_gc && q >= 0 && q < a.D3;
      const float* gbase = uniform_ptr(a.A + (long)(pok ? vol * a.D3 + q : 0) * plane_dw + (pok ? kc : 0) * 16);
      okm2[set] = pok ? okm_t : 0u;
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(voff[it]), "s"(gbase) : "memory");
    };
    auto ra_fence = [&](int set) {
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[set][it])::"memory");
    };
    auto store_A = [&](unsigned* buf, int it, int set) {
      const f32x4 v = ((okm2[set] >> it) & 1u) ? ra[set][it] : f32x4{0, 0, 0, 0};
      u32x2 q0, q1, q2;
      split3_bf16x4(v, q0, q1, q2);
      if (ldsA[it] >= 0) {
        unsigned* d = buf + ldsA[it];
        *reinterpret_cast<u32x2_ma*>(d) = q0; *reinterpret_cast<u32x2_ma*>(d + 8) = q1; *reinterpret_cast<u32x2_ma*>(d + 16) = q2;
      }
    };
    auto refill = [&](int gs) {                    // LDS-DMA: the weights of global step gs (chunk gs / 5, tap pair gs % 5; all three depth taps) -> slot gs & 3
      const int t = gs / 5, s = gs - t * 5, kc = t % nk;
      const float* base = wbase + s * wslot2 + kc * 24;
      unsigned* const dst0 = Bs + (gs & 3) * G::SLOT_DW;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool zero = wtl[i] == 2 || (s == 4 && wtl[i] == 1);
        const float* src = zero ? zrow + wq[i] : base + woff[i];
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst0 + (i * 4 + wid) * 256), 16, 0, 0);
      }
    };

    // prologue.  Per-wave VMEM order: A0 | R0 R1 R2 | A1 A2
    for (int i = tid; i < G::BIAS_DW; i += 256) bias_s[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    load_A(0, 0);
    refill(0); refill(1); refill(2);
    wait_vm<3 * NB>();                // chunk 0's activations
    ra_fence(0);
#pragma unroll
    for (int it = 0; it < NA; ++it) store_A(As, it, 0);
    load_A(1, 1);
    load_A(2, 0);
    wait_vm<NL>();                    // chunk 1's activations, slots 0-2
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();

    auto pchunk = [&](auto SET_, int gc) {
      constexpr int SET = decltype(SET_)::value;          // register set of chunk gc + 1
      unsigned* Anxt = As + ((gc + 1) & 1) * G::A_DW;
      auto step = [&](auto S_) {
        constexpr int St = decltype(S_)::value;
        const int gs = gc * 5 + St;
        // the barrier of step gs certifies the weights of steps gs and gs + 1; younger than R(gs + 1) in the per-wave VMEM order:
        // R(gs + 2) and, at steps 0 and 1, the activation loads issued at the previous chunk's step 4
        constexpr int NS = NB + (St <= 1 ? NL : 0);
        wait_vm<NS>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        refill(gs + 3);                // into the slot step gs - 1 has finished with
        if (St == 0) ra_fence(SET);
        if (St <= 3) {
#pragma unroll
          for (int it = St * NA4; it < (St + 1) * NA4 && it < NA; ++it) store_A(Anxt, it, SET);
        }
        if (St == 4) load_A(gc + 3, SET);
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
    };
    for (int gc = 0; gc < total_gc; gc += 2) {
      pchunk(std::integral_constant<int, 1>{}, gc);
      if (gc + 1 < total_gc) pchunk(std::integral_constant<int, 0>{}, gc + 1);
    }
    wait_vm<0>();
    return;
  }

  // ================================================================== consumer waves
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + tl > 8 ? 8 : 2 * s + tl;
    aoff[s] = ((tap / 3) * Wp + tap % 3) * 24;
  }
  const int laneA = (wid * A_T * 16 + li) * 24 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 24 + (g & 1) * 4;
  constexpr int DZ_DW = 2 * BN * 24;                // one depth tap's [tapL][n][24] block of a slot
  bf16x8 fa[2][A_T][3], fb[2][3];
  f32x4 acc[3][A_T][C_T];              // [role]: 0 = output plane q + 1 (dz 0), 1 = q (dz 1), 2 = q - 1 (dz 2) of the current input plane q
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int i = 0; i < A_T; ++i)
#pragma unroll
      for (int j = 0; j < C_T; ++j) acc[r][i][j] = f32x4{0, 0, 0, 0};
  // output addressing of this lane's positions (in-plane pixel offset, -1: padding column / past the plane)
  const float inv_wp = 1.0f / (float)Wp;
  int pixo[A_T];
#pragma unroll
  for (int at = 0; at < A_T; ++at) {
    const int f = f0 + (wid * A_T + at) * 16 + li;
    const int y = (int)(((float)f + 0.5f) * inv_wp), xq = f - y * Wp;
    pixo[at] = (y < a.H && xq >= 1 && xq <= a.W) ? y * a.W + xq - 1 : -1;
  }
  const bool has_stats = a.stat_sum != nullptr;
  const int n0 = nblk * BN;

  __builtin_amdgcn_s_barrier();          // the producers' prologue barrier: chunk 0's activations, slots 0 - 2
#pragma unroll
  for (int at = 0; at < A_T; ++at)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[0][at][pl] = lds_bf16x8(As + laneA + aoff[0] + at * 16 * 24 + 8 * pl);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) fb[0][pl] = lds_bf16x8(Bs + laneB + 8 * pl);

  auto chunk = [&](auto CP_, int gc, int dzmask) {
    constexpr int CP = decltype(CP_)::value;
    const unsigned* Acur = As + (gc & 1) * G::A_DW;
    const unsigned* Anxt = As + ((gc + 1) & 1) * G::A_DW;
    auto step = [&](auto S_) {
      constexpr int St = decltype(S_)::value;
      constexpr int P = (CP + St) & 1, Q = P ^ 1;
      const int gs = gc * 5 + St;
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      const unsigned* Bc = Bs + (gs & 3) * G::SLOT_DW + laneB;
      const unsigned* Bn = Bs + ((gs + 1) & 3) * G::SLOT_DW + laneB;
      const unsigned* An = (St < 4 ? Acur : Anxt) + laneA + aoff[(St + 1) % 5];
      constexpr int NG = 3 * C_T, NCH = NG * A_T;
      constexpr int APC = (A_T * 3 + (NCH - 2) - 1) / (NCH - 2);
#pragma unroll
      for (int j = 0; j < NG; ++j) {               // weight group j = (dz, ct); six groups: the B parity returns to 0 every step
        const int dz = j / C_T, ct = j % C_T;
        const int BPc = j & 1, BPn = BPc ^ 1;
        const bool last = j + 1 == NG;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fb[BPn][pl] = lds_bf16x8(last ? Bn + 8 * pl : Bc + ((j + 1) / C_T) * DZ_DW + ((j + 1) % C_T) * 16 * 24 + 8 * pl);
        __builtin_amdgcn_sched_barrier(0);
        const bool on = (dzmask >> dz) & 1;
#pragma unroll
        for (int at = 0; at < A_T; ++at) {          // D = W . X^T; small terms first
          if (on) {
            mfma_acc(acc[dz][at][ct], fb[BPc][0], fa[P][at][2]);
            mfma_acc(acc[dz][at][ct], fb[BPc][2], fa[P][at][0]);
            mfma_acc(acc[dz][at][ct], fb[BPc][1], fa[P][at][1]);
            mfma_acc(acc[dz][at][ct], fb[BPc][0], fa[P][at][1]);
            mfma_acc(acc[dz][at][ct], fb[BPc][1], fa[P][at][0]);
            mfma_acc(acc[dz][at][ct], fb[BPc][0], fa[P][at][0]);
          }
          const int chain = j * A_T + at;
#pragma unroll
          for (int k = chain * APC; k < (chain + 1) * APC && k < A_T * 3; ++k)
            fa[Q][k / 3][k % 3] = lds_bf16x8(An + (k / 3) * 16 * 24 + 8 * (k % 3));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
  };

  // The accumulator sets change roles once per input plane.  Written as assignments (acc[2] = acc[1] ...) the compiler renames
  // registers instead and places its v_mov copies wherever the live ranges split - directly behind an MFMA it cannot see inside the
  // asm statements, i.e. inside that MFMA's result latency (measured: the last tap pair's products of a plane were lost).  So the
  // roles move on the matrix cores themselves, D = 0 . 0 + C, in place on three variables the compiler keeps where they are:
  // the hardware interlocks MFMA -> MFMA accumulator dependencies.  3 of 1443 MFMAs per 16 x 16 tile and plane.
  const bf16x8 zfrag = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  auto rotate = [&](f32x4& r0, f32x4& r1, f32x4& r2) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %2, %3, %3, %1\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %3, %3, %0\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %3, %3, 0"
                 : "+v"(r0), "+v"(r1), "+v"(r2) : "v"(zfrag));
  };
  auto plane_end = [&](int i) {        // input plane i of the column is done: output plane p0 + i - 2 is complete; the roles move up
    // the last MFMAs' results (asm: no hazard padding by hipcc); the finished set passes THROUGH the statement, so that every
    // compiler-visible read of it is ordered behind the wait
#pragma unroll
    for (int at = 0; at < A_T; ++at) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[2][at][0]), "+v"(acc[2][at][1]) :: "memory");
    if (i >= 2) {
      const int img = vol * a.D3 + p0 + i - 2;
      const long pbase = (long)img * a.H * a.W;
      float s1[C_T][4], s2[C_T][4];
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct) {
        const int n = n0 + ct * 16 + 4 * g;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; }
#pragma unroll
        for (int at = 0; at < A_T; ++at) {
          if (pixo[at] >= 0) {
            const long pix = pbase + pixo[at];
            f32x4 v = acc[2][at][ct] + bv;
            if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + pix * a.ldr + n);
            *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + n) = v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
          }
        }
      }
      if (has_stats) {
        const long slab = ((long)img * per_plane + tile) * 4 + wid, nslab = (long)a.n_mblocks * 4;
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
    }
#pragma unroll
    for (int at = 0; at < A_T; ++at)
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct) rotate(acc[0][at][ct], acc[1][at][ct], acc[2][at][ct]);
  };

  const int s_eff = p1 - p0;
  int pi = 0, pc = 0;                  // input plane of the column, 16-channel chunk within it
  auto mask_of = [&](int i) {
    const int q = p0 - 1 + i;
    int m = 0;
    if (q >= 0 && q < a.D3) {
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) m |= (i - dz >= 0 && i - dz < s_eff) ? (1 << dz) : 0;
    }
    return m;
  };
  int dzmask = mask_of(0);
  auto chunk_done = [&]() {
    if (++pc == nk) { plane_end(pi); pc = 0; ++pi; dzmask = mask_of(pi); }
  };
  // two chunks per trip: the A sets are back in their roles at the loop edge
  for (int gc = 0; gc < total_gc; gc += 2) {
    chunk(std::integral_constant<int, 0>{}, gc, dzmask);
    chunk_done();
    if (gc + 1 < total_gc) {
      chunk(std::integral_constant<int, 1>{}, gc + 1, dzmask);
      chunk_done();
    }
  }
}

template <int A_T>
static int launch_dw(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = DwGeom<A_T>;
  const int per_plane = (a.H * (a.W + 2) + G::BM - 1) / G::BM;
  const int mblocks = a.NB * per_plane;
  if (q) { q[0] = 4 * mblocks; q[1] = 9280000 + A_T * 1000 + G::BN; q[2] = 1630; return ARCO_OK; }
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = a.Npad / G::BN;
  // column height: the columns should about fill the CUs once (one workgroup per CU); never below 2 planes (each column stages two
  // more input planes than it has output planes)
  const long cols = (long)(a.NB / a.D3) * per_plane * b.n_nblocks;
  static const int s_forced = getenv("ARCO_CONV3D_DW_S") ? atoi(getenv("ARCO_CONV3D_DW_S")) : 0;
  int nseg = (int)(conv_sp_cus() / (cols > 0 ? cols : 1));
  if (nseg < 1) nseg = 1;
  if (nseg > a.D3) nseg = a.D3;
  int S = (a.D3 + nseg - 1) / nseg;
  if (S < 2 && a.D3 >= 2) S = 2;
  if (s_forced > 0) S = s_forced < a.D3 ? s_forced : a.D3;
  nseg = (a.D3 + S - 1) / S;
  auto kern = conv3d_dw_kernel<A_T>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(cols * nseg)), dim3(512), G::LDS_BYTES, st, b, S, nseg);
  return arco_launch_status();
}

// A/B knob: ARCO_CONV3D_FL=0 / arco_conv3d_fl_set(0) keeps every 3x3x3 launch on igemm_kernel; ARCO_CONV3D_FL_CFG=<A_T><C_T> (e.g. 42)
// forces one tile shape
static int& conv3d_fl_flag() { static int on = !(getenv("ARCO_CONV3D_FL") && atoi(getenv("ARCO_CONV3D_FL")) == 0); return on; }
extern "C" int arco_conv3d_fl_set(int on) { const int prev = conv3d_fl_flag(); conv3d_fl_flag() = on ? 1 : 0; return prev; }

// Time of a launch with tiles of 64 A_T positions x 16 C_T channels, in microseconds, from a fit of the per-level measurements
// (tools/micro/fl_bench.py at 2 and 4 volumes, profiles/r06_notes.md section 7): the persistent workgroups run ceil(tiles / CUs)
// tiles one after another; a tile is 3 K / 16 chunks of 30 A_T C_T MFMAs per wave (10.7 ns each in these 20-launch bursts, 9.5 ns with four
// weight fragments per activation fragment; sustained launches run ~15 % faster at the same ranking: tools/micro/fc_clock.py) plus the
// rendezvous of a chunk (five at 0.136 us in the per-step form, one at 0.30 us in the per-chunk form) plus 1.9 us of epilogue and launch
// share.  Reproduces the measured times to 0-8 %, and their order except among shapes within 3 %.
static double fl_cost(const IgemmArgs& a, int a_t, int c_t, bool per_chunk) {
  const long tiles = (long)a.NB * ((a.H * (a.W + 2) + 64 * a_t - 1) / (64 * a_t)) * (a.Npad / (16 * c_t));
  const long rounds = (tiles + conv_sp_cus() - 1) / conv_sp_cus();
  const double mfma = 30.0 * a_t * c_t * (c_t == 4 ? 0.0095 : 0.0107) * (a_t == 1 && c_t == 4 ? 1.15 : 1.0);
  return (double)rounds * (3.0 * (a.K >> 4) * (mfma + (per_chunk ? 0.30 : 0.68)) + 1.9);
}

// returns -1 when the shape is not taken (the caller falls through to igemm_kernel).  The choice depends on the plane count, the
// plane size and the channel counts only - never on D3: the tile-count query (arco_conv_mblocks_mma) describes a launch without it.
int conv3d_fl_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  if (!conv3d_fl_flag() || a.mma != 3) return -1;
  if ((a.K & 15) != 0 || a.K < 16 || (a.N & 31) != 0 || a.N != a.Npad || a.N > 256 || a.W + 2 > 63) return -1;
  if ((a.lda & 3) != 0 || (a.ldc & 3) != 0 || (a.R && (a.ldr & 3) != 0)) return -1;
  if ((long)a.H * a.W * a.lda * 4 >= (1l << 31)) return -1;          // 32-bit piece offsets within a plane
  static const int forced = getenv("ARCO_CONV3D_FL_CFG") ? atoi(getenv("ARCO_CONV3D_FL_CFG")) : 0;
  // the depth-walking form (opt-in: measured level with conv3d_fl_kernel<4,2> per launch, 25.4 against 25.5-25.7 ms in the LA step,
  // profiles/r06_notes.md section 7): ARCO_CONV3D_DW = 0 (default) off, 1 the 32-channel layers, 2 every layer it can take
  // (n-blocks of 32); ARCO_CONV3D_FL_CFG = 92 / 93 / 94 forces it with 128- / 192- / 256-position tiles
  static const int dw = getenv("ARCO_CONV3D_DW") ? atoi(getenv("ARCO_CONV3D_DW")) : 0;
  if (a.pro.mean && forced && forced < 100) return -1;
  if (!a.pro.mean && (forced == 94 || forced == 93 || forced == 92 || (!forced && dw && (a.N == 32 || dw >= 2)))) {
    const int a_t = forced ? forced - 90 : 2;
    return a_t == 4 ? launch_dw<4>(a, st, q) : (a_t == 3 ? launch_dw<3>(a, st, q) : launch_dw<2>(a, st, q));
  }
  // conv3d_fc_kernel (one rendezvous per chunk; ids 1xx): ARCO_CONV3D_FC=0 keeps the per-step form
  static const int fc = getenv("ARCO_CONV3D_FC") ? atoi(getenv("ARCO_CONV3D_FC")) : 1;
  int best = forced;
  if (!best) {
    double bc = 1e300;
    const int cand[8] = {44, 34, 24, 14, 42, 32, 22, 12};
    for (int i = 0; i < 8 && !a.pro.mean; ++i) {          // (a consumer-side activation: the per-chunk form only)
      const int a_t = cand[i] / 10, c_t = cand[i] % 10;
      if ((a.N % (16 * c_t)) != 0) continue;
      const double c = fl_cost(a, a_t, c_t, false);
      if (c < bc * 0.999) { bc = c; best = cand[i]; }
    }
    if (fc || a.pro.mean) {
      const int cand2[7] = {124, 114, 152, 142, 132, 122, 112};
      for (int i = 0; i < 7; ++i) {
        const int a_t = (cand2[i] / 10) % 10, c_t = cand2[i] % 10;
        static const int no5 = getenv("ARCO_CONV3D_FL_NO5") ? atoi(getenv("ARCO_CONV3D_FL_NO5")) : 0;      // A/B: without the 320-position tiles
        if ((a.N % (16 * c_t)) != 0 || (no5 && a_t == 5)) continue;
        if ((size_t)(2 * (64 * a_t + 2 * (a.W + 2) + 2) * 96 + 2 * ((10 * 16 * c_t * 6 + 255) / 256) * 4096 + 1024 + 4 * fc_pro_dw(a)) > 160 * 1024) continue;
        const double c = fl_cost(a, a_t, c_t, true);
        if (c < bc * 0.999) { bc = c; best = cand2[i]; }
      }
    }
  }
  switch (best) {
    case 124: if ((a.N & 63) == 0 && fc_fits<2, 4>(a)) return launch_fc<2, 4>(a, st, q); break;
    case 114: if ((a.N & 63) == 0 && fc_fits<1, 4>(a)) return launch_fc<1, 4>(a, st, q); break;
    case 222: if (!a.pro.mean && fc_fits<2, 2, 8>(a)) return launch_fc<2, 2, 8>(a, st, q); break;      // eight MFMA waves
    case 232: if (!a.pro.mean && fc_fits<3, 2, 8>(a)) return launch_fc<3, 2, 8>(a, st, q); break;
    case 212: if (!a.pro.mean && fc_fits<1, 2, 8>(a)) return launch_fc<1, 2, 8>(a, st, q); break;
    case 214: if (!a.pro.mean && (a.N & 63) == 0 && fc_fits<1, 4, 8>(a)) return launch_fc<1, 4, 8>(a, st, q); break;
    case 152: if (fc_fits<5, 2>(a)) return launch_fc<5, 2>(a, st, q); break;
    case 142: if (fc_fits<4, 2>(a)) return launch_fc<4, 2>(a, st, q); break;
    case 132: if (fc_fits<3, 2>(a)) return launch_fc<3, 2>(a, st, q); break;
    case 122: if (fc_fits<2, 2>(a)) return launch_fc<2, 2>(a, st, q); break;
    case 112: if (fc_fits<1, 2>(a)) return launch_fc<1, 2>(a, st, q); break;
    case 44: if ((a.N & 63) == 0) return launch_fl<4, 4>(a, st, q); break;
    case 34: if ((a.N & 63) == 0) return launch_fl<3, 4>(a, st, q); break;
    case 24: if ((a.N & 63) == 0) return launch_fl<2, 4>(a, st, q); break;
    case 14: if ((a.N & 63) == 0) return launch_fl<1, 4>(a, st, q); break;
    case 42: return launch_fl<4, 2>(a, st, q);
    case 32: return launch_fl<3, 2>(a, st, q);
    case 22: return launch_fl<2, 2>(a, st, q);
    case 12: return launch_fl<1, 2>(a, st, q);
  }
  return -1;
}
