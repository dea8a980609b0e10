// Software-pipelined split-bf16 3x3x3 convolution for the V-Net levels below full resolution (ConvBlock / the stages of the
// up path, vnetWithArgs.py:5-31, 222-262: 32 -> 32 on 56 x 40 planes, 64 -> 64 on 28 x 20, 128 -> 128 on 14 x 10, 256 -> 256 on
// 7 x 5 at the LA patch; forward and, with flipped + transposed packed weights, the data gradient).
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
  if (q) { q[0] = 4 * mblocks; q[1] = 9700000 + A_T * 1000 + G::BN; q[2] = 1630; return ARCO_OK; }      // 4 stat slabs per tile
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

// A/B knob: ARCO_CONV3D_FL=0 / arco_conv3d_fl_set(0) keeps every 3x3x3 launch on igemm_kernel; ARCO_CONV3D_FL_CFG=<A_T><C_T> (e.g. 42)
// forces one tile shape
static int& conv3d_fl_flag() { static int on = !(getenv("ARCO_CONV3D_FL") && atoi(getenv("ARCO_CONV3D_FL")) == 0); return on; }
extern "C" int arco_conv3d_fl_set(int on) { const int prev = conv3d_fl_flag(); conv3d_fl_flag() = on ? 1 : 0; return prev; }

// Cost of a launch with tiles of 64 A_T positions x 16 C_T channels, in units of one consumer wave's MFMA issue slots: the persistent
// workgroups run ceil(tiles / CUs) tiles one after another; a step is A_T C_T chains of six MFMAs + a rendezvous with the producers
static double fl_cost(const IgemmArgs& a, int a_t, int c_t) {
  const long tiles = (long)a.NB * ((a.H * (a.W + 2) + 64 * a_t - 1) / (64 * a_t)) * (a.Npad / (16 * c_t));
  const long rounds = (tiles + conv_sp_cus() - 1) / conv_sp_cus();
  return (double)rounds * (a_t * c_t * 6.0 + 14.0);
}

// returns -1 when the shape is not taken (the caller falls through to igemm_kernel).  The choice depends on the plane count, the
// plane size and the channel counts only - never on D3: the tile-count query (arco_conv_mblocks_mma) describes a launch without it.
int conv3d_fl_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  if (!conv3d_fl_flag() || a.mma != 3) return -1;
  if ((a.K & 15) != 0 || a.K < 16 || (a.N & 31) != 0 || a.N != a.Npad || a.N > 256 || a.W + 2 > 63) return -1;
  if ((a.lda & 3) != 0 || (a.ldc & 3) != 0 || (a.R && (a.ldr & 3) != 0)) return -1;
  if ((long)a.H * a.W * a.lda * 4 >= (1l << 31)) return -1;          // 32-bit piece offsets within a plane
  static const int forced = getenv("ARCO_CONV3D_FL_CFG") ? atoi(getenv("ARCO_CONV3D_FL_CFG")) : 0;
  int best = forced;
  if (!best) {
    double bc = 1e300;
    const int cand[6] = {44, 24, 14, 42, 22, 12};
    for (int i = 0; i < 6; ++i) {
      const int a_t = cand[i] / 10, c_t = cand[i] % 10;
      if ((a.N % (16 * c_t)) != 0) continue;
      const double c = fl_cost(a, a_t, c_t);
      if (c < bc * 0.999) { bc = c; best = cand[i]; }
    }
  }
  switch (best) {
    case 44: if ((a.N & 63) == 0) return launch_fl<4, 4>(a, st, q); break;
    case 24: if ((a.N & 63) == 0) return launch_fl<2, 4>(a, st, q); break;
    case 14: if ((a.N & 63) == 0) return launch_fl<1, 4>(a, st, q); break;
    case 42: return launch_fl<4, 2>(a, st, q);
    case 22: return launch_fl<2, 2>(a, st, q);
    case 12: return launch_fl<1, 2>(a, st, q);
  }
  return -1;
}
