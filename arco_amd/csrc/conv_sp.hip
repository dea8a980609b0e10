// Software-pipelined split-bf16 3x3 convolution for the wide 2-D levels (ConvBlock, unetWithArgs.py:31-47; forward and,
// with flipped+transposed packed weights, the data gradient) - the same arithmetic as igemm_kernel<9,...,MMA=3>
// (igemm.hip: six v_mfma_f32_16x16x32_bf16 per fp32-accurate product, fp32 accumulate), restructured around what
// bounded that kernel (profiles/README.md, round 2 ablation: its staging phases and its matrix phases did not overlap):
//
//   * ONE persistent workgroup per CU (4 waves, one per SIMD, the whole register file), walking its tiles in a
//     flattened (tile, 16-channel chunk, tap-pair step) sequence, so the next tile's loads run under this tile's MFMAs
//     and the output stores of a tile drain under the next one's.
//   * a wave owns A_T x C_T MFMA tiles (64 pixels x 64 channels at 4 x 4): one A or B fragment read from LDS feeds
//     four MFMA groups instead of two - LDS fragment traffic per MFMA drops by a third.
//   * the pre-split weights never touch a register: global_load_lds_dwordx4 (LDS-DMA) fills a RING of five slots, one
//     per tap-pair step; the slot a step has finished with is refilled with the next chunk's weights of that step while
//     the following four steps compute.  Counted s_waitcnt vmcnt(N) + raw s_barrier keep the DMA in flight across the
//     step barriers (a __syncthreads() would drain it).
//   * the activation halo tile of the NEXT chunk is loaded to registers a chunk ahead, split into its three bf16 planes
//     and written to the second A buffer in thirds at the head of steps 1-3, under the MFMAs of the current chunk.
//   * the fragments of step s+1 (A of the next tap pair, the first B group of slot s+1) are read while step s is on the
//     matrix cores: the barrier of step s certifies slots s AND s+1.
//   * D = W . X^T orientation: a lane ends up with 4 consecutive output channels of one pixel -> 16-byte stores.
//
// LDS: [2][AROWS][24] A planes + [5][slot] B ring + N bias floats (A_T = 4, C_T = 4: 62,208 + 61,440 + 1,024 bytes).
// Shapes taken (conv_sp_dispatch): K % 16 == 0, N % 64 == 0, H % 16 == 0, W % 16 == 0, 16-byte aligned rows, and
// enough tiles to give most CUs one; everything else stays on igemm_kernel.
#include "igemm_args.h"
#include <stdlib.h>
#include <type_traits>

#include "sp_util.h"
__device__ __attribute__((aligned(64))) unsigned int conv_sp_zero_row[32];      // zero-initialised: the "tap 9" weights

template <int A_T, int C_T>
struct SpGeom {
  static constexpr int TH = 4 * A_T, HR = TH + 2, AROWS = HR * 18, BN = C_T * 16;
  static constexpr int A_DW = AROWS * 24;                       // dwords per A buffer
  static constexpr int NBI = (2 * BN * 6 + 255) / 256;          // LDS-DMA instructions per thread and slot
  static constexpr int SLOT_DW = NBI * 256 * 4;                 // slot stride in dwords (whole wave-instructions)
  static constexpr int NA_IT = (AROWS * 4 + 255) / 256;         // 16-byte activation loads per thread and chunk
  static constexpr int BIAS_DW = 256;
  static constexpr size_t LDS_BYTES = (size_t)(2 * A_DW + 5 * SLOT_DW + BIAS_DW) * 4;
};

template <int A_T, int C_T, bool PRO = false>
__global__ __launch_bounds__(512) void conv3x3_sp_kernel(IgemmArgs a) {
  using G = SpGeom<A_T, C_T>;
  constexpr int TH = G::TH, BN = G::BN, NA = G::NA_IT, NB = G::NBI;
  constexpr int NA4 = (NA + 3) / 4;                              // activation pieces split + written per staging step
  constexpr int NL = NA + (PRO ? 4 : 0);     // VMEM loads per chunk and wave: the activation pieces (+ the four parameter quads of a consumer-side activation)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const As = reinterpret_cast<unsigned*>(smem);
  unsigned* const Bs = As + 2 * G::A_DW;
  float* const bias_s = reinterpret_cast<float*>(Bs + 5 * G::SLOT_DW);

  // 8 waves: 0-3 consume (fragment reads + MFMAs, one per SIMD), 4-7 produce (LDS-DMA refills, activation loads, split and
  // LDS writes).  A SIMD then always has a wave that can issue while the other waits - on a single wave per SIMD the
  // issue cost of the DMA instructions and the split VALU work sat between the MFMAs.  Both roles pass the same
  // barriers: one before the first step, one per step.
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const bool producer = threadIdx.x >= 256;
  const int tiles_x = a.W >> 4, tiles_y = a.H / TH, tiles_img = tiles_x * tiles_y;
  const int nchunks = a.K >> 4;
  const int total_tiles = a.n_mblocks * a.n_nblocks;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so XCD x gets the x-th
  // CONTIGUOUS eighth of the tiles - neighbouring tiles (shared halo rows, the N-blocks of one pixel tile) meet in one L2
  const bool xcd_map = (gridDim.x & 7) == 0;
  const int G8 = xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int T8 = xcd_map ? (total_tiles + 7) >> 3 : total_tiles;
  const int tile0 = xcd_map ? ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int tile_end_x = xcd_map ? min(total_tiles, (((int)blockIdx.x & 7) + 1) * T8) : total_tiles;
  const int my_tiles = tile0 < tile_end_x ? (tile_end_x - tile0 + G8 - 1) / G8 : 0;
  const int total_gc = my_tiles * nchunks;
  if (my_tiles == 0) return;
  const bool has_stats = a.stat_sum != nullptr;

  // chunk descriptors (tile ordinal j of this workgroup, 16-channel chunk c), advanced incrementally: the divisions of the
  // tile decode run once per tile, not once per load
  struct Desc { int j, c, img, y0, x0, nblk, mblk, grp; };
  const int ipg = PRO ? a.NB / (a.pro.groups > 1 ? a.pro.groups : 1) : 1;      // images per BatchNorm group of the producing layer
  auto decode = [&](Desc& d) {
    const int v = tile0 + d.j * G8;
    d.mblk = v / a.n_nblocks; d.nblk = v - d.mblk * a.n_nblocks;
    d.img = d.mblk / tiles_img;
    d.grp = PRO ? d.img / ipg : 0;
    const int r = d.mblk - d.img * tiles_img, ty = r / tiles_x;
    d.y0 = ty * TH; d.x0 = (r - ty * tiles_x) * 16;
  };
  auto advance = [&](Desc& d) { if (++d.c == nchunks) { d.c = 0; ++d.j; decode(d); } };
  Desc d0{0, 0, 0, 0, 0, 0, 0, 0};
  decode(d0);

  if (producer) {
    // ================================================================ producer waves
    Desc d1 = d0; advance(d1);
    Desc d2 = d1; advance(d2);
    Desc d3 = d2; advance(d3);
    const int qA = tid & 3;
    // per-thread geometry of the NA activation pieces, fixed for the launch (see conv3x3_rw_kernel): LDS offset, byte offset
    // from the halo origin, masks of the pieces on the tile's top / bottom halo row and left / right halo column
    int ldsA[NA]; unsigned voff[NA]; int poff[NA];
    unsigned m_valid = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int row = (tid + it * 256) >> 2;
      const int hy = row / 18, hx = row - hy * 18;
      const bool valid = row < G::AROWS;
      poff[it] = hy * a.W + hx;                    // (PRO: pixel offset from the halo origin, for the dropout element index)
      ldsA[it] = valid ? row * 24 + qA * 2 : -1;
      voff[it] = valid ? (unsigned)(((hy * a.W + hx) * (int)a.lda + qA * 4) * 4) : 0u;
      m_valid |= valid ? (1u << it) : 0u;
      m_top |= (valid && hy == 0) ? (1u << it) : 0u; m_bot |= (valid && hy == TH + 1) ? (1u << it) : 0u;
      m_left |= (valid && hx == 0) ? (1u << it) : 0u; m_right |= (valid && hx == 17) ? (1u << it) : 0u;
    }
    const unsigned safe_off = (unsigned)(((a.W + 1) * (int)a.lda + qA * 4) * 4);      // the tile's first pixel: always inside the tensor
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
    const float* const zrow = reinterpret_cast<const float*>(conv_sp_zero_row);

    // Activation loads of a chunk.  Every lane loads (clamped address, the mask zeroes at the split), and the loads are asm
    // statements: hipcc, which drains the LDS-DMA queue (vmcnt(0)) at the use of any load it knows of, does not see
    // them.  They sit in the per-wave VMEM order right behind a refill, so the step-1 wait (at most 2 NB younger
    // instructions in flight) certifies them; ra_fence() then orders every reader behind that wait.
    // Two register sets, loaded TWO chunks ahead (set = parity of the chunk the data belongs to): with 32-channel blocks
    // a chunk lasts ~2 us, less than an HBM miss under load.
    f32x4 ra[2][NA]; unsigned okm2[2] = {0, 0};
    // consumer-side activation (see conv3x3_rw_kernel): four parameter quads per chunk behind its activation loads, NL loads per chunk
    f32x4 rp[2][4]; unsigned ebase2[2] = {0, 0}; bool bord2[2] = {true, true};
    const float* const pm_ = PRO ? uniform_ptr(a.pro.mean) : nullptr; const float* const pi_ = PRO ? uniform_ptr(a.pro.istd) : nullptr;
    const float* const pg_ = PRO ? uniform_ptr(a.pro.gamma) : nullptr; const float* const pb_ = PRO ? uniform_ptr(a.pro.beta) : nullptr;
    const bool pdrop = PRO && a.pro.drop_mode == 1;
    uint32_t dkey = 0, dthr = 0; float keep_scale = 1.f;
    u32x2 salt = {0u, 0u};        // the per-replay dropout salt of a graph-captured pass: the kernel's OLDEST counted load (an asm load as
    if (PRO && pdrop && a.pro.seed_dev) {     // the activations': a load hipcc knows of would drain the DMA queue in front of its use)
      const float* sp_ = uniform_ptr(reinterpret_cast<const float*>(a.pro.seed_dev));
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(salt) : "v"(0u), "s"(sp_) : "memory");
    }
    auto drop_setup = [&]() {     // behind the first vmcnt wait of the prologue
      if (PRO && pdrop) {
        asm volatile("" : "+v"(salt)::"memory");
        const unsigned long long sv = ((unsigned long long)salt[1] << 32) | salt[0];
        const unsigned long long sd = a.pro.seed_dev ? a.pro.seed ^ (sv * 0x9E3779B97F4A7C15ull) : a.pro.seed;
        dkey = drop_key32(sd); dthr = drop_thr16(a.pro.p); keep_scale = 1.0f / (1.0f - a.pro.p);
      }
    };
    auto load_A = [&](const Desc& d, bool real, int set) {
      // scalar base of the halo origin + the per-piece byte offsets: a tile costs a few scalar instructions and one mask
      // expression, not ~20 VALU instructions per piece; out-of-image pieces load the tile's first pixel and are zeroed at
      // the split; a descriptor past the last chunk (real == false: dummy loads that keep the vmcnt counts exact) may name
      // a tile outside the tensor and loads the tensor's first pixels
      const float* gbase = uniform_ptr(real ? a.A + ((((long)d.img * a.H + d.y0 - 1) * a.W + d.x0 - 1) * a.lda + d.c * 16) : a.A);
      unsigned bad = (d.y0 == 0 ? m_top : 0u) | (d.y0 + TH == a.H ? m_bot : 0u) | (d.x0 == 0 ? m_left : 0u) | (d.x0 + 16 == a.W ? m_right : 0u);
      if (!real) bad = ~0u;
      const unsigned okm = m_valid & ~bad;
      okm2[set] = okm;
      const bool border = !real || d.y0 == 0 || d.y0 + TH == a.H || d.x0 == 0 || d.x0 + 16 == a.W;     // wave-uniform: NA loads per wave either way
      if constexpr (PRO) bord2[set] = border;
      if (!border) {
#pragma unroll
        for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(voff[it]), "s"(gbase) : "memory");
      } else {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
          const unsigned o = ((okm >> it) & 1u) ? voff[it] : safe_off;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(o), "s"(gbase) : "memory");
        }
      }
      if constexpr (PRO) {
        const int ch = d.c * 16 + qA * 4;
        const unsigned po = (unsigned)(((real ? d.grp : 0) * a.K + ch) * 4), pa = (unsigned)(ch * 4);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][0]) : "v"(po), "s"(pm_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][1]) : "v"(po), "s"(pi_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][2]) : "v"(pa), "s"(pg_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][3]) : "v"(pa), "s"(pb_) : "memory");
        ebase2[set] = (unsigned)((((d.img * a.H + d.y0 - 1) * a.W + d.x0 - 1) * a.K) + ch);
      }
    };
    auto ra_fence = [&](int set) {
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[set][it])::"memory");
      if constexpr (PRO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rp[set][j])::"memory");
      }
    };
    auto store_A = [&](unsigned* buf, int it, int set) {    // split piece `it` into its three bf16 planes (zero outside the image)
      f32x4 v = ra[set][it];
      if constexpr (PRO) {
        v = pro_bn_lrelu(v, ProQuad{rp[set][0], rp[set][1], rp[set][2], rp[set][3]}, a.pro.slope);
        if (pdrop) v = pro_dropout(v, dkey, ebase2[set] + (unsigned)poff[it] * (unsigned)a.K, dthr, keep_scale);
        if (bord2[set]) v = pro_mask(v, okm2[set] >> it);       // (interior tiles: every staged piece is inside the image - wave-uniform)
      } else {
        v = ((okm2[set] >> it) & 1u) ? v : f32x4{0, 0, 0, 0};
      }
      u32x2 p0, p1, p2;
      split3_bf16x4(v, p0, p1, p2);
      if (ldsA[it] >= 0) {
        unsigned* d = buf + ldsA[it];
        *reinterpret_cast<u32x2_ma*>(d) = p0; *reinterpret_cast<u32x2_ma*>(d + 8) = p1; *reinterpret_cast<u32x2_ma*>(d + 16) = p2;
      }
    };
    auto refill = [&](int slot, const Desc& d) {   // LDS-DMA: weights of step `slot` of chunk d -> ring slot
      const float* base = a.Wp + slot * wslot2 + ((long)d.nblk * BN * a.Kg + d.c) * 24;
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
    drop_setup();
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
        // s0 R4 | s1 R0' | s2 R1' | s3 R2' | s4 R3' A+3   (the last chunks issue the same instructions on dummy addresses).
        // The activations split at steps 0-3 were loaded two chunks ago: older than anything these waits leave in flight.
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
  // fragment addressing: lanes g = 0,1 take tap 2s, g = 2,3 tap 2s+1 (step 4: tap 8 and the zero tap)
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int tap = 2 * s + tl > 8 ? 8 : 2 * s + tl;
    aoff[s] = ((tap / 3) * 18 + tap % 3) * 24;
  }
  const int laneA = (wid * A_T * 18 + li) * 24 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 24 + (g & 1) * 4;
  // Fragment registers: two A sets used alternately by consecutive steps (a chunk has five steps, so the roles of the sets
  // swap from chunk to chunk: the chunk body exists once per parity - no register copies between steps), and two B sets of
  // ONE 16-channel group each, used alternately by consecutive groups: group k+1 (the next group of this step, or the
  // first group of the next step) is read while group k is on the matrix cores.
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
    for (int pl = 0; pl < 3; ++pl) fa[0][at][pl] = lds_bf16x8(As + laneA + aoff[0] + at * 18 * 24 + 8 * pl);
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
      // LDS reads in the shadow of the MFMA chains: at the head of every group the three fragments of the NEXT group, and
      // behind the first chains the next step's A fragments (set Q), APC per chain.  Nothing crosses the sched_barriers,
      // so the reads stay where they are put (LDS returns in order, hipcc counts the waits).
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
            fa[Q][k / 3][k % 3] = lds_bf16x8(An + laneA + aoff[(S + 1) % 5] + (k / 3) * 18 * 24 + 8 * (k % 3));
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

  auto tile_end = [&]() {              // ---- tile done: bias, store, BN partial statistics; accumulators back to zero
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results (asm: no hazard padding by hipcc)
    const int n0 = d0.nblk * BN;
    float s1[C_T][4], s2[C_T][4];
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct) {
      const int n = n0 + ct * 16 + 4 * g;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; }
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
        const long pix = ((long)d0.img * a.H + d0.y0 + wid * A_T + at) * a.W + d0.x0 + li;
        f32x4 v = acc[at][ct] + bv;
        if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + pix * a.ldr + n);
        *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + n) = v;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
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
    if (d0.c + 1 == nchunks) tile_end();
    advance(d0);
    if (gc + 1 < total_gc) {
      chunk(std::integral_constant<int, 1>{}, gc + 1);
      if (d0.c + 1 == nchunks) tile_end();
      advance(d0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Resident-weights variant for the shallow levels (K <= 32 input channels, N = 16 or 32 outputs: the 256^2 and 128^2 planes
// of the U-Net).  All packed weights of the launch (at most 61 KB: 2 chunks x 10 tap rows x 32 channels x 96 B) are copied to
// LDS once by LDS-DMA, so a step needs no ring slot, no refill and NO barrier: the consumer streams a chunk's 5 steps
// back to back and meets the producers once per chunk (at the head of step 4: "the other activation buffer is complete,
// this one is free from now on"), instead of five times.  With 24-48 MFMAs per wave and step the per-step barrier and DMA
// issue of conv3x3_sp_kernel cost as much as the MFMAs themselves at these widths.
// Pixel-tile-major MFMA order: ONE A fragment set - the chains of a 16-pixel tile run over the step's B groups back to
// back, then the next step's fragments of that tile are read into the registers just released - and two sets of a step's
// B groups, used alternately by consecutive steps.  Producers as in conv3x3_sp_kernel (asm loads two chunks ahead, split
// to bf16 planes), a whole chunk time per tile.  LDS: [2][AROWS][24] A planes + [nchunks][10][BN][24] weights + bias.
// ---------------------------------------------------------------------------------------------------------------------------
// NCW = MFMA waves per workgroup: 4 (one per SIMD, A_T MFMA tiles of 16 pixels each) or 8 (two per SIMD, twelve waves with the loaders: one wave's
// epilogue and LDS latencies under the other's MFMAs; the tile - NCW A_T rows x 16 columns - and the LDS image are the same)
template <int A_T, int C_T, int NCW = 4>
struct RwGeom {
  static constexpr int TH = NCW * A_T, AROWS = (TH + 2) * 18, BN = C_T * 16;
  static constexpr int A_DW = AROWS * 24;
  static constexpr int WCH_DW = 10 * BN * 24;                   // one chunk of weights: 9 taps + the zero tap
  static constexpr int NA_IT = (AROWS * 4 + 255) / 256;
  static constexpr int BIAS_DW = 64;
  static size_t lds_bytes(int nchunks) { return (size_t)(2 * A_DW + nchunks * WCH_DW + BIAS_DW) * 4; }
};

template <int A_T, int C_T, bool PRO = false, int NCW = 4>
__global__ __launch_bounds__((NCW + 4) * 64) void conv3x3_rw_kernel(IgemmArgs a) {
  using G = RwGeom<A_T, C_T, NCW>;
  constexpr int TH = G::TH, BN = G::BN, NA = G::NA_IT;
  constexpr int NL = NA + (PRO ? 4 : 0);     // VMEM loads per chunk and wave: the activation pieces (+ the four parameter quads of a consumer-side activation)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const As = reinterpret_cast<unsigned*>(smem);
  const int nchunks = (a.K + 15) >> 4;                        // (K = 4 / 8 / 12: one chunk, the missing channels read as zero)
  unsigned* const Ws = As + 2 * G::A_DW;
  float* const bias_s = reinterpret_cast<float*>(Ws + nchunks * G::WCH_DW);
  const bool producer = threadIdx.x >= NCW * 64;              // waves 0 .. NCW-1: MFMA; the last four: loaders
  const int tid = producer ? (int)threadIdx.x - NCW * 64 : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int tiles_x = a.W >> 4, tiles_y = a.H / TH, tiles_img = tiles_x * tiles_y;
  const int total_tiles = a.n_mblocks;                        // (one N-block: N == BN)
  const bool xcd_map = (gridDim.x & 7) == 0;
  const int G8 = xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int T8 = xcd_map ? (total_tiles + 7) >> 3 : total_tiles;
  const int tile0 = xcd_map ? ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int tile_end_x = xcd_map ? min(total_tiles, (((int)blockIdx.x & 7) + 1) * T8) : total_tiles;
  const int my_tiles = tile0 < tile_end_x ? (tile_end_x - tile0 + G8 - 1) / G8 : 0;
  const int total_gc = my_tiles * nchunks;
  if (my_tiles == 0) return;
  const bool has_stats = a.stat_sum != nullptr;
  struct Desc { int j, c, img, y0, x0, mblk, grp; };
  const int ipg = PRO ? a.NB / (a.pro.groups > 1 ? a.pro.groups : 1) : 1;      // images per BatchNorm group of the producing layer
  auto decode = [&](Desc& d) {
    d.mblk = tile0 + d.j * G8;
    d.img = d.mblk / tiles_img;
    d.grp = PRO ? d.img / ipg : 0;
    const int r = d.mblk - d.img * tiles_img, ty = r / tiles_x;
    d.y0 = ty * TH; d.x0 = (r - ty * tiles_x) * 16;
  };
  auto advance = [&](Desc& d) { if (++d.c == nchunks) { d.c = 0; ++d.j; decode(d); } };
  Desc d0{0, 0, 0, 0, 0, 0, 0};
  decode(d0);

  if (producer) {
    // ---- weights: [chunk][step][tapL][n][24 dwords]; the zero tap (step 4, tapL 1) comes from the zero row
    const float* const zrow = reinterpret_cast<const float*>(conv_sp_zero_row);
    const int wpieces = nchunks * 10 * BN * 6;
    for (int p0 = wid * 64; p0 < wpieces; p0 += 256) {        // one wave-instruction = 64 pieces = 1 KB, contiguous in LDS
      const int p = p0 + lane;
      const int q6 = p % 6, row = p / 6, n = row % BN, tp = (row / BN) % 10, c = row / (BN * 10);
      const float* src = (tp < 9 && p < wpieces) ? a.Wp + (((long)tp * a.Npad + n) * a.Kg + c) * 24 + q6 * 4 : zrow + q6 * 4;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(Ws + p0 * 4), 16, 0, 0);
    }
    for (int i = tid; i < G::BIAS_DW; i += 256) bias_s[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    Desc d1 = d0; advance(d1);
    Desc d2 = d1; advance(d2);
    Desc d3 = d2; advance(d3);
    const int qA = tid & 3;
    // Per-thread geometry of the NA activation pieces, fixed for the whole launch: LDS offset, byte offset from the halo
    // origin (pixel (y0 - 1, x0 - 1), channel 16 c + 4 qA), and bit masks over the pieces that fall on the tile's top /
    // bottom halo row and left / right halo column.  A tile then costs a handful of scalar instructions (base address, which
    // image borders it touches) and ONE mask expression - no per-piece coordinate arithmetic (round 2: ~20 VALU instructions
    // per piece, 2300 cycles per tile next to the MFMA waves of the same SIMDs).
    int ldsA[NA]; unsigned voff[NA]; int poff[NA];
    unsigned m_valid = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int row = (tid + it * 256) >> 2;
      const int hy = row / 18, hx = row - hy * 18;
      const bool valid = row < G::AROWS;
      poff[it] = hy * a.W + hx;                    // (PRO: pixel offset from the halo origin, for the dropout element index)
      ldsA[it] = valid ? row * 24 + qA * 2 : -1;
      voff[it] = valid ? (unsigned)(((hy * a.W + hx) * (int)a.lda + qA * 4) * 4) : 0u;
      m_valid |= valid ? (1u << it) : 0u;
      m_top |= (valid && hy == 0) ? (1u << it) : 0u; m_bot |= (valid && hy == TH + 1) ? (1u << it) : 0u;
      m_left |= (valid && hx == 0) ? (1u << it) : 0u; m_right |= (valid && hx == 17) ? (1u << it) : 0u;
    }
    const unsigned safe_off = (unsigned)(((a.W + 1) * (int)a.lda + qA * 4) * 4);      // the tile's first pixel: always inside the tensor
    f32x4 ra[2][NA]; unsigned okm2[2] = {0, 0};
    // consumer-side activation: the four parameter quads of the chunk's channels travel with its activation loads (same asm
    // loads, same vmcnt accounting: NL instructions per chunk), the element index base of the dropout mask with the register set
    f32x4 rp[2][4]; unsigned ebase2[2] = {0, 0}; bool bord2[2] = {true, true};
    const float* const pm_ = PRO ? uniform_ptr(a.pro.mean) : nullptr; const float* const pi_ = PRO ? uniform_ptr(a.pro.istd) : nullptr;
    const float* const pg_ = PRO ? uniform_ptr(a.pro.gamma) : nullptr; const float* const pb_ = PRO ? uniform_ptr(a.pro.beta) : nullptr;
    const bool pdrop = PRO && a.pro.drop_mode == 1;
    uint32_t dkey = 0, dthr = 0; float keep_scale = 1.f;
    u32x2 salt = {0u, 0u};        // the per-replay dropout salt of a graph-captured pass: the kernel's OLDEST counted load (an asm load as
    if (PRO && pdrop && a.pro.seed_dev) {     // the activations': a load hipcc knows of would drain the DMA queue in front of its use)
      const float* sp_ = uniform_ptr(reinterpret_cast<const float*>(a.pro.seed_dev));
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(salt) : "v"(0u), "s"(sp_) : "memory");
    }
    auto drop_setup = [&]() {     // behind the first vmcnt wait of the prologue
      if (PRO && pdrop) {
        asm volatile("" : "+v"(salt)::"memory");
        const unsigned long long sv = ((unsigned long long)salt[1] << 32) | salt[0];
        const unsigned long long sd = a.pro.seed_dev ? a.pro.seed ^ (sv * 0x9E3779B97F4A7C15ull) : a.pro.seed;
        dkey = drop_key32(sd); dthr = drop_thr16(a.pro.p); keep_scale = 1.0f / (1.0f - a.pro.p);
      }
    };
    auto load_A = [&](const Desc& d, bool real, int set) {
      // (scalar) base of the halo origin; out-of-image pieces load the tile's first pixel instead and are zeroed at the split
      // (a descriptor past the workgroup's last chunk - real == false, the loads are dummies that keep the vmcnt counts
      //  exact - may name a tile outside the tensor: those load the tensor's first pixels)
      const float* gbase = uniform_ptr(real ? a.A + ((((long)d.img * a.H + d.y0 - 1) * a.W + d.x0 - 1) * a.lda + d.c * 16) : a.A);
      unsigned bad = (d.y0 == 0 ? m_top : 0u) | (d.y0 + TH == a.H ? m_bot : 0u) | (d.x0 == 0 ? m_left : 0u) | (d.x0 + 16 == a.W ? m_right : 0u);
      if (!real || d.c * 16 + qA * 4 >= a.K) bad = ~0u;
      okm2[set] = m_valid & ~bad;
      // WAVE-UNIFORM choice (every wave must issue exactly NA load instructions: the vmcnt waits count them)
      const bool border = !real || d.y0 == 0 || d.y0 + TH == a.H || d.x0 == 0 || d.x0 + 16 == a.W || d.c * 16 + 16 > a.K;
      if constexpr (PRO) bord2[set] = border;
#ifdef RW_NO_LOAD          // (timing-only ablation builds, tools/debug/rw_abl.sh: never in the shipped library)
      if (true) { (void)gbase; } else
#endif
      if (!border) {
#pragma unroll
        for (int it = 0; it < NA; ++it) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(voff[it]), "s"(gbase) : "memory");
      } else {
        const unsigned okm = okm2[set];
#pragma unroll
        for (int it = 0; it < NA; ++it) {
          const unsigned o = ((okm >> it) & 1u) ? voff[it] : safe_off;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[set][it]) : "v"(o), "s"(gbase) : "memory");
        }
      }
      if constexpr (PRO) {
        const int ch = d.c * 16 + qA * 4;
        const unsigned po = (unsigned)(((real ? d.grp : 0) * a.K + (ch < a.K ? ch : 0)) * 4), pa = (unsigned)((ch < a.K ? ch : 0) * 4);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][0]) : "v"(po), "s"(pm_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][1]) : "v"(po), "s"(pi_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][2]) : "v"(pa), "s"(pg_) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rp[set][3]) : "v"(pa), "s"(pb_) : "memory");
        // element index of channel `ch` of the halo origin (pixel (y0 - 1, x0 - 1)); pieces outside the image are masked anyway
        ebase2[set] = (unsigned)((((d.img * a.H + d.y0 - 1) * a.W + d.x0 - 1) * a.K) + ch);
      }
    };
    auto store_all = [&](unsigned* buf, int set) {
#pragma unroll
      for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(ra[set][it])::"memory");
      ProQuad pq;
      if constexpr (PRO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rp[set][j])::"memory");
        pq = ProQuad{rp[set][0], rp[set][1], rp[set][2], rp[set][3]};
      }
      const unsigned okm = okm2[set];
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        f32x4 v = ra[set][it];
        if constexpr (PRO) {
          v = pro_bn_lrelu(v, pq, a.pro.slope);
          if (pdrop) v = pro_dropout(v, dkey, ebase2[set] + (unsigned)poff[it] * (unsigned)a.K, dthr, keep_scale);
          if (bord2[set]) v = pro_mask(v, okm >> it);        // (interior tiles: every staged piece is inside the image - wave-uniform)
        } else {
          v = ((okm >> it) & 1u) ? v : f32x4{0, 0, 0, 0};
        }
        u32x2 p0, p1, p2;
#ifdef RW_NO_SPLIT
        p0 = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}; p1 = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])}; p2 = p0;
#else
        split3_bf16x4(v, p0, p1, p2);
#endif
        if (ldsA[it] >= 0) {
          unsigned* d = buf + ldsA[it];
          *reinterpret_cast<u32x2_ma*>(d) = p0; *reinterpret_cast<u32x2_ma*>(d + 8) = p1; *reinterpret_cast<u32x2_ma*>(d + 16) = p2;
        }
      }
    };
    // Prologue: chunks 0 and 1 are requested together, but the consumers are released as soon as chunk 0 is staged
    // (round 2 staged both first: two splits and a second burst of loads in front of the first MFMA, ~4 us of a 40 us launch);
    // chunk 1 is split into buffer 1 while the consumers run steps 0-3 of chunk 0 - exactly what every later chunk does.
    load_A(d0, true, 0);
    load_A(d1, total_gc > 1, 1);
    wait_vm<NL>();                     // chunk 0 and, older in the queue, the weight DMA (only chunk 1's loads are younger)
    drop_setup();
    store_all(As, 0);
    load_A(d2, total_gc > 2, 0);
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();      // B0: weights, bias, buffer 0
    wait_vm<NL>();                     // chunk 1 (only chunk 2's loads are younger)
    store_all(As + G::A_DW, 1);
    load_A(d3, total_gc > 3, 1);
    // barrier k (k = 0 .. total_gc - 1; the consumer passes it at the head of step 4 of chunk k): buffer k & 1 is free,
    // buffer (k + 1) & 1 is complete.  Behind it: chunk k + 2 -> buffer k & 1, then the loads of chunk k + 4.
    Desc dn = d3; advance(dn);         // chunk k + 4
    for (int k = 0; k < total_gc; k += 2) {          // two chunks per trip: the register sets keep their roles at the loop edge
      wait_lgkm0();
      __builtin_amdgcn_s_barrier();
      wait_vm<NL>();                   // chunk k + 2 has landed (only the loads of chunk k + 3 are younger)
      store_all(As, 0);
      load_A(dn, k + 4 < total_gc, 0);
      advance(dn);
      if (k + 1 < total_gc) {
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        wait_vm<NL>();
        store_all(As + G::A_DW, 1);
        load_A(dn, k + 5 < total_gc, 1);
        advance(dn);
      }
    }
    wait_vm<0>();
    return;
  }

  // ================================================================== consumer waves
  const int tl = g >> 1;
  int aoff[5];
#pragma unroll
  for (int s_ = 0; s_ < 5; ++s_) {
    const int tap = 2 * s_ + tl > 8 ? 8 : 2 * s_ + tl;
    aoff[s_] = ((tap / 3) * 18 + tap % 3) * 24;
  }
  const int laneA = (wid * A_T * 18 + li) * 24 + (g & 1) * 4;
  const int laneB = (tl * BN + li) * 24 + (g & 1) * 4;             // within a step's [tapL][n] block
  bf16x8 fa[A_T][3], fb[2][C_T][3];
  f32x4 acc[A_T][C_T];
#pragma unroll
  for (int i = 0; i < A_T; ++i)
#pragma unroll
    for (int j = 0; j < C_T; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();        // B0
#pragma unroll
  for (int at = 0; at < A_T; ++at)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fa[at][pl] = lds_bf16x8(As + laneA + aoff[0] + at * 18 * 24 + 8 * pl);
#pragma unroll
  for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fb[0][ct][pl] = lds_bf16x8(Ws + laneB + ct * 16 * 24 + 8 * pl);

  auto chunk = [&](auto CP_, int gc, int c, bool more) {
    constexpr int CP = decltype(CP_)::value;               // B set of step 0 (5 steps per chunk: alternates per chunk)
    const unsigned* Acur = As + (gc & 1) * G::A_DW;
    const unsigned* Anxt = As + ((gc + 1) & 1) * G::A_DW;
    const unsigned* Wc = Ws + c * G::WCH_DW;
    const unsigned* Wn = Ws + (c + 1 == nchunks ? 0 : c + 1) * G::WCH_DW;
    auto step = [&](auto S_) {
      constexpr int S = decltype(S_)::value;
      constexpr int P = (CP + S) & 1, Q = P ^ 1;
      if (S == 4) {                    // the one rendezvous of the chunk: the next chunk's activations are complete
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
      }
      constexpr int BPA = (C_T * 3 + A_T - 1) / A_T;        // next-step B reads per pixel tile
      const unsigned* An = S < 4 ? Acur : Anxt;
      const unsigned* Wb = (S < 4 ? Wc : Wn) + ((S + 1) % 5) * (2 * BN * 24);
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
#pragma unroll
        for (int ct = 0; ct < C_T; ++ct) {                  // D = W . X^T; small terms first
#ifdef RW_NO_MFMA
          asm volatile("" :: "v"(fb[P][ct][0]), "v"(fb[P][ct][1]), "v"(fb[P][ct][2]), "v"(fa[at][0]), "v"(fa[at][1]), "v"(fa[at][2]));
          continue;
#endif
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][2]);
          mfma_acc(acc[at][ct], fb[P][ct][2], fa[at][0]);
          mfma_acc(acc[at][ct], fb[P][ct][1], fa[at][1]);
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][1]);
          mfma_acc(acc[at][ct], fb[P][ct][1], fa[at][0]);
          mfma_acc(acc[at][ct], fb[P][ct][0], fa[at][0]);
        }
#ifndef RW_NO_FRAG
        if (S < 4 || more) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[at][pl] = lds_bf16x8(An + laneA + aoff[(S + 1) % 5] + at * 18 * 24 + 8 * pl);
        }
#pragma unroll
        for (int k = at * BPA; k < (at + 1) * BPA && k < C_T * 3; ++k)
          fb[Q][k / 3][k % 3] = lds_bf16x8(Wb + laneB + (k / 3) * 16 * 24 + 8 * (k % 3));
#else
        (void)An; (void)Wb;
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
  };
  auto tile_end = [&]() {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s1[C_T][4], s2[C_T][4];
#pragma unroll
    for (int ct = 0; ct < C_T; ++ct) {
      const int n = ct * 16 + 4 * g;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; }
#pragma unroll
      for (int at = 0; at < A_T; ++at) {
        const long pix = ((long)d0.img * a.H + d0.y0 + wid * A_T + at) * a.W + d0.x0 + li;
        f32x4 v = acc[at][ct] + bv;
        if (n < a.N) {                                       // (N = 4 / 8 / 12 of the 16-wide block: the other lane groups idle)
          if (a.R) v += *reinterpret_cast<const f32x4*>(a.R + pix * a.ldr + n);
#ifdef RW_NO_STORE
          if (v[0] == 1.2345e30f)
#endif
          *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + n) = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
        acc[at][ct] = f32x4{0, 0, 0, 0};
      }
    }
    if (has_stats) {
      const long slab = (long)d0.mblk * NCW + wid, nslab = (long)a.n_mblocks * NCW;
#pragma unroll
      for (int ct = 0; ct < C_T; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v1 = row16_sum(s1[ct][r]), v2 = row16_sum(s2[ct][r]);
          if (li == 0 && ct * 16 + 4 * g < a.N) {
            a.stat_sum[(ct * 16 + 4 * g + r) * nslab + slab] = v1;
            a.stat_sq[(ct * 16 + 4 * g + r) * nslab + slab] = v2;
          }
        }
    }
  };
  for (int gc = 0; gc < total_gc; gc += 2) {
    chunk(std::integral_constant<int, 0>{}, gc, d0.c, gc + 1 < total_gc);
    if (d0.c + 1 == nchunks) tile_end();
    advance(d0);
    if (gc + 1 < total_gc) {
      chunk(std::integral_constant<int, 1>{}, gc + 1, d0.c, gc + 2 < total_gc);
      if (d0.c + 1 == nchunks) tile_end();
      advance(d0);
    }
  }
}

template <int A_T, int C_T, int NCW = 4>
static int launch_rw(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = RwGeom<A_T, C_T, NCW>;
  const int mblocks = a.NB * (a.H / G::TH) * (a.W / 16);
  if (q) { q[0] = NCW * mblocks; q[1] = 9350000 + (G::TH / 4) * 1000 + G::BN; q[2] = 1610; return ARCO_OK; }      // (id by tile height: <4,1,NCW=8> is <8,1>'s tile)
  if ((a.ldc & 3) != 0 || (a.R && (a.ldr & 3) != 0)) return ARCO_ERR_UNSUPPORTED;
  const size_t lds = G::lds_bytes((a.K + 15) >> 4);
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = 1;
  const int cus = conv_sp_cus();
  if (a.pro.mean) {           // consumer-side activation of the input (the block's first BatchNorm + LeakyReLU + dropout)
    if ((a.K & 15) != 0) return ARCO_ERR_UNSUPPORTED;
    auto kern = conv3x3_rw_kernel<A_T, C_T, true, NCW>;
    static unsigned long long attr_set = 0;
    if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks < cus ? mblocks : cus)), dim3((NCW + 4) * 64), lds, st, b);
    return arco_launch_status();
  }
  auto kern = conv3x3_rw_kernel<A_T, C_T, false, NCW>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks < cus ? mblocks : cus)), dim3((NCW + 4) * 64), lds, st, b);
  return arco_launch_status();
}

template <int A_T, int C_T>
static int launch_sp(const IgemmArgs& a, hipStream_t st, int* q) {
  using G = SpGeom<A_T, C_T>;
  const int mblocks = a.NB * (a.H / G::TH) * (a.W / 16);
  if (q) { q[0] = 4 * mblocks; q[1] = 9300000 + A_T * 1000 + G::BN; q[2] = 1610; return ARCO_OK; }      // 4 stat slabs per tile
  if ((a.ldc & 3) != 0 || (a.R && (a.ldr & 3) != 0)) return ARCO_ERR_UNSUPPORTED;
  IgemmArgs b = a;
  b.n_mblocks = mblocks; b.n_nblocks = a.Npad / G::BN;
  const int total = mblocks * b.n_nblocks, cus = conv_sp_cus();
  if (a.pro.mean) {           // consumer-side activation of the input (the block's first BatchNorm + LeakyReLU + dropout)
    auto kern = conv3x3_sp_kernel<A_T, C_T, true>;
    static unsigned long long attr_set = 0;
    if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), G::LDS_BYTES, st, b);
    return arco_launch_status();
  }
  auto kern = conv3x3_sp_kernel<A_T, C_T>;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
  hipLaunchKernelGGL(kern, dim3((unsigned)(total < cus ? total : cus)), dim3(512), G::LDS_BYTES, st, b);
  return arco_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------
// 3-D counterpart for the full-resolution level of the V-Net (3x3x3, 16 -> 16, planes with H % 16 == 0 and W % 16 == 0: block_nine and
// its data gradient, vnetWithArgs.py:222-238).  igemm_kernel<9,128,16,..,DEPTH=3> re-stages the 27 taps' weights for every tile
// (as many bytes as the activations at this width) and reads every input plane three times.  Here a persistent workgroup of 8 waves
//   * keeps ALL 27 taps' pre-split weights in LDS for the launch (41 KB),
//   * owns units of (volume, 16 x 16 tile, S consecutive planes) and walks the depth axis with a RING of three split input plane
//     tiles: output plane p needs only plane p + 1 as new input (loaded during plane p - 1's MFMAs, split on the way into LDS),
//   * pairs the 27 taps freely across the three resident planes: 14 MFMA steps per plane instead of 15.
// Same arithmetic as the other split-bf16 kernels (six v_mfma_f32_16x16x32_bf16 per fp32-accurate product, small terms first,
// D = W . X^T, 16-byte stores, BN partial statistics: one slab per plane tile).
// ---------------------------------------------------------------------------------------------------------------------------
struct Rw3Geom {
  static constexpr int TH = 16, HR = (TH + 2) * 18, NT = 512, NW = 8;
  static constexpr int WROWS = 27 * 16 + 1;                 // + the zero row
  static constexpr int W_DW = WROWS * 24, A_DW = HR * 24;
  static constexpr int NP = (HR * 4 + NT - 1) / NT;         // 16-byte activation pieces (4 channels) per thread and plane
  static constexpr size_t LDS_BYTES = (size_t)(W_DW + 3 * A_DW + 2 * NW * 16) * 4;
};

__global__ __launch_bounds__(512) void conv3d_rw16_kernel(IgemmArgs a, int S, int nseg) {
  using G = Rw3Geom;
  constexpr int HR = G::HR, WROWS = G::WROWS, NP = G::NP, NT = G::NT, NW = G::NW, TH = G::TH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* const Ws = reinterpret_cast<unsigned*>(smem);            // [WROWS][24]
  unsigned* const As = Ws + G::W_DW;                                 // [3][HR][24]
  float* const red = reinterpret_cast<float*>(As + 3 * G::A_DW);     // [2][NW][16]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4, h = g >> 1;
  const int tiles_x = a.W >> 4, tiles_y = a.H / TH;
  const int units = (a.NB / a.D3) * tiles_y * tiles_x * nseg;

  // weights: the split pack [tap][Npad = 16][Kg = 2][3][16] bf16 -> LDS rows [tap * 16 + n] of the first 16-k group (24 dwords)
  for (int i = tid; i < (WROWS - 1) * 6; i += NT) {
    const int r = i / 6, q6 = i - r * 6;
    *reinterpret_cast<u32x4_ma*>(Ws + r * 24 + q6 * 4) = *reinterpret_cast<const u32x4*>(a.Wp + ((long)r * a.Kg) * 24 + q6 * 4);
  }
  for (int i = tid; i < 24; i += NT) Ws[(WROWS - 1) * 24 + i] = 0u;

  int prow[NP], pq[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int idx = tid + i * NT;
    prow[i] = idx < HR * 4 ? idx >> 2 : -1; pq[i] = idx & 3;
  }
  f32x4 bv;
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = a.bias ? a.bias[4 * g + r] : 0.f;

  f32x4 R[NP];
  const bool xmap = (gridDim.x & 7) == 0;         // XCD-aware unit order: a contiguous eighth of the units per XCD (shared halos meet in one L2)
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
        f32x4 val = f32x4{0, 0, 0, 0};
        if (prow[i] >= 0 && pok) {
          const int hy = prow[i] / 18, hx = prow[i] - hy * 18;
          const int y = y0 + hy - 1, x = x0 + hx - 1;
          if (y >= 0 && y < a.H && x >= 0 && x < a.W)
            val = *reinterpret_cast<const f32x4*>(a.A + ((((long)v * a.D3 + p) * a.H + y) * a.W + x) * a.lda + pq[i] * 4);
        }
        R[i] = val;
      }
    };
    auto store_plane = [&](int slot) {           // split into the three bf16 planes of the row's 16-k group
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (prow[i] >= 0) {
          u32x2 p0_, p1_, p2_;
          split3_bf16x4(R[i], p0_, p1_, p2_);
          unsigned* d = As + slot * G::A_DW + prow[i] * 24 + pq[i] * 2;
          *reinterpret_cast<u32x2_ma*>(d) = p0_; *reinterpret_cast<u32x2_ma*>(d + 8) = p1_; *reinterpret_cast<u32x2_ma*>(d + 16) = p2_;
        }
    };
    f32x4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
    int sA = 0, sB = 1, sC = 2;                  // ring slots of planes p - 1, p, p + 1
    load_plane(p0 - 1); store_plane(sA);
    load_plane(p0); store_plane(sB);
    load_plane(p0 + 1);
    for (int p = p0; p < p1; ++p) {
      store_plane(sC);
      __syncthreads();
      if (p + 1 < p1) load_plane(p + 2);         // in flight during this plane's MFMAs
      f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      const int sb0 = sA * G::A_DW, sb1 = sB * G::A_DW, sb2 = sC * G::A_DW;
      // the fragments of step s + 1 are read in front of step s's MFMAs (two register sets): no LDS round trip between steps
      bf16x8 fb[2][3], fa[2][2][3];
      auto read_step = [&](int sp_, int set) {
        // lanes g = 0,1 take tap 2s, g = 2,3 tap 2s + 1 (tap 27: the zero weight row on tap 26's activations)
        const int tp0 = 2 * sp_, tp1 = 2 * sp_ + 1 > 26 ? 26 : 2 * sp_ + 1;
        const int dz0 = tp0 / 9, dz1 = tp1 / 9;
        const int o0 = ((tp0 % 9) / 3) * 18 + tp0 % 3, o1 = ((tp1 % 9) / 3) * 18 + tp1 % 3;
        const int base0 = (dz0 == 0 ? sb0 : (dz0 == 1 ? sb1 : sb2)) + o0 * 24;
        const int base1 = (dz1 == 0 ? sb0 : (dz1 == 1 ? sb1 : sb2)) + o1 * 24;
        const int abase = (h ? base1 : base0) + (g & 1) * 4;
        const int wrow = h ? (2 * sp_ + 1 > 26 ? WROWS - 1 : tp1 * 16 + li) : tp0 * 16 + li;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[set][pl] = lds_bf16x8(Ws + wrow * 24 + (g & 1) * 4 + 8 * pl);
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[set][at][pl] = lds_bf16x8(As + abase + ((2 * wid + at) * 18 + li) * 24 + 8 * pl);
      };
      read_step(0, 0);
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        if (s + 1 < 14) read_step(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int at = 0; at < 2; ++at) {         // D = W . X^T; small terms first
          f32x4 c = acc[at];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][0], fa[s & 1][at][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][2], fa[s & 1][at][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][1], fa[s & 1][at][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][0], fa[s & 1][at][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][1], fa[s & 1][at][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s & 1][0], fa[s & 1][at][0], c, 0, 0, 0);
          acc[at] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int at = 0; at < 2; ++at) {
        const long pix = ((((long)v * a.D3 + p) * a.H + y0 + 2 * wid + at) * a.W + x0 + li);
        const f32x4 o = acc[at] + bv;
        *reinterpret_cast<f32x4*>(a.C + pix * a.ldc + 4 * g) = o;
        t1 += o; t2 += o * o;
      }
      __syncthreads();                           // slot sA (plane p - 1) is free from here on
      const int t = sA; sA = sB; sB = sC; sC = t;
    }
    if (a.stat_sum) {       // the unit's sums go to the slab of its first plane tile, zeros to the slabs of its other planes
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v1 = row16_sum(t1[r]), v2 = row16_sum(t2[r]);
        if (li == 0) { red[(0 * NW + wid) * 16 + 4 * g + r] = v1; red[(1 * NW + wid) * 16 + 4 * g + r] = v2; }
      }
      __syncthreads();
      if (tid < 16) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { v1 += red[(0 * NW + w) * 16 + tid]; v2 += red[(1 * NW + w) * 16 + tid]; }
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

// returns -1 when the shape is not taken (the caller falls through to igemm_kernel)
int conv3d_rw_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  static const bool off = getenv("ARCO_CONV3D_RW") && atoi(getenv("ARCO_CONV3D_RW")) == 0;      // A/B switch
  if (off || a.mma != 3 || a.K != 16 || a.N != 16 || a.Npad != 16 || a.Kg != 2 || (a.H & 15) != 0 || (a.W & 15) != 0 ||
      (a.lda & 3) != 0 || (a.ldc & 3) != 0 || a.R != nullptr) return -1;
  using G = Rw3Geom;
  const long tiles = (long)(a.H / G::TH) * (a.W >> 4);
  if (q) { q[0] = (int)(a.NB * tiles); q[1] = 9 * 1000000 + 450000 + 16; q[2] = 16 * 100 + 30 + 1; return ARCO_OK; }
  if (a.D3 < 1 || a.NB % a.D3 != 0) return ARCO_ERR_ARG;
  const long cols = (long)(a.NB / a.D3) * tiles, slots = conv_sp_cus();
  int S = 8;
  while (S > 2 && cols * ((a.D3 + S - 1) / S) < 3 * slots) S >>= 1;
  if (S > a.D3) S = a.D3;
  const int nseg = (a.D3 + S - 1) / S;
  const long units = cols * nseg;
  static unsigned long long attr_set = 0;
  if (arco_first_on_device(attr_set)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_rw16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES); }
  IgemmArgs b = a;
  b.n_mblocks = (int)(a.NB * tiles);
  hipLaunchKernelGGL(conv3d_rw16_kernel, dim3((unsigned)(slots < units ? slots : units)), dim3(G::NT), G::LDS_BYTES, st, b, S, nseg);
  return arco_launch_status();
}

// A/B knob: ARCO_CONV_SP=0 / arco_conv_sp_set(0) keeps every shape on igemm_kernel
static int& conv_sp_flag() { static int on = !(getenv("ARCO_CONV_SP") && atoi(getenv("ARCO_CONV_SP")) == 0); return on; }
static bool conv_sp_on() { return conv_sp_flag() != 0; }
extern "C" int arco_conv_sp_set(int on) { const int prev = conv_sp_flag(); conv_sp_flag() = on ? 1 : 0; return prev; }

template <int C_T>
static int dispatch_rows(const IgemmArgs& a, hipStream_t st, int* q, int min_tiles) {
  // the tallest tile (4 A_T rows x 16 columns per workgroup) that still gives min_tiles work items: the deep levels have
  // few pixels (32 x 32, 16 x 16 per image) and many channels
  const long cols = (long)a.NB * (a.W / 16) * (a.N / (16 * C_T));
  if (cols * (a.H / 16) >= min_tiles) return launch_sp<4, C_T>(a, st, q);
  if ((a.H & 7) == 0 && cols * (a.H / 8) >= min_tiles) return launch_sp<2, C_T>(a, st, q);
  if ((a.H & 3) == 0 && cols * (a.H / 4) >= min_tiles) return launch_sp<1, C_T>(a, st, q);
  return -1;
}

int conv_sp_dispatch(const IgemmArgs& a, hipStream_t st, int* q) {
  if (!conv_sp_on() || a.mma != 3 || a.D3 != 1) return -1;
  if ((a.lda & 3) != 0 || (a.W & 15) != 0 || (a.H & 3) != 0) return -1;
  static const int min_tiles = getenv("ARCO_CONV_SP_TILES") ? atoi(getenv("ARCO_CONV_SP_TILES")) : 192;
  static const int min_n = getenv("ARCO_CONV_SP_MINN") ? atoi(getenv("ARCO_CONV_SP_MINN")) : 16;
  static const int rw = getenv("ARCO_CONV_RW") ? atoi(getenv("ARCO_CONV_RW")) : 1;
  static const int narrow = getenv("ARCO_CONV_RW_NARROW") ? atoi(getenv("ARCO_CONV_RW_NARROW")) : 1;
  // one 16-wide output block from at most 32 inputs: also the few-channel ends of the network (the 16 -> 4 logits layer, its
  // 4 -> 16 data gradient) - HBM-bound launches, the idle MFMA columns / zero channels cost nothing
  if (rw && a.Npad == 16 && (a.N & 3) == 0 && (a.N == 16 || narrow) && a.K <= 32 && (a.K & 3) == 0 && ((a.K & 15) == 0 || (a.K < 16 && narrow)) &&
      (a.H & 31) == 0 && (long)a.NB * (a.H / 32) * (a.W / 16) >= min_tiles) {
    // eight MFMA waves (two per SIMD) on the same 32 x 16 tiles (round 6): a one-chunk tile (K <= 16) is 240 MFMAs against an epilogue of the same
    // order, and the second wave's MFMAs run under it - tools/micro/rw_bench.py (inputs from HBM): 52.5 -> 49.2 us at 16 -> 16 @256^2 x16, 47.3 -> 43.7 at
    // 16 -> 4, 40.7 -> 38.2 at 4 -> 16, 40.2 -> 38.9 at 32 -> 32 @128^2 (<2,2,8>), 81 -> 82-87 at 32 -> 16 (two chunks per tile).  In bench.py's event-timed
    // pass the family averages 43.4 us with four waves, 41.7 with eight at K <= 16, 40.3 with eight everywhere (roofline.frac 0.313 / 0.327 / 0.340);
    // the replayed step 10.57 / 10.48 / 10.52 ms (five alternations, one box).  ARCO_CONV_RW8 = 0: four waves, 1: eight at K <= 16, 2 (default): eight
    static const int rw8 = getenv("ARCO_CONV_RW8") ? atoi(getenv("ARCO_CONV_RW8")) : 2;
    if (rw8 == 2 || (rw8 == 1 && a.K <= 16)) return launch_rw<4, 1, 8>(a, st, q);
    return launch_rw<8, 1>(a, st, q);
  }
  if ((a.K & 15) != 0 || a.K < 16 || (a.N & 15) != 0 || a.N != a.Npad || a.N > 256) return -1;
  if (a.N < min_n) return -1;
  if (rw && a.K <= 32 && a.N == 32) {                           // shallow levels: resident weights, one rendezvous per chunk
    static const int rw8b = getenv("ARCO_CONV_RW8") ? atoi(getenv("ARCO_CONV_RW8")) : 2;
    if ((a.H & 15) == 0 && (long)a.NB * (a.H / 16) * (a.W / 16) >= min_tiles) return rw8b == 2 ? launch_rw<2, 2, 8>(a, st, q) : launch_rw<4, 2>(a, st, q);
  }
  if ((a.N & 63) == 0) return dispatch_rows<4>(a, st, q, min_tiles);
  if ((a.N & 31) == 0) return dispatch_rows<2>(a, st, q, min_tiles);
  return (a.H & 15) == 0 ? launch_sp<4, 1>(a, st, q) : -1;
}
