// Loss front-end: per-pixel class masks, stable
// compaction lists, masked prototype means, key gather, FIFO bank append.
// All HBM-bound byte/integer work: one pixel per lane, coalesced plane reads,
// wave-ballot histograms, no atomics on the ordering path (stable compaction).
#include "common.h"

// ---------------------------------------------------------------------------
// (1) mask codes + per-block counts.   Reference: loss_helper_3d.py:341-342,
// 352-358 (rank of each class in a descending stable sort), :364-401 (masks).
// Inputs are NC[spatial] planes exactly as the reference receives them.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_codes_kernel(
    const int64_t* __restrict__ lab_l, const int64_t* __restrict__ lab_u,
    const float* __restrict__ prob_l, const float* __restrict__ prob_u,
    const float* __restrict__ low_mask, const float* __restrict__ high_mask,
    int n_l_img, int C, long P, long n_pix, float delta_p, float delta_n, int low_rank, int high_rank,
    uint64_t* __restrict__ codes, uint32_t* __restrict__ block_counts, int nblocks) {
  const long gp = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t code = 0;
  if (gp < n_pix) {
    const long img = gp / P, s = gp - img * P;
    const bool labeled = img < n_l_img;
    const int64_t* lab = labeled ? lab_l + (img * C) * P + s : lab_u + ((img - n_l_img) * C) * P + s;
    const float* prob = labeled ? prob_l + (img * C) * P + s : prob_u + ((img - n_l_img) * C) * P + s;
    const float lowm = low_mask[gp], highm = high_mask[gp];
    float p[ARCO_MAXC];
    int64_t lb[ARCO_MAXC];
#pragma unroll
    for (int c = 0; c < ARCO_MAXC; ++c) {
      p[c] = c < C ? prob[(long)c * P] : 0.f;
      lb[c] = c < C ? lab[(long)c * P] : 0;
    }
#pragma unroll
    for (int c = 0; c < ARCO_MAXC; ++c) {
      if (c < C) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < ARCO_MAXC; ++j)
          if (j < C) rank += (p[j] > p[c]) || (p[j] == p[c] && j < c);
        const float labf = (float)lb[c];
        const bool lv = (labf * lowm) != 0.f;
        const bool hv = (labf * highm) != 0.f;
        const bool anchor = (p[c] > delta_p) && lv;
        const bool hard = (p[c] < delta_n) && hv;
        const bool cls = labeled ? (rank < low_rank && lb[c] == 0) : (rank >= low_rank && rank < high_rank);
        const bool neg = hard && cls;
        code |= (uint64_t)lv << ARCO_BIT_LV(c);
        code |= (uint64_t)anchor << ARCO_BIT_ANCHOR(c);
        code |= (uint64_t)neg << ARCO_BIT_NEG(c);
      }
    }
    codes[gp] = code;
  }
  __shared__ uint32_t cnt[3 * ARCO_MAXC];
  if (threadIdx.x < 3 * ARCO_MAXC) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const uint64_t b0 = __ballot((code >> ARCO_BIT_LV(c)) & 1);
    const uint64_t b1 = __ballot((code >> ARCO_BIT_ANCHOR(c)) & 1);
    const uint64_t b2 = __ballot((code >> ARCO_BIT_NEG(c)) & 1);
    if (lane == 0) {
      atomicAdd(&cnt[c], (uint32_t)__popcll(b0));
      atomicAdd(&cnt[ARCO_MAXC + c], (uint32_t)__popcll(b1));
      atomicAdd(&cnt[2 * ARCO_MAXC + c], (uint32_t)__popcll(b2));
    }
  }
  __syncthreads();
  (void)wid;
  if (threadIdx.x < 3 * C) {
    const int kind = threadIdx.x / C, c = threadIdx.x % C;
    block_counts[(long)(kind * C + c) * nblocks + blockIdx.x] = cnt[kind * ARCO_MAXC + c];
  }
}

// (2) exclusive scan of each counter column; one block per (kind, class).
__global__ __launch_bounds__(256) void scan_counts_kernel(const uint32_t* __restrict__ counts, int nblocks,
                                                         uint32_t* __restrict__ offsets, int64_t* __restrict__ totals) {
  const uint32_t* col = counts + (long)blockIdx.x * nblocks;
  uint32_t* out = offsets + (long)blockIdx.x * nblocks;
  const int per = (nblocks + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
  uint32_t s = 0;
  for (int i = b0; i < b1; ++i) s += col[i];
  __shared__ uint32_t part[256];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { uint32_t t = part[i]; part[i] = run; run += t; }
    totals[blockIdx.x] = run;
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int i = b0; i < b1; ++i) { out[i] = run; run += col[i]; }
}

// (3) stable compaction: row ids of anchor / negative pixels per class, in
// (b, spatial) row-major order == torch boolean-mask order (loss_helper_3d.py:377,403).
__global__ __launch_bounds__(256) void compact_rows_kernel(const uint64_t* __restrict__ codes, long n_pix, int C,
                                                          const uint32_t* __restrict__ offsets, int nblocks,
                                                          int32_t* __restrict__ lists /*[2C][n_pix]*/) {
  const long gp = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint64_t code = gp < n_pix ? codes[gp] : 0;
  __shared__ uint32_t wcnt[4][2 * ARCO_MAXC];
  for (int c = 0; c < C; ++c) {
    const uint64_t b1 = __ballot((code >> ARCO_BIT_ANCHOR(c)) & 1);
    const uint64_t b2 = __ballot((code >> ARCO_BIT_NEG(c)) & 1);
    if (lane == 0) { wcnt[wid][c] = __popcll(b1); wcnt[wid][ARCO_MAXC + c] = __popcll(b2); }
  }
  __syncthreads();
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      const int bit = kind == 0 ? ARCO_BIT_ANCHOR(c) : ARCO_BIT_NEG(c);
      const bool on = (code >> bit) & 1;
      const uint64_t b = __ballot(on);
      if (on) {
        uint32_t base = offsets[(long)((kind + 1) * C + c) * nblocks + blockIdx.x];
        for (int w = 0; w < wid; ++w) base += wcnt[w][kind * ARCO_MAXC + c];
        lists[(long)(kind * C + c) * n_pix + base + __popcll(b & lt)] = (int32_t)gp;
      }
    }
  }
}

// (4) prototype partial sums: sum over low-valid pixels of teacher rows
// (loss_helper_3d.py:380-384).  Rows are channels-last [n_pix][ldt]; LPR lanes
// cover one row with float4 loads, 64/LPR rows per wave step.  Deterministic:
// per-block slabs, fixed-order finalize.
template <int NDI>
__global__ __launch_bounds__(256) void masked_row_sum_kernel(const float* __restrict__ T, long ldt,
                                                            const uint64_t* __restrict__ codes, long n_pix, int D,
                                                            int c0, int nc, int lpr, long rows_per_block,
                                                            float* __restrict__ partial /*[grid][nc][D]*/) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][8][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, dl = lane % lpr;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(n_pix, r0 + rows_per_block);
  f32x4 acc[8][NDI];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < NDI; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  const uint32_t cmask = (1u << nc) - 1u;
  for (long base = r0 + (long)wid * rpw; base < r1; base += 4 * rpw) {
    const long row = base + sub;
    uint32_t bits = 0;
    if (row < r1) bits = (uint32_t)(codes[row] >> c0) & cmask;
    if (bits) {
      const float* src = T + row * ldt;
#pragma unroll
      for (int di = 0; di < NDI; ++di) {
        const int d = (di * lpr + dl) * 4;
        if (d < D) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + d);
#pragma unroll
          for (int cc = 0; cc < 8; ++cc)
            if ((bits >> cc) & 1) acc[cc][di] += v;
        }
      }
    }
  }
  // reduce the 64/lpr row-subgroups of the wave
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
#pragma unroll
      for (int di = 0; di < NDI; ++di)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[cc][di][e] += __shfl_xor(acc[cc][di][e], o, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)          // static indices: a runtime-bounded loop sent the accumulators to scratch memory
#pragma unroll
      for (int di = 0; di < NDI; ++di) {
        const int d = (di * lpr + dl) * 4;
        if (cc < nc && d < D) *reinterpret_cast<f32x4*>(&red[((long)wid * 8 + cc) * D + d]) = acc[cc][di];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc * D; i += 256) {
    const int cc = i / D, d = i % D;
    const float s = ((red[(0 * 8 + cc) * (long)D + d] + red[(1 * 8 + cc) * (long)D + d]) +
                     (red[(2 * 8 + cc) * (long)D + d] + red[(3 * 8 + cc) * (long)D + d]));
    partial[((long)blockIdx.x * nc + cc) * D + d] = s;
  }
}

// low-valid bits -> float weight rows [n_pix][Cp] (1.0 / 0.0), input of the bilinear adjoint
__global__ void lv_weights_kernel(const uint64_t* __restrict__ codes, long n_pix, int C, int Cp, float* __restrict__ W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pix * Cp) return;
  const long p = i / Cp; const int c = (int)(i - p * Cp);
  W[i] = (c < C && ((codes[p] >> ARCO_BIT_LV(c)) & 1)) ? 1.f : 0.f;
}

// weighted row sums: out[c][:] = sum_rows Wt[row][c0+c] * T[row][:]   (same tiling as masked_row_sum)
template <int NDI, typename TS>
__global__ __launch_bounds__(256) void weighted_row_sum_kernel(const TS* __restrict__ T, long ldt,
                                                              const float* __restrict__ Wt, long ldw, long n_rows, int D,
                                                              int c0, int nc, int lpr, long rows_per_block,
                                                              float* __restrict__ partial /*[grid][nc][D]*/) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][8][D]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rpw = 64 / lpr, sub = lane / lpr, dl = lane % lpr;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
  f32x4 acc[8][NDI];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < NDI; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  for (long base = r0 + (long)wid * rpw; base < r1; base += 4 * rpw) {
    const long row = base + sub;
    float w[8]; bool any = false;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) { w[cc] = (row < r1 && cc < nc) ? Wt[row * ldw + c0 + cc] : 0.f; any |= (w[cc] != 0.f); }
    if (any) {
      const TS* src = T + row * ldt;
#pragma unroll
      for (int di = 0; di < NDI; ++di) {
        const int d = (di * lpr + dl) * 4;
        if (d < D) {
          const f32x4 v = ld4f(src + d);
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) if (w[cc] != 0.f) acc[cc][di] += v * w[cc];
        }
      }
    }
  }
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
#pragma unroll
      for (int di = 0; di < NDI; ++di)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[cc][di][e] += __shfl_xor(acc[cc][di][e], o, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)          // static indices: a runtime-bounded loop sent the accumulators to scratch memory
#pragma unroll
      for (int di = 0; di < NDI; ++di) {
        const int d = (di * lpr + dl) * 4;
        if (cc < nc && d < D) *reinterpret_cast<f32x4*>(&red[((long)wid * 8 + cc) * D + d]) = acc[cc][di];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc * D; i += 256) {
    const int cc = i / D, d = i % D;
    const float s = ((red[(0 * 8 + cc) * (long)D + d] + red[(1 * 8 + cc) * (long)D + d]) +
                     (red[(2 * 8 + cc) * (long)D + d] + red[(3 * 8 + cc) * (long)D + d]));
    partial[((long)blockIdx.x * nc + cc) * D + d] = s;
  }
}
// out[(c0+cc)*ldo + d] = (sum over slabs) * (totals ? 1/totals[c0+cc] : 1)
__global__ __launch_bounds__(1024) void row_sum_finalize_kernel(const float* __restrict__ partial, int nblk, int nc, int D, int c0,
                                        const int64_t* __restrict__ totals, float* __restrict__ out, long ldo) {
  // block = 64 outputs x 16 slab segments (threadIdx.y): segment y sums slabs y, y+16, ... (two chains in flight), the 16
  // partial sums are combined through LDS in a fixed order (deterministic)
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool ok = i < nc * D;
  const int cc = ok ? i / D : 0, d = ok ? i % D : 0;
  double s0 = 0.0, s1 = 0.0;
  if (ok) {
    int b = threadIdx.y;
    for (; b + 16 < nblk; b += 32) {
      s0 += (double)partial[((long)b * nc + cc) * D + d]; s1 += (double)partial[((long)(b + 16) * nc + cc) * D + d];
    }
    for (; b < nblk; b += 16) s0 += (double)partial[((long)b * nc + cc) * D + d];
  }
  __shared__ double sh[16][64];
  sh[threadIdx.y][threadIdx.x] = s0 + s1;
  __syncthreads();
  if (threadIdx.y != 0 || !ok) return;
  double s = 0.0;
#pragma unroll
  for (int y = 0; y < 16; y += 4) s += (sh[y][threadIdx.x] + sh[y + 1][threadIdx.x]) + (sh[y + 2][threadIdx.x] + sh[y + 3][threadIdx.x]);
  if (totals) s /= (double)totals[c0 + cc];
  out[(long)(c0 + cc) * ldo + d] = (float)s;
}

__global__ void proto_finalize_kernel(const float* __restrict__ partial, int nblk, int nc, int D, int c0,
                                      const int64_t* __restrict__ totals /*[3C], lv counts first*/,
                                      float* __restrict__ proto /*[C][D]*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc * D) return;
  const int cc = i / D, d = i % D;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partial[((long)b * nc + cc) * D + d];
  const double n = (double)totals[c0 + cc];
  proto[(long)(c0 + cc) * D + d] = (float)(s / n);  // 0/0 -> NaN like torch.mean of an empty set
}

// (5) row gather: out[j] = src[list ? list[idx[j]] : idx[j]]   (rows of D floats)
template <typename TS>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TS* __restrict__ src, long lds_, int D,
                                                         const int32_t* __restrict__ list,
                                                         const int64_t* __restrict__ idx64,
                                                         const int32_t* __restrict__ idx32, long first, long n,
                                                         float* __restrict__ out, long ldo) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  long r = idx64 ? idx64[first + j] : (idx32 ? (long)idx32[first + j] : first + j);
  if (list) r = list[r];
  const TS* s = src + r * lds_;
  float* o = out + j * ldo;
  if ((D & 3) == 0) {
    for (int d = lane * 4; d < D; d += 256) *reinterpret_cast<f32x4*>(o + d) = ld4f(s + d);
  } else {
    for (int d = lane; d < D; d += 64) o[d] = (float)s[d];
  }
}

// (6) FIFO-by-truncation bank append (loss_helper_3d.py:23-28):
// out = cat(old, keys)[-min(len_old+n, Q):]
__global__ void bank_append_kernel(const float* __restrict__ old, long len_old, const float* __restrict__ keys,
                                   long n, long drop, long out_len, int D, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = out_len * D;
  if (i >= tot) return;
  const long j = i / D, d = i - j * D;
  const long srcj = j + drop;
  out[i] = srcj < len_old ? old[srcj * D + d] : keys[(srcj - len_old) * D + d];
}

// ---------------------------------------------------------------------------
// InfoNCE pieces (L6, loss_helper_3d.py:503-509)
// ---------------------------------------------------------------------------
// normalized copy: y = x / max(||x||, eps); optional transposed copy yt[D][n]; inv norm out
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, long ldx, long n, int D,
                                                            float eps, float* __restrict__ y, long ldy,
                                                            float* __restrict__ yt, long ldyt,
                                                            float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float* s = x + j * ldx;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = s[d]; ss += v * v; }
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  const float iv = 1.0f / fmaxf(nrm, eps);
  if (lane == 0 && inv) inv[j] = iv;
  for (int d = lane; d < D; d += 64) {
    const float v = s[d] * iv;
    if (y) y[j * ldy + d] = v;
    if (yt) yt[(long)d * ldyt + j] = v;
  }
}

// multiplicity matrix M[q][k] += 1 for every sampled negative (exact integer atomics)
__global__ void neg_multiplicity_kernel(const int64_t* __restrict__ idx, int Q, int Nn, long L, long ld,
                                        uint32_t* __restrict__ M) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Q * Nn) return;
  const long q = i / Nn;
  long k = idx[i];
  if (k < 0) k += L;
  atomicAdd(&M[q * ld + k], 1u);
}

// forward + weight matrix: one block per query.
//   lse_q = log( exp(pos/T) + sum_k M[q,k] exp(S[q,k]/T) ),  loss_q = lse_q - pos/T
//   W[q,k] = M[q,k] * exp(S[q,k]/T - lse_q) / T            (d loss_q / d S[q,k])
//   gpos[q] = (exp(pos/T - lse_q) - 1) / T                   (d loss_q / d pos)
__global__ __launch_bounds__(256) void infonce_fwd_kernel(const float* __restrict__ S, long lds_,
                                                         const uint32_t* __restrict__ M, long L,
                                                         const float* __restrict__ An, const float* __restrict__ Pn_,
                                                         long ldp, int D, float inv_temp, float* __restrict__ W,
                                                         float* __restrict__ gpos, float* __restrict__ loss_q) {
  const int q = blockIdx.x;
  const float* s = S + (long)q * lds_;
  const uint32_t* m = M + (long)q * lds_;
  const float* Pn = Pn_ + (long)q * ldp;
  __shared__ float sh[8];
  // positive logit = <An[q], Pn>
  float dp = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) dp += An[(long)q * D + d] * Pn[d];
  dp = wave_sum(dp);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dp;
  __syncthreads();
  const float pos = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  float mx = pos * inv_temp;
  for (long k = threadIdx.x; k < L; k += 256)
    if (m[k]) mx = fmaxf(mx, s[k] * inv_temp);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  double se = 0.0;
  for (long k = threadIdx.x; k < L; k += 256) {
    const uint32_t mk = m[k];
    if (mk) se += (double)mk * (double)expf(s[k] * inv_temp - mx);
  }
  se = wave_sum_d(se);
  __shared__ double shd[4];
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = se;
  __syncthreads();
  const double tot = ((shd[0] + shd[1]) + (shd[2] + shd[3])) + (double)expf(pos * inv_temp - mx);
  const float lse = mx + (float)log(tot);
  if (threadIdx.x == 0) {
    loss_q[q] = lse - pos * inv_temp;
    gpos[q] = (expf(pos * inv_temp - lse) - 1.0f) * inv_temp;
  }
  for (long k = threadIdx.x; k < L; k += 256) {
    const uint32_t mk = m[k];
    W[(long)q * lds_ + k] = mk ? (float)mk * expf(s[k] * inv_temp - lse) * inv_temp : 0.f;
  }
}

// ---------------------------------------------------------------------------
// Grouped InfoNCE: all classes (entries) of a step in one launch per stage (blockIdx.y / z = entry).
// Per-entry data travels by value in NceTable: bank pointer + length, prototype row, first sampled index.
// ---------------------------------------------------------------------------
struct NceTable {
  const float* bank[ARCO_MAXC];
  int len[ARCO_MAXC];       // bank rows of the entry
  int prow[ARCO_MAXC];      // row of the entry's positive in the normalised prototype matrix
};

// y[j][0..Dp) = x[j] / max(||x[j]||, eps), zero in the pad columns; inv[j] = 1 / max(||x||, eps)
__global__ __launch_bounds__(256) void normalize_rows_pad_kernel(const float* __restrict__ x, long ldx, long n, int D, int Dp,
                                                                float eps, float* __restrict__ y, long ldy,
                                                                float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float* s = x + j * ldx;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = s[d]; ss += v * v; }
  ss = wave_sum(ss);
  const float iv = 1.0f / fmaxf(sqrtf(ss), eps);
  if (lane == 0 && inv) inv[j] = iv;
  for (int d = lane; d < Dp; d += 64) y[j * ldy + d] = d < D ? s[d] * iv : 0.f;
}

// Bn[e][j][0..Dp) (rows >= len and pad columns zero) and its transpose Bt[e][d][j] (nullable), j < Lp.
// One block = 16 bank rows: each wave normalises 4 rows (coalesced row writes), the 16 x Dp tile goes through LDS so that the
// transposed copy is written as 64-byte segments along j (it was one 4-byte store per element at stride Lp: 82 -> ~25 us).
__global__ __launch_bounds__(256) void normalize_banks_kernel(NceTable t, int D, int Dp, long Lp, float eps,
                                                             float* __restrict__ Bn, float* __restrict__ Bt) {
  extern __shared__ float tile[];                    // [16][Dp + 1]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, e = blockIdx.y;
  const long j0 = (long)blockIdx.x * 16;
  const int LT = Dp + 1;
  for (int r = 0; r < 4; ++r) {
    const int jj = 4 * w + r;
    const long j = j0 + jj;
    if (j >= Lp) break;
    float* y = Bn + ((long)e * Lp + j) * Dp;
    if (j >= t.len[e]) {
      for (int d = lane; d < Dp; d += 64) { y[d] = 0.f; tile[jj * LT + d] = 0.f; }
      continue;
    }
    const float* s = t.bank[e] + j * (long)D;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = s[d]; ss += v * v; }
    ss = wave_sum(ss);
    const float iv = 1.0f / fmaxf(sqrtf(ss), eps);
    for (int d = lane; d < Dp; d += 64) {
      const float v = d < D ? s[d] * iv : 0.f;
      y[d] = v;
      tile[jj * LT + d] = v;
    }
  }
  if (!Bt) return;
  __syncthreads();
  float* bt = Bt + (long)e * Dp * Lp;
  for (int idx = threadIdx.x; idx < Dp * 16; idx += 256) {
    const int d = idx >> 4, jj = idx & 15;
    if (j0 + jj < Lp) bt[(long)d * Lp + j0 + jj] = tile[jj * LT + d];
  }
}

// One block per (query, entry): negative multiplicities built in LDS (16-bit counters, two per word) from the entry's
// Nn sampled indices, then exactly the arithmetic of infonce_fwd_kernel.  idx of entry e: idx_all + e*idx_stride + idx_off.
__global__ __launch_bounds__(256) void infonce_fused_kernel(const float* __restrict__ S, long ld, NceTable t,
                                                           const int64_t* __restrict__ idx_all, long idx_off, long idx_stride,
                                                           int Q, int Nn, const float* __restrict__ An,
                                                           const float* __restrict__ Pn_all, int Dp, float inv_temp,
                                                           float* __restrict__ W, float* __restrict__ gpos,
                                                           float* __restrict__ loss_q) {
  extern __shared__ uint32_t cnt[];                  // ceil(L / 2) words
  const int q = blockIdx.x, e = blockIdx.y;
  const long L = t.len[e];
  const long row = (long)e * Q + q;
  const float* s = S + row * ld;
  const float* An_q = An + row * Dp;
  const float* Pn = Pn_all + (long)t.prow[e] * Dp;
  for (long k = threadIdx.x; k < (L + 1) / 2; k += 256) cnt[k] = 0u;
  __syncthreads();
  const int64_t* idx = idx_all + (long)e * idx_stride + idx_off + (long)q * Nn;
  for (int i = threadIdx.x; i < Nn; i += 256) {
    long k = idx[i];
    if (k < 0) k += L;
    atomicAdd(&cnt[k >> 1], 1u << (16 * (int)(k & 1)));
  }
  __shared__ float sh[8];
  float dp = 0.f;
  for (int d = threadIdx.x; d < Dp; d += 256) dp += An_q[d] * Pn[d];
  dp = wave_sum(dp);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dp;
  __syncthreads();
  const float pos = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  float mx = pos * inv_temp;
  for (long k = threadIdx.x; k < L; k += 256)
    if ((cnt[k >> 1] >> (16 * (int)(k & 1))) & 0xffffu) mx = fmaxf(mx, s[k] * inv_temp);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  double se = 0.0;
  for (long k = threadIdx.x; k < L; k += 256) {
    const uint32_t mk = (cnt[k >> 1] >> (16 * (int)(k & 1))) & 0xffffu;
    if (mk) se += (double)mk * (double)expf(s[k] * inv_temp - mx);
  }
  se = wave_sum_d(se);
  __shared__ double shd[4];
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = se;
  __syncthreads();
  const double tot = ((shd[0] + shd[1]) + (shd[2] + shd[3])) + (double)expf(pos * inv_temp - mx);
  const float lse = mx + (float)log(tot);
  if (threadIdx.x == 0) {
    loss_q[row] = lse - pos * inv_temp;
    gpos[row] = (expf(pos * inv_temp - lse) - 1.0f) * inv_temp;
  }
  if (W) {
    float* w = W + row * ld;
    for (long k = threadIdx.x; k < ld; k += 256) {       // the whole padded row: columns >= L are zero (GEMM K range)
      uint32_t mk = 0;
      if (k < L) mk = (cnt[k >> 1] >> (16 * (int)(k & 1))) & 0xffffu;
      w[k] = mk ? (float)mk * expf(s[k] * inv_temp - lse) * inv_temp : 0.f;
    }
  }
}

// anchor gradient of every entry in one launch: row r = e*Q + q uses the entry's positive; writes dA[r][0..D) (row stride ldd)
__global__ __launch_bounds__(256) void infonce_anchor_grad_batched_kernel(const float* __restrict__ G, const float* __restrict__ An,
                                                                         const float* __restrict__ Pn_all, NceTable t,
                                                                         const float* __restrict__ gpos, const float* __restrict__ inv,
                                                                         int Q, long n_rows, int D, int Dp, float eps, float scale,
                                                                         float* __restrict__ dA, long ldd, const float* __restrict__ gscale = nullptr) {
  // gscale != null: G holds the UNNORMALISED weighted bank sums of arco_nce_score; row r's softmax normalisation 1 / (T Z) is applied here
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float* Pn = Pn_all + (long)t.prow[r / Q] * Dp;
  const float gp = gpos[r], iv = inv[r], gs = gscale ? gscale[r] : 1.0f;
  float dot = 0.f;
  for (int d = lane; d < Dp; d += 64) {
    const float g = G[r * Dp + d] * gs + gp * Pn[d];
    dot += g * An[r * Dp + d];
  }
  dot = wave_sum(dot);
  const bool clamped = iv >= 1.0f / eps;
  for (int d = lane; d < D; d += 64) {
    const float g = G[r * Dp + d] * gs + gp * Pn[d];
    const float v = clamped ? g * iv : iv * (g - An[r * Dp + d] * dot);
    dA[r * ldd + d] = v * scale;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the pixel-feature x memory-bank score as an MFMA GEMM with the temperature-scaled, multiplicity-weighted softmax-CE in
// its epilogue (loss_helper_3d.py:503-509) - no S[E,Q,Lp] round trip, the bank normalisation folded into the B staging.
//   arco_nce_prep     one launch: anchors and prototypes normalised (rows of An / Pn, inv norms) and, per (entry, query), the 16-bit
//                     multiplicity row M[e][q][0..Lp) of its Nn sampled negatives
//   arco_nce_score    grid (q-tile of 64, bank-row tile of 128, entry): S = An . bank^T on the fp32 matrix cores
//                     (v_mfma_f32_16x16x4_f32, K chunks of 16 double-buffered in LDS); the B staging accumulates every bank row's
//                     sum of squares, so the epilogue has 1 / max(||b_l||, eps) without a normalised copy of the bank;
//                     epilogue: Wu[q][l] = M[q][l] * exp((s - 1) / T) * inv||b_l||  (cosines are <= 1: a fixed shift, no running
//                     maximum, any summation order) and the partial row sums Zp[row][l-tile] of M * exp((s - 1) / T).
//                     The q-tile-0 workgroups also write the raw bank transposed (Bt [E][Dp][Lp]) - the operand the anchor-gradient
//                     GEMM takes - from the chunks they stage anyway.
//   arco_nce_finish   per row: Z = sum of the partials + exp((pos - 1) / T), loss = log Z - (pos - 1) / T,
//                     gpos = (exp((pos - 1) / T) / Z - 1) / T, gscale = 1 / (T Z); the step's loss sum (fixed order) in the same launch
// The anchor gradient is then Gu = Wu . bank (arco_gemm_batched on Bt) and arco_nce_anchor_grad with gscale:
// G = gscale * Gu + gpos * Pn.
// ---------------------------------------------------------------------------------------------------------------------------
struct NceIdx { const int64_t* idx_all; long idx_off, idx_stride; };

__global__ __launch_bounds__(256) void nce_prep_kernel(const float* __restrict__ A, long n_a, const float* __restrict__ P, long n_p, int D, int Dp,
                                                      float eps, float* __restrict__ An, float* __restrict__ invA, float* __restrict__ Pn,
                                                      NceTable t, NceIdx ix, int Q, int Nn, long Lp, unsigned short* __restrict__ M) {
  extern __shared__ uint32_t cnt[];                  // multiplicity blocks: Lp / 2 words
  const long nrow_blocks = (n_a + n_p + 3) / 4;
  if ((long)blockIdx.x < nrow_blocks) {              // ---- four rows per block: anchors first, then the prototypes
    const int lane = threadIdx.x & 63;
    const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_a + n_p) return;
    const bool isa = j < n_a;
    const float* s = isa ? A + j * (long)D : P + (j - n_a) * (long)D;
    float* y = isa ? An + j * (long)Dp : Pn + (j - n_a) * (long)Dp;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = s[d]; ss += v * v; }
    ss = wave_sum(ss);
    const float iv = 1.0f / fmaxf(sqrtf(ss), eps);
    if (lane == 0 && isa) invA[j] = iv;
    for (int d = lane; d < Dp; d += 64) y[d] = d < D ? s[d] * iv : 0.f;
    return;
  }
  // ---- one (entry, query) per block: multiplicities of its sampled negatives, 16-bit counters two per word
  const long r = (long)blockIdx.x - nrow_blocks;
  const int e = (int)(r / Q), q = (int)(r - (long)e * Q);
  const long L = t.len[e];
  for (long k = threadIdx.x; k < Lp / 2; k += 256) cnt[k] = 0u;
  __syncthreads();
  const int64_t* idx = ix.idx_all + (long)e * ix.idx_stride + ix.idx_off + (long)q * Nn;
  for (int i = threadIdx.x; i < Nn; i += 256) {
    long k = idx[i];
    if (k < 0) k += L;
    atomicAdd(&cnt[k >> 1], 1u << (16 * (int)(k & 1)));
  }
  __syncthreads();
  uint32_t* out = reinterpret_cast<uint32_t*>(M + r * Lp);
  for (long k = threadIdx.x; k < Lp / 2; k += 256) out[k] = cnt[k];
}

constexpr int NS_BM = 64, NS_BN = 128, NS_KC = 16, NS_LDK = NS_KC + 4;      // (NS_KC = 32 measured: 80 vs 77 us - the chunk depth is not what bounds the kernel)
constexpr int NS_Q = NS_KC / 4;                       // 16-byte pieces per staged row and chunk
constexpr int NS_RPT = 256 / NS_Q;                    // rows covered by one sweep of the 256 staging threads
constexpr int NS_NA = NS_BM / NS_RPT, NS_NB = NS_BN / NS_RPT;
__global__ __launch_bounds__(256) void nce_score_kernel(const float* __restrict__ An, int Dp, int D, NceTable t, long Lp, int Q,
                                                       const unsigned short* __restrict__ M, float inv_temp, float eps,
                                                       float* __restrict__ Wu, float* __restrict__ Zp, int n_ltiles,
                                                       float* __restrict__ Bt, const float* __restrict__ Pn_all, float* __restrict__ pos) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (NS_BM + NS_BN) * NS_LDK];
  __shared__ float ssq[NS_BN][NS_Q];
  __shared__ float invb[NS_BN];
  __shared__ float zred[2][NS_BM];
  __shared__ float pred[NS_BM][NS_Q];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, g = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;                    // 2 x 2 waves: 32 queries x 64 bank rows each
  const int e = blockIdx.z, q0 = blockIdx.x * NS_BM, l0 = blockIdx.y * NS_BN;
  const long L = t.len[e];
  const float* bank = t.bank[e];
  constexpr int BUF = (NS_BM + NS_BN) * NS_LDK;
  // staging geometry: thread -> (row, quad); NS_NA pieces of the A tile and NS_NB of the B tile per chunk
  const int qd = tid % NS_Q, ar = tid / NS_Q;
  const float* a_src[NS_NA]; bool a_ok[NS_NA];
#pragma unroll
  for (int i = 0; i < NS_NA; ++i) {
    const int r = ar + NS_RPT * i;
    a_ok[i] = q0 + r < Q;
    a_src[i] = An + ((long)e * Q + q0 + r) * Dp + 4 * qd;
  }
  const float* b_src[NS_NB]; bool b_ok[NS_NB];
#pragma unroll
  for (int i = 0; i < NS_NB; ++i) {
    const long l = l0 + ar + NS_RPT * i;
    b_ok[i] = l < L;
    b_src[i] = bank + l * (long)D + 4 * qd;
  }
  const int nchunks = (Dp + NS_KC - 1) / NS_KC;
  f32x4 ra[2][NS_NA], rb[2][NS_NB], rp[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};      // two register sets: the global loads run TWO chunks ahead of the MFMAs
  float ss[NS_NB], pdot[NS_NA];
#pragma unroll
  for (int i = 0; i < NS_NB; ++i) ss[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NS_NA; ++i) pdot[i] = 0.f;
  // the bank-tile-0 workgroups also form the positive logits pos[q] = An[q] . Pn[prow] from the anchor chunks they stage
  const bool want_pos = blockIdx.y == 0;
  const float* p_src = Pn_all + (long)t.prow[e] * Dp + 4 * qd;
  auto load_chunk = [&](int c, auto SET_) {
    constexpr int S = decltype(SET_)::value;
    const int k = c * NS_KC + 4 * qd;
#pragma unroll
    for (int i = 0; i < NS_NA; ++i) ra[S][i] = (a_ok[i] && k < Dp) ? *reinterpret_cast<const f32x4*>(a_src[i] + c * NS_KC) : f32x4{0, 0, 0, 0};
    if (want_pos) rp[S] = k < Dp ? *reinterpret_cast<const f32x4*>(p_src + c * NS_KC) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NS_NB; ++i) rb[S][i] = (b_ok[i] && k < D) ? *reinterpret_cast<const f32x4*>(b_src[i] + c * NS_KC) : f32x4{0, 0, 0, 0};
  };
  auto store_chunk = [&](float* buf, auto SET_) {
    constexpr int S = decltype(SET_)::value;
#pragma unroll
    for (int i = 0; i < NS_NA; ++i) {
      const f32x4 v = ra[S][i];
      *reinterpret_cast<f32x4*>(&buf[(ar + NS_RPT * i) * NS_LDK + 4 * qd]) = v;
      if (want_pos) pdot[i] += (v[0] * rp[S][0] + v[1] * rp[S][1]) + (v[2] * rp[S][2] + v[3] * rp[S][3]);
    }
#pragma unroll
    for (int i = 0; i < NS_NB; ++i) {
      const f32x4 v = rb[S][i];
      *reinterpret_cast<f32x4*>(&buf[(NS_BM + ar + NS_RPT * i) * NS_LDK + 4 * qd]) = v;
      ss[i] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);     // the bank row's sum of squares, this quad's share
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  auto compute = [&](const float* buf) {
    const float* As = buf; const float* Bs = buf + NS_BM * NS_LDK;
#pragma unroll
    for (int kk = 0; kk < NS_KC / 16; ++kk) {
      f32x4 af[2], bf[4];
#pragma unroll
      for (int at = 0; at < 2; ++at) af[at] = *reinterpret_cast<const f32x4*>(&As[((wm * 2 + at) * 16 + li) * NS_LDK + kk * 16 + 4 * g]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bf[ct] = *reinterpret_cast<const f32x4*>(&Bs[((wn * 4 + ct) * 16 + li) * NS_LDK + kk * 16 + 4 * g]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[at][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[at][j], bf[ct][j], acc[at][ct], 0, 0, 0);
    }
  };
  // The transposed raw bank Bt[e][k][l0 .. l0 + 127] (the anchor-gradient GEMM's operand) is written from the staged chunks, the k quads of a
  // chunk dealt over the q-tile workgroups of this bank tile (quad j by workgroup j % gridDim.x): a thread reads the four k of one
  // bank row with one conflict-free ds_read_b128 and stores them to four Bt rows - 64 consecutive l per wave instruction, 256-byte runs.
  // (A first version let the q-tile-0 workgroups write everything with 4-byte stores in 32-byte runs: 45 of the kernel's 107 us.)
  const bool write_bt = Bt != nullptr;
  auto emit_bt = [&](const float* buf, int c) {
    if (tid >= NS_BN) return;
    const float* Bs = buf + NS_BM * NS_LDK;
#pragma unroll
    for (int j = 0; j < NS_Q; ++j) {
      if (j % (int)gridDim.x != (int)blockIdx.x) continue;
      const int k = c * NS_KC + 4 * j;
      if (k >= Dp) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[tid * NS_LDK + 4 * j]);
      if (l0 + tid < Lp) {
        float* dst = Bt + ((long)e * Dp + k) * Lp + l0 + tid;
        dst[0] = v[0]; dst[Lp] = v[1]; dst[2 * Lp] = v[2]; dst[3 * Lp] = v[3];
      }
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  load_chunk(0, S0{});
  if (nchunks > 1) load_chunk(1, S1{});
  store_chunk(smem, S0{});
  __syncthreads();
  // chunk c is computed from LDS buffer c & 1 while chunk c + 1 (in registers since the previous trip) is written to the other buffer and
  // the loads of chunk c + 2 are in flight; two chunks per trip keep the register sets compile-time
  auto trip = [&](int c, auto SETN_) {           // SETN: register set of chunk c + 1 (= (c + 1) & 1); chunk c + 2 reuses chunk c's set
    constexpr int SN = decltype(SETN_)::value;
    float* cur = smem + (c & 1) * BUF;
    float* nxt = smem + ((c + 1) & 1) * BUF;
    if (c + 2 < nchunks) load_chunk(c + 2, std::integral_constant<int, SN ^ 1>{});
    compute(cur);
    if (write_bt) emit_bt(cur, c);
    if (c + 1 < nchunks) store_chunk(nxt, SETN_);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    trip(c, S1{});
    if (c + 1 < nchunks) trip(c + 1, S0{});
  }
  // bank-row inverse norms / positive logits: the quad partials of a row, summed in a fixed order
#pragma unroll
  for (int i = 0; i < NS_NB; ++i) ssq[ar + NS_RPT * i][qd] = ss[i];
  if (want_pos) {
#pragma unroll
    for (int i = 0; i < NS_NA; ++i) pred[ar + NS_RPT * i][qd] = pdot[i];
  }
  __syncthreads();
  if (tid < NS_BN) {
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NS_Q; ++j) s2 += ssq[tid][j];
    invb[tid] = 1.0f / fmaxf(sqrtf(s2), eps);
  }
  if (want_pos && tid < NS_BM && q0 + tid < Q) {
    float pv = 0.f;
#pragma unroll
    for (int j = 0; j < NS_Q; ++j) pv += pred[tid][j];
    pos[(long)e * Q + q0 + tid] = pv;
  }
  __syncthreads();
  // epilogue: lane (li, g) of tile (at, ct) holds S[q = 4g + r][l = li]
  float zrow[2][4];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) zrow[at][r] = 0.f;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int ll = (wn * 4 + ct) * 16 + li;
    const long l = l0 + ll;
    const float ib = invb[ll];
#pragma unroll
    for (int at = 0; at < 2; ++at)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + (wm * 2 + at) * 16 + 4 * g + r;
        if (q < Q && l < Lp) {
          const long row = (long)e * Q + q;
          const unsigned mk = l < L ? (unsigned)M[row * Lp + l] : 0u;
          float w = 0.f;
          if (mk) {
            const float ex = (float)mk * expf((acc[at][ct][r] * ib - 1.0f) * inv_temp);
            zrow[at][r] += ex;
            w = ex * ib;
          }
          Wu[row * Lp + l] = w;
        }
      }
  }
  // partial row sums of this 128-row bank tile: 16 lanes (li) of a row group, then the two wave columns, fixed order
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = zrow[at][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
      if (li == 0) zred[wn][(wm * 2 + at) * 16 + 4 * g + r] = v;
    }
  __syncthreads();
  if (tid < NS_BM && q0 + tid < Q) Zp[((long)e * Q + q0 + tid) * n_ltiles + blockIdx.y] = zred[0][tid] + zred[1][tid];
}

__global__ __launch_bounds__(1024) void nce_finish_kernel(const float* __restrict__ pos, long n_rows, const float* __restrict__ Zp, int n_ltiles,
                                                         float inv_temp, float scale, float* __restrict__ gpos, float* __restrict__ gscale,
                                                         float* __restrict__ loss_q, float* __restrict__ loss_sum) {
  // ONE block, one thread per row (n_rows = E * Q, strided beyond 1024), then the loss sum in a fixed order
  __shared__ double part[1024];
  double acc = 0.0;
  for (long r = threadIdx.x; r < n_rows; r += 1024) {
    double z = 0.0;
    const float* zr = Zp + r * n_ltiles;
    int k = 0;
    for (; k + 3 < n_ltiles && ((reinterpret_cast<uintptr_t>(zr) & 15) == 0); k += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(zr + k);
      z += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    }
    for (; k < n_ltiles; ++k) z += (double)zr[k];
    const float dp = pos[r];
    const float epos = expf((dp - 1.0f) * inv_temp);
    const float tot = (float)(z + (double)epos);
    const float lq = logf(tot) - (dp - 1.0f) * inv_temp;
    loss_q[r] = lq;
    gpos[r] = (epos / tot - 1.0f) * inv_temp;
    gscale[r] = inv_temp / tot;
    acc += (double)lq;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_sum[0] = (float)(part[0] * (double)scale);
}

// out[e*Q + q] = lists[k_e][ idx_e[q] ]  (global pixel id of every sampled anchor, entries back to back)
struct PixTable { int k[ARCO_MAXC]; };
__global__ void anchor_pix_kernel(const int32_t* __restrict__ lists, long n_pix, PixTable t, const int64_t* __restrict__ idx_all,
                                  long idx_stride, int Q, int E, int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q * E) return;
  const int e = i / Q, q = i - e * Q;
  out[i] = (int64_t)lists[(long)t.k[e] * n_pix + idx_all[(long)e * idx_stride + q]];
}

// anchor gradient through the normalisation:  dAhat = G + gpos*Pn ;
// dA = inv*(dAhat - Ahat*(Ahat.dAhat)) if ||A||>eps else dAhat/eps ; scaled by `scale`
__global__ __launch_bounds__(256) void infonce_anchor_grad_kernel(const float* __restrict__ G, const float* __restrict__ An,
                                                                 const float* __restrict__ Pn_, long ldp,
                                                                 const float* __restrict__ gpos,
                                                                 const float* __restrict__ inv, int Q, int D, float eps,
                                                                 float scale, float* __restrict__ dA) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const float* Pn = Pn_ + (long)q * ldp;
  const float gp = gpos[q], iv = inv[q];
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float g = G[(long)q * D + d] + gp * Pn[d];
    dot += g * An[(long)q * D + d];
  }
  dot = wave_sum(dot);
  const bool clamped = iv >= 1.0f / eps;   // ||A|| <= eps : y = A/eps, no norm term
  for (int d = lane; d < D; d += 64) {
    const float g = G[(long)q * D + d] + gp * Pn[d];
    const float v = clamped ? g * iv : iv * (g - An[(long)q * D + d] * dot);
    dA[(long)q * D + d] = v * scale;
  }
}

// dst[rows[j]] += alpha * src[j]   (anchor gradient scatter; duplicates add)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, long lds_, int D,
                                                              const int32_t* __restrict__ list,
                                                              const int64_t* __restrict__ idx, long n,
                                                              const float* __restrict__ alpha_dev, float alpha,
                                                              float* __restrict__ dst, long ldd) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  long r = idx[j];
  if (list) r = list[r];
  const float a = alpha_dev ? alpha * alpha_dev[0] : alpha;
  for (int d = lane; d < D; d += 64) atomicAdd(&dst[r * ldd + d], a * src[j * lds_ + d]);
}

__global__ void sum_scale_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out, int accumulate) {
  // single block deterministic sum
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * (double)scale);
    out[0] = accumulate ? out[0] + v : v;
  }
}

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int arco_mask_codes(const int64_t* lab_l, const int64_t* lab_u, const float* prob_l, const float* prob_u,
                    const float* low_mask, const float* high_mask, int n_l_img, int n_u_img, int C, long P,
                    float delta_p, float delta_n, int low_rank, int high_rank, uint64_t* codes,
                    uint32_t* block_counts, uint32_t* block_offsets, int64_t* totals, void* stream) {
  ARCO_CHECK_ARG(C >= 1 && C <= ARCO_MAXC && P > 0 && n_l_img >= 0 && n_u_img >= 0);
  const long n_pix = (long)(n_l_img + n_u_img) * P;
  ARCO_CHECK_ARG(n_pix > 0 && n_pix < (1l << 31));
  const int nblocks = (int)((n_pix + 255) / 256);
  hipLaunchKernelGGL(mask_codes_kernel, dim3(nblocks), dim3(256), 0, as_stream(stream), lab_l, lab_u, prob_l, prob_u,
                     low_mask, high_mask, n_l_img, C, P, n_pix, delta_p, delta_n, low_rank, high_rank, codes,
                     block_counts, nblocks);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(3 * C), dim3(256), 0, as_stream(stream), block_counts, nblocks,
                     block_offsets, totals);
  return arco_launch_status();
}

int arco_compact_rows(const uint64_t* codes, long n_pix, int C, const uint32_t* block_offsets, int32_t* lists,
                      void* stream) {
  ARCO_CHECK_ARG(C >= 1 && C <= ARCO_MAXC && n_pix > 0);
  const int nblocks = (int)((n_pix + 255) / 256);
  hipLaunchKernelGGL(compact_rows_kernel, dim3(nblocks), dim3(256), 0, as_stream(stream), codes, n_pix, C,
                     block_offsets, nblocks, lists);
  return arco_launch_status();
}

// workspace: partial must hold arco_proto_ws_floats(...) floats
long arco_proto_ws_floats(long n_pix, int C, int D) {
  long grid = (n_pix + 255) / 256;                      // masked_row_sum slabs
  if (grid > 2048) grid = 2048;
  long g2 = (n_pix + 63) / 64;                          // weighted_row_sum slabs
  if (g2 > 1024) g2 = 1024;
  if (g2 > grid) grid = g2;
  if (grid < 1) grid = 1;
  const int nc = C < 8 ? C : 8;
  return grid * nc * D;
}

int arco_masked_proto(const float* T, long ldt, const uint64_t* codes, long n_pix, int C, int D,
                      const int64_t* totals, float* partial, float* proto, void* stream) {
  ARCO_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 512 && (ldt & 3) == 0 && C <= ARCO_MAXC);
  long grid = (n_pix + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  long rpb = (n_pix + grid - 1) / grid;
  int lpr = 1;
  while (lpr * 4 < D && lpr < 64) lpr <<= 1;
  const int ndi = (D + lpr * 4 - 1) / (lpr * 4);
  for (int c0 = 0; c0 < C; c0 += 8) {
    const int nc = (C - c0) < 8 ? (C - c0) : 8;
    const size_t sh = (size_t)4 * 8 * D * sizeof(float);
    if (ndi == 1)
      hipLaunchKernelGGL(masked_row_sum_kernel<1>, dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, codes, n_pix,
                         D, c0, nc, lpr, rpb, partial);
    else if (ndi == 2)
      hipLaunchKernelGGL(masked_row_sum_kernel<2>, dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, codes, n_pix,
                         D, c0, nc, lpr, rpb, partial);
    else
      hipLaunchKernelGGL(masked_row_sum_kernel<4>, dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, codes, n_pix,
                         D, c0, nc, lpr, rpb, partial);
    hipLaunchKernelGGL(proto_finalize_kernel, dim3((nc * D + 255) / 256), dim3(256), 0, as_stream(stream), partial,
                       (int)grid, nc, D, c0, totals, proto);
  }
  return arco_launch_status();
}

int arco_lv_weights(const uint64_t* codes, long n_pix, int C, int Cp, float* W, void* stream) {
  ARCO_CHECK_ARG(C <= ARCO_MAXC && Cp >= C);
  const long tot = n_pix * Cp;
  hipLaunchKernelGGL(lv_weights_kernel, dim3((tot + 255) / 256), dim3(256), 0, as_stream(stream), codes, n_pix, C, Cp, W);
  return arco_launch_status();
}

// out[c][0..D) = sum_rows Wt[row][c] * T[row][0..D)  (divided by totals[c] when totals != NULL);
// partial: arco_proto_ws_floats(n_rows, C, D) floats
extern "C++" {
template <typename TS>
static int weighted_row_sum_impl(const TS* T, long ldt, const float* Wt, long ldw, long n_rows, int C, int D,
                                 const int64_t* totals, float* partial, float* out, long ldo, void* stream) {
  ARCO_CHECK_ARG(D > 0 && (D & 3) == 0 && D <= 512 && (ldt & 3) == 0 && C <= ARCO_MAXC);
  long grid = (n_rows + 63) / 64;                      // HBM-bound row sweep: >= 4 blocks per CU in flight
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  long rpb = (n_rows + grid - 1) / grid;
  int lpr = 1;
  while (lpr * 4 < D && lpr < 64) lpr <<= 1;
  const int ndi = (D + lpr * 4 - 1) / (lpr * 4);
  for (int c0 = 0; c0 < C; c0 += 8) {
    const int nc = (C - c0) < 8 ? (C - c0) : 8;
    const size_t sh = (size_t)4 * 8 * D * sizeof(float);
    if (ndi == 1)
      hipLaunchKernelGGL((weighted_row_sum_kernel<1, TS>), dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, Wt, ldw, n_rows, D, c0, nc, lpr, rpb, partial);
    else if (ndi == 2)
      hipLaunchKernelGGL((weighted_row_sum_kernel<2, TS>), dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, Wt, ldw, n_rows, D, c0, nc, lpr, rpb, partial);
    else
      hipLaunchKernelGGL((weighted_row_sum_kernel<4, TS>), dim3(grid), dim3(256), sh, as_stream(stream), T, ldt, Wt, ldw, n_rows, D, c0, nc, lpr, rpb, partial);
    hipLaunchKernelGGL(row_sum_finalize_kernel, dim3((nc * D + 63) / 64), dim3(64, 16), 0, as_stream(stream), partial,
                       (int)grid, nc, D, c0, totals, out, ldo);
  }
  return arco_launch_status();
}
template <typename TS>
static int gather_rows_impl(const TS* src, long ld_src, int D, const int32_t* list, const int64_t* idx64,
                            const int32_t* idx32, long first, long n, float* out, long ld_out, void* stream) {
  ARCO_CHECK_ARG(D > 0 && n >= 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(gather_rows_kernel<TS>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), src, ld_src, D, list,
                     idx64, idx32, first, n, out, ld_out);
  return arco_launch_status();
}
}  // extern "C++"
int arco_weighted_row_sum(const float* T, long ldt, const float* Wt, long ldw, long n_rows, int C, int D,
                          const int64_t* totals, float* partial, float* out, long ldo, void* stream) {
  return weighted_row_sum_impl<float>(T, ldt, Wt, ldw, n_rows, C, D, totals, partial, out, ldo, stream);
}
// ... of an f16 row matrix (f16 activation storage: the full-resolution V-Net feature maps stay f16, sums are fp32)
int arco_weighted_row_sum_h(const void* T, long ldt, const float* Wt, long ldw, long n_rows, int C, int D,
                            const int64_t* totals, float* partial, float* out, long ldo, void* stream) {
  return weighted_row_sum_impl<_Float16>(reinterpret_cast<const _Float16*>(T), ldt, Wt, ldw, n_rows, C, D, totals, partial, out, ldo, stream);
}

int arco_gather_rows(const float* src, long ld_src, int D, const int32_t* list, const int64_t* idx64,
                     const int32_t* idx32, long first, long n, float* out, long ld_out, void* stream) {
  return gather_rows_impl<float>(src, ld_src, D, list, idx64, idx32, first, n, out, ld_out, stream);
}
// ... from an f16 row matrix into fp32 rows
int arco_gather_rows_h(const void* src, long ld_src, int D, const int32_t* list, const int64_t* idx64,
                       const int32_t* idx32, long first, long n, float* out, long ld_out, void* stream) {
  return gather_rows_impl<_Float16>(reinterpret_cast<const _Float16*>(src), ld_src, D, list, idx64, idx32, first, n, out, ld_out, stream);
}

int arco_bank_append(const float* old, long len_old, const float* keys, long n, long queue_size, int D, float* out,
                     void* stream) {
  ARCO_CHECK_ARG(len_old >= 0 && n >= 0 && queue_size > 0 && D > 0);
  const long tot = len_old + n;
  const long out_len = tot >= queue_size ? queue_size : tot;
  const long drop = tot - out_len;
  if (out_len == 0) return ARCO_OK;
  const long work = out_len * D;
  hipLaunchKernelGGL(bank_append_kernel, dim3((work + 255) / 256), dim3(256), 0, as_stream(stream), old, len_old, keys,
                     n, drop, out_len, D, out);
  return arco_launch_status();
}

int arco_normalize_rows(const float* x, long ldx, long n, int D, float eps, float* y, long ldy, float* yt, long ldyt,
                        float* inv, void* stream) {
  ARCO_CHECK_ARG(n >= 0 && D > 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), x, ldx, n, D, eps, y,
                     ldy, yt, ldyt, inv);
  return arco_launch_status();
}

int arco_neg_multiplicity(const int64_t* idx, int Q, int Nn, long L, long ld, uint32_t* M, void* stream) {
  ARCO_CHECK_ARG(Q > 0 && Nn > 0 && L > 0 && ld >= L);
  (void)hipMemsetAsync(M, 0, (size_t)Q * ld * sizeof(uint32_t), as_stream(stream));
  const long n = (long)Q * Nn;
  hipLaunchKernelGGL(neg_multiplicity_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), idx, Q, Nn, L, ld, M);
  return arco_launch_status();
}

// S, M, W share the row stride `ld` (>= L); An/Pn rows have D (padded) floats; ldp = 0 for a shared positive
int arco_infonce_fwd(const float* S, long ld, const uint32_t* M, long L, const float* An, const float* Pn, long ldp,
                     int Q, int D, float temp, float* W, float* gpos, float* loss_q, void* stream) {
  ARCO_CHECK_ARG(Q > 0 && L > 0 && temp > 0.f && ld >= L);
  hipLaunchKernelGGL(infonce_fwd_kernel, dim3(Q), dim3(256), 0, as_stream(stream), S, ld, M, L, An, Pn, ldp, D,
                     1.0f / temp, W, gpos, loss_q);
  return arco_launch_status();
}

int arco_infonce_anchor_grad(const float* G, const float* An, const float* Pn, long ldp, const float* gpos,
                             const float* inv, int Q, int D, float eps, float scale, float* dA, void* stream) {
  hipLaunchKernelGGL(infonce_anchor_grad_kernel, dim3((Q + 3) / 4), dim3(256), 0, as_stream(stream), G, An, Pn, ldp, gpos,
                     inv, Q, D, eps, scale, dA);
  return arco_launch_status();
}

int arco_scatter_add_rows(const float* src, long ld_src, int D, const int32_t* list, const int64_t* idx, long n,
                          const float* alpha_dev, float alpha, float* dst, long ld_dst, void* stream) {
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), src, ld_src, D, list, idx, n,
                     alpha_dev, alpha, dst, ld_dst);
  return arco_launch_status();
}

// ---- grouped InfoNCE (all classes of a step per launch); host arrays: banks[E] device pointers, lens[E], prow[E] -------------
static int fill_table(NceTable& t, const void* const* banks, const int* lens, const int* prow, int E) {
  if (E < 1 || E > ARCO_MAXC) return ARCO_ERR_ARG;
  for (int e = 0; e < E; ++e) {
    t.bank[e] = banks ? reinterpret_cast<const float*>(banks[e]) : nullptr;
    t.len[e] = lens ? lens[e] : 0; t.prow[e] = prow ? prow[e] : 0;
  }
  return ARCO_OK;
}
int arco_normalize_rows_pad(const float* x, long ldx, long n, int D, int Dp, float eps, float* y, long ldy, float* inv,
                            void* stream) {
  ARCO_CHECK_ARG(n >= 0 && D > 0 && Dp >= D && y);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(normalize_rows_pad_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), x, ldx, n, D, Dp, eps, y, ldy, inv);
  return arco_launch_status();
}
// Bn [E][Lp][Dp] (+ Bt [E][Dp][Lp], nullable): normalised rows of every entry's bank, zero padded (loss_helper_3d.py:503)
int arco_nce_normalize_banks(const void* const* banks, const int* lens, int E, int D, int Dp, long Lp, float eps, float* Bn,
                             float* Bt, void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, banks, lens, nullptr, E) == ARCO_OK && D > 0 && Dp >= D && Lp > 0 && Bn);
  for (int e = 0; e < E; ++e) ARCO_CHECK_ARG(t.bank[e] && t.len[e] > 0 && t.len[e] <= Lp);
  hipLaunchKernelGGL(normalize_banks_kernel, dim3((unsigned)((Lp + 15) / 16), (unsigned)E), dim3(256), (size_t)16 * (Dp + 1) * sizeof(float),
                     as_stream(stream), t, D, Dp, Lp, eps, Bn, Bt);
  return arco_launch_status();
}
// S / W: [E][Q][ld]; An [E*Q][Dp]; Pn_all [*][Dp] (row prow[e] = the entry's positive); idx_all: int64 indices, the Q*Nn
// negatives of entry e start at idx_all + e*idx_stride + idx_off; gpos / loss_q [E*Q].  loss_helper_3d.py:503-509.
long arco_nce_max_len() { return 2l * ((160 * 1024 - 512) / 4); }      // longest bank arco_nce_fused accepts: 16-bit counters in <= 160 KB of LDS
int arco_nce_fused(const float* S, long ld, const int* lens, const int* prow, int E, const int64_t* idx_all, long idx_off,
                   long idx_stride, int Q, int Nn, const float* An, const float* Pn_all, int Dp, float temp, float* W,
                   float* gpos, float* loss_q, void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, nullptr, lens, prow, E) == ARCO_OK && Q > 0 && Nn > 0 && Nn < 65536 && temp > 0.f && S && An && Pn_all);
  long Lmax = 0;
  for (int e = 0; e < E; ++e) { ARCO_CHECK_ARG(t.len[e] > 0 && t.len[e] <= ld); if (t.len[e] > Lmax) Lmax = t.len[e]; }
  const size_t sh = (size_t)((Lmax + 1) / 2) * sizeof(uint32_t);
  ARCO_CHECK_ARG(sh <= (size_t)(160 * 1024 - 512));
  static unsigned long long attr_set = 0;
  if (sh > 48 * 1024 && arco_first_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(infonce_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  }
  hipLaunchKernelGGL(infonce_fused_kernel, dim3((unsigned)Q, (unsigned)E), dim3(256), sh, as_stream(stream), S, ld, t, idx_all,
                     idx_off, idx_stride, Q, Nn, An, Pn_all, Dp, 1.0f / temp, W, gpos, loss_q);
  return arco_launch_status();
}
int arco_nce_anchor_grad(const float* G, const float* An, const float* Pn_all, const int* prow, int E, const float* gpos,
                         const float* inv, int Q, int D, int Dp, float eps, float scale, float* dA, long ld_dA, void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, nullptr, nullptr, prow, E) == ARCO_OK && Q > 0 && D > 0 && Dp >= D);
  const long n = (long)E * Q;
  hipLaunchKernelGGL(infonce_anchor_grad_batched_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), G, An, Pn_all,
                     t, gpos, inv, Q, n, D, Dp, eps, scale, dA, ld_dA);
  return arco_launch_status();
}
// ---- round 6: score GEMM with the softmax-CE in its epilogue (see nce_score_kernel) ---------------------------------------------
// A [n_a = E*Q][D] sampled anchors, P [n_p][D] prototypes -> An [n_a][Dp], invA [n_a], Pn [n_p][Dp]; M [E*Q][Lp] uint16 multiplicities
// of each query's Nn sampled negatives (indices as in arco_nce_fused).  Lp % 16 == 0, Lp <= arco_nce_max_len().
int arco_nce_prep(const float* A, long n_a, const float* P, long n_p, int D, int Dp, float eps, float* An, float* invA, float* Pn,
                  const int* lens, int E, const int64_t* idx_all, long idx_off, long idx_stride, int Q, int Nn, long Lp, void* M,
                  void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, nullptr, lens, nullptr, E) == ARCO_OK && A && P && An && invA && Pn && M && idx_all && n_a == (long)E * Q &&
                 n_p > 0 && D > 0 && Dp >= D && Q > 0 && Nn > 0 && Nn < 65536 && Lp > 0 && (Lp & 15) == 0 && Lp <= arco_nce_max_len());
  for (int e = 0; e < E; ++e) ARCO_CHECK_ARG(t.len[e] > 0 && t.len[e] <= Lp);
  const size_t sh = (size_t)Lp * 2;
  static unsigned long long attr_set = 0;
  if (sh > 48 * 1024 && arco_first_on_device(attr_set))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nce_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  const long blocks = (n_a + n_p + 3) / 4 + n_a;
  NceIdx ix{idx_all, idx_off, idx_stride};
  hipLaunchKernelGGL(nce_prep_kernel, dim3((unsigned)blocks), dim3(256), sh, as_stream(stream), A, n_a, P, n_p, D, Dp, eps, An, invA, Pn, t, ix,
                     Q, Nn, Lp, reinterpret_cast<unsigned short*>(M));
  return arco_launch_status();
}
long arco_nce_score_ltiles(long Lp) { return (Lp + NS_BN - 1) / NS_BN; }
// Wu [E][Q][Lp], Zp [E*Q][arco_nce_score_ltiles(Lp)], pos [E*Q] (= An . Pn[prow[e]]), Bt [E][Dp][Lp] (nullable: no gradient wanted);
// banks: raw [len][D] rows, D % 4 == 0
int arco_nce_score(const float* An, int Dp, int D, const void* const* banks, const int* lens, const int* prow, int E, long Lp, int Q,
                   const void* M, const float* Pn_all, float temp, float eps, float* Wu, float* Zp, float* pos, float* Bt, void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, banks, lens, prow, E) == ARCO_OK && An && M && Wu && Zp && Pn_all && pos && Q > 0 && D > 0 && (D & 3) == 0 && Dp >= D &&
                 (Dp & 15) == 0 && Lp > 0 && (Lp & 15) == 0 && temp > 0.f);
  for (int e = 0; e < E; ++e) ARCO_CHECK_ARG(t.bank[e] && t.len[e] > 0 && t.len[e] <= Lp && (reinterpret_cast<uintptr_t>(t.bank[e]) & 15) == 0);
  const int n_lt = (int)arco_nce_score_ltiles(Lp);
  hipLaunchKernelGGL(nce_score_kernel, dim3((unsigned)((Q + NS_BM - 1) / NS_BM), (unsigned)n_lt, (unsigned)E), dim3(256), 0, as_stream(stream),
                     An, Dp, D, t, Lp, Q, reinterpret_cast<const unsigned short*>(M), 1.0f / temp, eps, Wu, Zp, n_lt, Bt, Pn_all, pos);
  return arco_launch_status();
}
// per row (pos, Zp from arco_nce_score): loss_q, gpos, gscale; loss_sum[0] = scale * sum of loss_q (fixed order)
int arco_nce_finish(const float* pos, long n_rows, const float* Zp, long Lp, float temp, float scale, float* gpos, float* gscale,
                    float* loss_q, float* loss_sum, void* stream) {
  ARCO_CHECK_ARG(pos && Zp && gpos && gscale && loss_q && loss_sum && n_rows > 0 && temp > 0.f);
  hipLaunchKernelGGL(nce_finish_kernel, dim3(1), dim3(1024), 0, as_stream(stream), pos, n_rows, Zp,
                     (int)arco_nce_score_ltiles(Lp), 1.0f / temp, scale, gpos, gscale, loss_q, loss_sum);
  return arco_launch_status();
}
// arco_nce_anchor_grad on the unnormalised weighted bank sums of arco_nce_score: G_row = gscale[row] * Gu_row + gpos[row] * Pn
int arco_nce_anchor_grad_scaled(const float* Gu, const float* An, const float* Pn_all, const int* prow, int E, const float* gpos,
                                const float* inv, const float* gscale, int Q, int D, int Dp, float eps, float scale, float* dA, long ld_dA,
                                void* stream) {
  NceTable t;
  ARCO_CHECK_ARG(fill_table(t, nullptr, nullptr, prow, E) == ARCO_OK && Q > 0 && D > 0 && Dp >= D && gscale);
  const long n = (long)E * Q;
  hipLaunchKernelGGL(infonce_anchor_grad_batched_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), Gu, An, Pn_all,
                     t, gpos, inv, Q, n, D, Dp, eps, scale, dA, ld_dA, gscale);
  return arco_launch_status();
}
// out[e*Q + q] = lists[k[e]][idx_all[e*idx_stride + q]]: pixel ids of the sampled anchors (loss_helper_3d.py:455-457)
int arco_anchor_pix(const int32_t* lists, long n_pix, const int* k, int E, const int64_t* idx_all, long idx_stride, int Q,
                    int64_t* out, void* stream) {
  ARCO_CHECK_ARG(lists && k && idx_all && out && E >= 1 && E <= ARCO_MAXC && Q > 0);
  PixTable t;
  for (int e = 0; e < E; ++e) t.k[e] = k[e];
  hipLaunchKernelGGL(anchor_pix_kernel, dim3((unsigned)((Q * E + 255) / 256)), dim3(256), 0, as_stream(stream), lists, n_pix, t, idx_all,
                     idx_stride, Q, E, out);
  return arco_launch_status();
}

int arco_sum_scale(const float* x, int n, float scale, float* out, int accumulate, void* stream) {
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, n, scale, out, accumulate);
  return arco_launch_status();
}

}  // extern "C"
