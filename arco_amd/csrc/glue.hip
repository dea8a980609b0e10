// Step glue of the trainer (train_arco_2d.py:284-286,342-393,492-498):
// class softmax / max / argmax / entropy, one-hot labels, exact np.percentile (linear) of the
// entropy via radix select, and the low/high entropy masks.  HBM-bound, one pixel per lane.
#include <cstddef>
#include <cstring>
#include "common.h"

#define GL_MAXC 32

// logits rows [M][ld] (channels-last) -> any of: prob planes [b][C][P], max prob, argmax, entropy
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ X, long ld, long M, int C, long P,
                                                          float* __restrict__ prob_planes, float* __restrict__ maxp,
                                                          int64_t* __restrict__ amax, float* __restrict__ entropy) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    const float* x = X + r * ld;
    float v[GL_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < GL_MAXC; ++c) if (c < C) { v[c] = x[c]; mx = fmaxf(mx, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < GL_MAXC; ++c) if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
    float best = -1.f; int bi = 0; float ent = 0.f;
    const long n = r / P, sp = r - n * P;
#pragma unroll
    for (int c = 0; c < GL_MAXC; ++c) if (c < C) {
      const float p = v[c] / s;
      if (p > best) { best = p; bi = c; }
      ent += p * logf(p + 1e-10f);
      if (prob_planes) prob_planes[(n * C + c) * P + sp] = p;
    }
    if (maxp) maxp[r] = best;
    if (amax) amax[r] = bi;
    if (entropy) entropy[r] = -ent;
  }
}

// labels [M] int64 -> one-hot int64 planes [b][C][P]; negatives clamp to class 0 (train_arco_2d.py:492-498)
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* __restrict__ lab, long M, int C, long P,
                                                    int64_t* __restrict__ out) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    int64_t l = lab[r]; if (l < 0) l = 0;
    const long n = r / P, sp = r - n * P;
    for (int c = 0; c < C; ++c) out[(n * C + c) * P + sp] = (c == l) ? 1 : 0;
  }
}

// ---- exact order statistics by 4-pass radix select (R ranks at once) --------------------
struct SelState {          // device-resident
  unsigned long long n_valid;
  unsigned long long k[4];     // remaining rank inside the current prefix bucket
  unsigned int prefix[4];
  double gamma[2];             // interpolation weights of the two percentiles
  double thr[2];               // results
  unsigned int hist[4][256];
};
__device__ __forceinline__ unsigned int fkey(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned int k) {
  const unsigned int u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}
__global__ __launch_bounds__(256) void sel_count_kernel(const int64_t* __restrict__ lab, long n, SelState* st) {
  unsigned int c = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) c += lab[i] >= 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned int sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const unsigned long long t = (unsigned long long)sh[0] + sh[1] + sh[2] + sh[3]; if (t) atomicAdd(&st->n_valid, t); }
}
// np.percentile(method='linear'): virtual index (n-1)*q/100 in float64, neighbours floor / floor+1
__global__ void sel_init_kernel(SelState* st, double q_lo, double q_hi) {
  const double nm1 = (double)(st->n_valid - 1);
  const double qs[2] = {q_lo, q_hi};
  for (int j = 0; j < 2; ++j) {
    const double vi = nm1 * (qs[j] / 100.0);
    double lo = floor(vi);
    if (lo > nm1) lo = nm1;
    double hi = lo + 1.0; if (hi > nm1) hi = nm1;
    st->gamma[j] = vi - lo;
    st->k[2 * j] = (unsigned long long)lo; st->k[2 * j + 1] = (unsigned long long)hi;
    st->prefix[2 * j] = 0; st->prefix[2 * j + 1] = 0;
  }
  for (int r = 0; r < 4; ++r) for (int d = 0; d < 256; ++d) st->hist[r][d] = 0;
}
__global__ __launch_bounds__(256) void sel_hist_kernel(const float* __restrict__ val, const int64_t* __restrict__ lab,
                                                      long n, int pass, SelState* st) {
  __shared__ unsigned int h[4][256];
  for (int i = threadIdx.x; i < 1024; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  unsigned int pre[4];
  for (int r = 0; r < 4; ++r) pre[r] = st->prefix[r];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (lab[i] < 0) continue;
    const unsigned int key = fkey(val[i]);
    const unsigned int d = (key >> shift) & 255u;
    const unsigned int hi = pass == 0 ? 0u : (key >> (shift + 8));
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (pass == 0 || hi == (pre[r] >> (shift + 8))) atomicAdd(&h[r][d], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const unsigned int c = (&h[0][0])[i];
    if (c) atomicAdd(&(&st->hist[0][0])[i], c);
  }
}
// one wave per order statistic: lane l owns bins 4l .. 4l+3; inclusive wave scan of the lane sums finds the lane, then
// the bin, whose cumulative count first exceeds the remaining rank k (the serial 256-bin walk took 20 us per pass)
__global__ __launch_bounds__(256) void sel_pick_kernel(int pass, SelState* st) {
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int shift = 24 - 8 * pass;
  const unsigned long long k = st->k[r];
  unsigned long long c[4], tot = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { c[e] = st->hist[r][4 * lane + e]; tot += c[e]; }
  unsigned long long incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const unsigned long long excl = incl - tot;
  const bool mine = incl > k && excl <= k;                      // exactly one lane, unless k >= the grand total
  const unsigned long long any = __ballot(mine);
  __syncthreads();                                              // every wave has read its histogram row
  if (any ? mine : lane == 63) {
    unsigned long long run = excl; int d = 4 * lane;
    if (any) { for (int e = 0; e < 4; ++e) { if (run + c[e] > k) { d = 4 * lane + e; break; } run += c[e]; } }
    else { d = 255; run = incl; }                                // k beyond the total (no valid pixel): bin 255, as the serial walk
    st->k[r] = k - run;
    st->prefix[r] |= ((unsigned int)d) << shift;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) st->hist[r][4 * lane + e] = 0;
  __syncthreads();
  if (pass == 3 && threadIdx.x < 2) {      // numpy _lerp in float64
    const int j = threadIdx.x;
    __threadfence_block();
    const double a = (double)fkey_inv(st->prefix[2 * j]), b = (double)fkey_inv(st->prefix[2 * j + 1]);
    const double t = st->gamma[j], diff = b - a;
    st->thr[j] = t >= 0.5 ? b - diff * (1.0 - t) : a + diff * t;
  }
}
// low/high masks [B,1,P] (train_arco_2d.py:362-393): labeled part = (label_l >= 0); unlabeled part =
// (entropy <= float32(low_thr)) & valid   /   (entropy >= float32(high_thr)) & valid
__global__ __launch_bounds__(256) void entropy_masks_kernel(const float* __restrict__ ent, const int64_t* __restrict__ lab_l,
                                                           const int64_t* __restrict__ lab_u, long n_l, long n_u,
                                                           const SelState* st, float* __restrict__ low,
                                                           float* __restrict__ high) {
  const float tl = (float)st->thr[0], th = (float)st->thr[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_l + n_u; i += (long)gridDim.x * 256) {
    if (i < n_l) { const float v = lab_l[i] >= 0 ? 1.f : 0.f; low[i] = v; high[i] = v; }
    else {
      const long j = i - n_l; const bool ok = lab_u[j] >= 0; const float e = ent[j];
      low[i] = (ok && e <= tl) ? 1.f : 0.f; high[i] = (ok && e >= th) ? 1.f : 0.f;
    }
  }
}

// ---- segmentation loss terms on the logits: supervised CE + Dice
// (train_arco_2d.py:336-339, utils/losses.py:173-209) and the confidence-weighted unsupervised CE
// (train_arco_2d.py:482-489).  Forward = per-block partial sums (deterministic finalize in fp64);
// backward = one elementwise pass producing d loss / d logits (channels-last rows).
//   partial layout per block: [ce_sum, n_px, I_c (C), Z_c (C), Y_c (C)]            (supervised)
//                             per image: [n_conf, n_valid, sum(CE>0), n(CE>0)]     (unsupervised)
#define SEG_MAXP (2 + 3 * GL_MAXC)
// CM = compile-time bound of the class count (4 / 8 / GL_MAXC): loops fully unrolled over CM, accumulators indexed
// statically (the runtime-C indexing kept 98 fp64 accumulators in scratch memory: 74 us for 0.5 M rows)
template <int CM>
__global__ __launch_bounds__(256) void sup_loss_partial_kernel(const float* __restrict__ X, long ld, long M, int C,
                                                              const int64_t* __restrict__ lab, double* __restrict__ part) {
  // accumulators in a compile-time layout [2 + 3*CM] (registers); compacted to the [2 + 3*C] slab layout on the way out
  double acc[2 + 3 * CM];
#pragma unroll
  for (int i = 0; i < 2 + 3 * CM; ++i) acc[i] = 0.0;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    const float* x = X + r * ld;
    float v[CM]; float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) { v[c] = x[c]; mx = fmaxf(mx, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) s += expf(v[c] - mx);
    const float lse = mx + logf(s);
    const int64_t l = lab[r];
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) {
      const float p = expf(v[c] - lse);
      const float t = (l == c) ? 1.f : 0.f;
      if (l == c) acc[0] += (double)(logf(s) - (v[c] - mx));
      acc[2 + c] += (double)(p * t); acc[2 + CM + c] += (double)(p * p); acc[2 + 2 * CM + c] += (double)t;
    }
    acc[1] += 1.0;
  }
  __shared__ double sh[4][2 + 3 * CM];
#pragma unroll
  for (int i = 0; i < 2 + 3 * CM; ++i) {
    const double w = wave_sum_d(acc[i]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = w;
  }
  __syncthreads();
  const int np = 2 + 3 * C;
  if (threadIdx.x < np) {
    const int t = threadIdx.x, src = t < 2 ? t : 2 + ((t - 2) / C) * CM + (t - 2) % C;
    part[(long)blockIdx.x * np + t] = (sh[0][src] + sh[1][src]) + (sh[2][src] + sh[3][src]);
  }
}
// sums[0..np) = sum over blocks; out[0] = CE mean, out[1] = dice
__global__ void sup_loss_final_kernel(const double* __restrict__ part, int nblk, int C, double* __restrict__ sums, float* __restrict__ out) {
  const int np = 2 + 3 * C;
  __shared__ double s[SEG_MAXP];
  // one wave per quantity (blockDim = 256 = 4 waves), fixed summation order
  for (int i = threadIdx.x >> 6; i < np; i += 4) {
    double a = 0.0;
    for (int b = threadIdx.x & 63; b < nblk; b += 64) a += part[(long)b * np + i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) { s[i] = a; sums[i] = a; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (float)(s[0] / s[1]);
    double d = 0.0;
    for (int c = 0; c < C; ++c) d += 1.0 - (2.0 * s[2 + c] + 1e-5) / (s[2 + C + c] + s[2 + 2 * C + c] + 1e-5);
    out[1] = (float)(d / C);
  }
}
// dX = g_ce * (p - t)/M + g_dice * softmax-Jacobian^T * dDice/dp
template <int CM>
__global__ __launch_bounds__(256) void sup_loss_bwd_kernel(const float* __restrict__ X, long ld, long M, int C,
                                                          const int64_t* __restrict__ lab, const double* __restrict__ sums,
                                                          const float* __restrict__ g_ce, const float* __restrict__ g_dice,
                                                          float* __restrict__ dX, long ldo) {
  const float gce = g_ce[0] / (float)M, gd = g_dice[0] / (float)C;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    const float* x = X + r * ld;
    float v[CM]; float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) { v[c] = x[c]; mx = fmaxf(mx, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
    const int64_t l = lab[r];
    float dp[CM]; float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) {
      v[c] /= s;
      const float t = (l == c) ? 1.f : 0.f;
      const double den = sums[2 + C + c] + sums[2 + 2 * C + c] + 1e-5, num = 2.0 * sums[2 + c] + 1e-5;
      dp[c] = -gd * (float)((2.0 * t * den - num * 2.0 * v[c]) / (den * den));   // d dice / d p_c
      dot += dp[c] * v[c];
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C)
      dX[r * ldo + c] = gce * (v[c] - ((l == c) ? 1.f : 0.f)) + v[c] * (dp[c] - dot);
  }
}

// unsupervised: per-image partials [n_conf, n_valid, sum CE (where CE>0), n (CE>0)]; one block column per image
__global__ __launch_bounds__(256) void unsup_loss_partial_kernel(const float* __restrict__ X, long ld, long P, int C,
                                                                const int64_t* __restrict__ lab, const float* __restrict__ conf,
                                                                float thr, double* __restrict__ part /*[B][gridDim.x][4]*/) {
  const long img = blockIdx.y;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (long sp = (long)blockIdx.x * 256 + threadIdx.x; sp < P; sp += (long)gridDim.x * 256) {
    const long r = img * P + sp;
    const int64_t l = lab[r];
    a0 += conf[r] >= thr; a1 += l >= 0;
    if (l >= 0) {
      const float* x = X + r * ld;
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(x[c] - mx);
      // torch's log_softmax association, (x - max) - log(sum): with saturated logits `max + log(sum)` rounds log(sum) away
      // and CE > 0 - which selects the elements of the mean (train_arco_2d.py:488) - would drop rows torch keeps
      const float ce = logf(s) - (x[l] - mx);
      if (ce > 0.f) { a2 += (double)ce; a3 += 1.0; }
    }
  }
  __shared__ double sh[4][4];
  double w;
  w = wave_sum_d(a0); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][0] = w;
  w = wave_sum_d(a1); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][1] = w;
  w = wave_sum_d(a2); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][2] = w;
  w = wave_sum_d(a3); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][3] = w;
  __syncthreads();
  if (threadIdx.x < 4) part[(img * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// per-image weights w_b = n_conf/n_valid; loss = sum_b w_b * S_b / sum_b n_b ; stats[b] = {w_b}, stats[B] = N_sel
__global__ void unsup_loss_final_kernel(const double* __restrict__ part, int B, int nblk, double* __restrict__ stats, float* __restrict__ out) {
  double num = 0.0, cnt = 0.0;                       // one wave; lanes stride over the per-image partials
  for (int b = 0; b < B; ++b) {
    double s[4] = {0, 0, 0, 0};
    for (int k = threadIdx.x; k < nblk; k += 64) for (int i = 0; i < 4; ++i) s[i] += part[((long)b * nblk + k) * 4 + i];
    for (int i = 0; i < 4; ++i) s[i] = wave_sum_d(s[i]);
    const double w = s[0] / s[1];
    if (threadIdx.x == 0) stats[b] = w;
    num += w * s[2]; cnt += s[3];
  }
  if (threadIdx.x != 0) return;
  stats[B] = cnt;
  out[0] = (float)(num / cnt);
}
__global__ __launch_bounds__(256) void unsup_loss_bwd_kernel(const float* __restrict__ X, long ld, long P, long M, int C,
                                                            const int64_t* __restrict__ lab, const double* __restrict__ stats, int B,
                                                            const float* __restrict__ g, float* __restrict__ dX, long ldo) {
  const float gs = g[0] / (float)stats[B];
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    const int64_t l = lab[r];
    const float* x = X + r * ld;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - mx);
    const float lse = mx + logf(s);
    const bool sel = l >= 0 && (logf(s) - (x[l >= 0 ? l : 0] - mx)) > 0.f;
    const float w = sel ? gs * (float)stats[r / P] : 0.f;
    for (int c = 0; c < C; ++c) dX[r * ldo + c] = sel ? w * (expf(x[c] - lse) - (l == c ? 1.f : 0.f)) : 0.f;
  }
}


// ---- evaluation: per-class overlap counts of two label maps (test_2D.py:52-66, medpy dc / jc):
//      out[c] = {|pred == c|, |gt == c|, |pred == c and gt == c|}; integer atomics -> deterministic
__global__ __launch_bounds__(256) void overlap_counts_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ gt,
                                                            long n, int C, unsigned long long* __restrict__ out) {
  __shared__ unsigned int h[3 * GL_MAXC];
  for (int i = threadIdx.x; i < 3 * C; i += 256) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int64_t p = pred[i], g = gt[i];
    if (p >= 0 && p < C) atomicAdd(&h[3 * p], 1u);
    if (g >= 0 && g < C) atomicAdd(&h[3 * g + 1], 1u);
    if (p == g && p >= 0 && p < C) atomicAdd(&h[3 * p + 2], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) if (h[i]) atomicAdd(&out[i], (unsigned long long)h[i]);
}

// ---- equivariance loss (tps/rand_tps.py, tps_stn_pytorch/tps_grid_gen.py, train_arco_2d.py:404-423)
// grid[b][p][0..1] = rep[p][0..NR) . mapping[b][0..NR)[0..1]   (TPSGridGen.forward, NR = 25 control points + 3)
// One pixel per lane; the 28 basis values of the pixel in registers (seven 16-byte loads), the 2 x NR mapping coefficients of an
// image read through wave-uniform addresses (scalar loads: they cost no vector or LDS issue), blockIdx.y = a group of BG images.
// Round 6: the first form kept the mapping in LDS and ran one 256-thread block per CU over all B images - 1800 dependent
// broadcast ds_read_b32 per lane, 356 us per launch (the longest single launch of the 2-D step); this one ~15 us.  The launch
// sits on the second stream beside the contrastive stage: the replayed step did not move (10.51-10.70 -> 10.68-10.79 ms, same box).
// The sums run in the same order (k ascending, one fused multiply-add per term): results unchanged bit for bit.
constexpr int TPS_BG = 4;
__global__ __launch_bounds__(256) void tps_grid_kernel(const float* __restrict__ rep, const float* __restrict__ mapping, int B,
                                                      long HW, int NR, float* __restrict__ grid) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float r[32];                           // (static indexing: registers, no scratch - a scratch-using kernel's first launch
  if ((NR & 3) == 0) {                   //  waits ~28 ms for the runtime to set the scratch arena up)
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (4 * k4 < NR) v = *reinterpret_cast<const f32x4*>(rep + p * NR + 4 * k4);
      r[4 * k4] = v[0]; r[4 * k4 + 1] = v[1]; r[4 * k4 + 2] = v[2]; r[4 * k4 + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = k < NR ? rep[p * NR + k] : 0.f;
  }
  const int b0 = blockIdx.y * TPS_BG, b1 = min(B, b0 + TPS_BG);
  for (int b = b0; b < b1; ++b) {
    const float* __restrict__ m = mapping + (long)b * NR * 2;       // wave-uniform
    float gx = 0.f, gy = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (k < NR) { gx += r[k] * m[2 * k]; gy += r[k] * m[2 * k + 1]; }
    grid[((long)b * HW + p) * 2] = gx; grid[((long)b * HW + p) * 2 + 1] = gy;
  }
}
// ---- A2  AdvMorph (adv_morph.py:310-580): random diffeomorphic warp of the unlabeled stream -------------------------------
// 2-channel fields are kept channels-last [B, H, W, 2] = (x, y) per pixel, i.e. a field IS a grid_sample grid:
// applyComposition2D(f1, f2) (adv_morph.py:297-307) is grid_sample_fwd_kernel with X = f1, grid = f2, border padding.
// base grid value of pixel (i, j): (linspace(-1, 1, W)[j], linspace(-1, 1, H)[i])  (get_base_grid, adv_morph.py:184-207);
// torch.linspace fills from both ends: start + i*step for i < n/2, end - (n-1-i)*step otherwise.
__device__ __forceinline__ float linspace_m11(int i, int n) {
  if (n == 1) return -1.f;
  const float step = 2.f / (float)(n - 1);
  return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}
// out = alpha * in + beta * base_grid + gamma * in2  (in / in2 may be NULL), optionally clamped to [-1, 1]
__global__ void field_axpb_kernel(const float* __restrict__ in, float alpha, float beta, const float* __restrict__ in2, float gamma,
                                  int B, int H, int W, int clamp, float* __restrict__ out) {
  const long tot = (long)B * H * W * 2;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t & 1); const long p = t >> 1; const int x = (int)(p % W), y = (int)((p / W) % H);
    const float g = c == 0 ? linspace_m11(x, W) : linspace_m11(y, H);
    float v = beta * g;
    if (in) v = alpha * in[t] + v;
    if (in2) v = v + gamma * in2[t];
    if (clamp) v = fminf(fmaxf(v, -1.f), 1.f);
    out[t] = v;
  }
}
// depthwise KS x KS filter with zero padding (gaussian_smooth / get_gaussian_kernel, adv_morph.py:445-497); weights by value
struct SmoothW { float w[81]; };
__global__ void field_smooth_kernel(const float* __restrict__ in, int B, int H, int W, int C, int KS, SmoothW sw,
                                    float* __restrict__ out) {
  const long tot = (long)B * H * W * C;
  const int r = KS / 2;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C); const long p = t / C; const int x = (int)(p % W), y = (int)((p / W) % H); const long n = p / ((long)W * H);
    float acc = 0.f;
    for (int dy = 0; dy < KS; ++dy) {
      const int yy = y + dy - r;
      if (yy < 0 || yy >= H) continue;
      for (int dx = 0; dx < KS; ++dx) {
        const int xx = x + dx - r;
        if (xx < 0 || xx >= W) continue;
        acc += sw.w[dy * KS + dx] * in[((n * H + yy) * W + xx) * C + c];
      }
    }
    out[t] = acc;
  }
}
// F.interpolate(mode='bilinear', align_corners=False) of a channels-last field (DemonsCompose, adv_morph.py:507-508)
__global__ void field_resize_kernel(const float* __restrict__ in, int B, int h, int w, int C, int H, int W, float* __restrict__ out) {
  const long tot = (long)B * H * W * C;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C); const long p = t / C; const int X = (int)(p % W), Y = (int)((p / W) % H); const long n = p / ((long)W * H);
    float fy = sy * ((float)Y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = sx * ((float)X + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + n * (long)h * w * C + c;
    out[t] = hy * (hx * b[((long)y0 * w + x0) * C] + lx * b[((long)y0 * w + x1) * C]) +
             ly * (hx * b[((long)y1 * w + x0) * C] + lx * b[((long)y1 * w + x1) * C]);
  }
}
// F.grid_sample(mode='bilinear', align_corners=True), channels-last rows; padding 0: zeros, 1: border.
// D3 > 1: the same 2-D warp applied to every slice z of a volume [NB, H, W, D3, C] (tps/rand_tps_3d.py:147-166)
__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(const float* __restrict__ X, long ldx, int NB, int H, int W, int D3, int C,
                                                             const float* __restrict__ grid, int Ho, int Wo, int border,
                                                             float* __restrict__ Y, long ldy) {
  const long tot = (long)NB * Ho * Wo * D3 * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C); const long pz = i / C; const int z = (int)(pz % D3); const long po = pz / D3;
    const int n = (int)(po / ((long)Ho * Wo));
    float ix = (grid[po * 2] + 1.f) * 0.5f * (float)(W - 1), iy = (grid[po * 2 + 1] + 1.f) * 0.5f * (float)(H - 1);
    if (border) { ix = fminf(fmaxf(ix, 0.f), (float)(W - 1)); iy = fminf(fmaxf(iy, 0.f), (float)(H - 1)); }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const float* base = X + ((long)n * H * W * D3 + z) * ldx + c;
    const long sx = (long)D3 * ldx, sy = (long)W * D3 * ldx;
    float v = 0.f;
    if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) v += base[y0 * sy + x0 * sx] * (wx0 * wy0);
    if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) v += base[y0 * sy + x1 * sx] * (wx1 * wy0);
    if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) v += base[y1 * sy + x0 * sx] * (wx0 * wy1);
    if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) v += base[y1 * sy + x1 * sx] * (wx1 * wy1);
    Y[pz * ldy + c] = v;
  }
}
// masked KL(softmax(q) || softmax(p)) per image: partial [b][blk] = {sum mask*kl, sum mask}
__global__ __launch_bounds__(256) void eqv_loss_partial_kernel(const float* __restrict__ Pm, long ldp, const float* __restrict__ Qm, long ldq,
                                                              const float* __restrict__ mask, long P, int C, double* __restrict__ part) {
  const long img = blockIdx.y;
  double a0 = 0.0, a1 = 0.0;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < P; r += (long)gridDim.x * 256) {
    const long row = img * P + r;
    const float m = mask[row];
    const float* p = Pm + row * ldp; const float* q = Qm + row * ldq;
    float mp = -INFINITY, mq = -INFINITY;
    for (int c = 0; c < C; ++c) { mp = fmaxf(mp, p[c]); mq = fmaxf(mq, q[c]); }
    float sp = 0.f, sq = 0.f;
    for (int c = 0; c < C; ++c) { sp += expf(p[c] - mp); sq += expf(q[c] - mq); }
    const float lsp = mp + logf(sp), lsq = mq + logf(sq);
    float kl = 0.f;
    for (int c = 0; c < C; ++c) {
      const float lt = q[c] - lsq, t = expf(lt);
      if (t > 0.f) kl += t * (lt - (p[c] - lsp));
    }
    a0 += (double)(kl * m); a1 += (double)m;
  }
  __shared__ double sh[4][2];
  double w0 = wave_sum_d(a0), w1 = wave_sum_d(a1);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = w0; sh[threadIdx.x >> 6][1] = w1; }
  __syncthreads();
  if (threadIdx.x < 2) part[(img * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
// den[b] = sum mask + 1e-7; out = mean_b(num_b / den_b)
__global__ void eqv_loss_final_kernel(const double* __restrict__ part, int B, int nblk, double* __restrict__ den, float* __restrict__ out) {
  double tot = 0.0;
  for (int b = 0; b < B; ++b) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) { s0 += part[((long)b * nblk + k) * 2]; s1 += part[((long)b * nblk + k) * 2 + 1]; }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    const double d = s1 + 1e-7;
    if (threadIdx.x == 0) den[b] = d;
    tot += s0 / d;
  }
  if (threadIdx.x == 0) out[0] = (float)(tot / B);
}
// d loss / d p = g * mask / (den_b * B) * (softmax(p) - softmax(q))
__global__ __launch_bounds__(256) void eqv_loss_bwd_kernel(const float* __restrict__ Pm, long ldp, const float* __restrict__ Qm, long ldq,
                                                          const float* __restrict__ mask, long P, long M, int C, int B,
                                                          const double* __restrict__ den, const float* __restrict__ g,
                                                          float* __restrict__ dP, long ldo) {
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < M; row += (long)gridDim.x * 256) {
    const float w = g[0] * mask[row] / (float)(den[row / P] * (double)B);
    const float* p = Pm + row * ldp; const float* q = Qm + row * ldq;
    float mp = -INFINITY, mq = -INFINITY;
    for (int c = 0; c < C; ++c) { mp = fmaxf(mp, p[c]); mq = fmaxf(mq, q[c]); }
    float sp = 0.f, sq = 0.f;
    for (int c = 0; c < C; ++c) { sp += expf(p[c] - mp); sq += expf(q[c] - mq); }
    for (int c = 0; c < C; ++c) dP[row * ldo + c] = w * (expf(p[c] - mp) / sp - expf(q[c] - mq) / sq);
  }
}

static inline int gl_grid(long work) { long g = (work + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1; return (int)g; }
// slabs per image of the per-image loss partials (unsupervised CE, equivariance): 64 with >= 8 images, more with fewer (a 1+1-volume 3-D step has ONE image of
// 2.5 M voxels: 64 blocks left three quarters of the chip idle, 0.26 ms per call)
static inline int unsup_nblk(int B) { return B >= 8 ? 64 : (B >= 4 ? 128 : (B >= 2 ? 256 : 512)); }

// ---- utils/losses.py:173-209 DiceLoss on PROBABILITIES (the class the reference trainers instantiate: train_arco_2d.py:269,338):
// rows [M, C] of scores (any values, softmax already applied by the caller), integer labels; per class
//   dice_c = 1 - (2 I_c + 1e-5) / (Z_c + Y_c + 1e-5),  I = sum p t, Z = sum p^2, Y = sum t^2;  loss = sum_c w_c dice_c / C
// partial layout per block: [I_c (C), Z_c (C), Y_c (C)]; fp64 fixed-order finalize; backward = one elementwise pass.
template <int CM>
__global__ __launch_bounds__(256) void dice_probs_partial_kernel(const float* __restrict__ Pm, long ld, long M, int C,
                                                                const int64_t* __restrict__ lab, double* __restrict__ part) {
  double acc[3 * CM];
#pragma unroll
  for (int i = 0; i < 3 * CM; ++i) acc[i] = 0.0;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
    const float* x = Pm + r * ld;
    const int64_t l = lab[r];
#pragma unroll
    for (int c = 0; c < CM; ++c) if (c < C) {
      const float p = x[c], t = (l == c) ? 1.f : 0.f;
      acc[c] += (double)(p * t); acc[CM + c] += (double)(p * p); acc[2 * CM + c] += (double)t;
    }
  }
  __shared__ double sh[4][3 * CM];
#pragma unroll
  for (int i = 0; i < 3 * CM; ++i) {
    const double w = wave_sum_d(acc[i]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = w;
  }
  __syncthreads();
  const int np = 3 * C;
  if (threadIdx.x < np) {
    const int t = threadIdx.x, src = (t / C) * CM + t % C;
    part[(long)blockIdx.x * np + t] = (sh[0][src] + sh[1][src]) + (sh[2][src] + sh[3][src]);
  }
}
__global__ void dice_probs_final_kernel(const double* __restrict__ part, int nblk, int C, const float* __restrict__ wgt,
                                        double* __restrict__ sums, float* __restrict__ out) {
  const int np = 3 * C;
  __shared__ double s[3 * GL_MAXC];
  for (int i = threadIdx.x >> 6; i < np; i += 4) {
    double a = 0.0;
    for (int b = threadIdx.x & 63; b < nblk; b += 64) a += part[(long)b * np + i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) { s[i] = a; sums[i] = a; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double d = 0.0;
    for (int c = 0; c < C; ++c)
      d += (double)(wgt ? wgt[c] : 1.f) * (1.0 - (2.0 * s[c] + 1e-5) / (s[C + c] + s[2 * C + c] + 1e-5));
    out[0] = (float)(d / C);
  }
}
__global__ __launch_bounds__(256) void dice_probs_bwd_kernel(const float* __restrict__ Pm, long ld, long M, int C,
                                                            const int64_t* __restrict__ lab, const double* __restrict__ sums,
                                                            const float* __restrict__ wgt, const float* __restrict__ g,
                                                            float* __restrict__ dP, long ldo) {
  const float gd = g[0] / (float)C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * C; i += (long)gridDim.x * 256) {
    const long r = i / C; const int c = (int)(i - r * C);
    const double t = (lab[r] == c) ? 1.0 : 0.0, p = (double)Pm[r * ld + c];
    const double den = sums[C + c] + sums[2 * C + c] + 1e-5, num = 2.0 * sums[c] + 1e-5;
    dP[r * ldo + c] = -gd * (wgt ? wgt[c] : 1.f) * (float)((2.0 * t * den - num * 2.0 * p) / (den * den));
  }
}

extern "C" {

int arco_softmax_rows(const float* X, long ld, long M, int C, long P, float* prob_planes, float* maxp, int64_t* amax,
                      float* entropy, void* stream) {
  ARCO_CHECK_ARG(C >= 1 && C <= GL_MAXC && M > 0 && P > 0);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(gl_grid(M)), dim3(256), 0, as_stream(stream), X, ld, M, C, P,
                     prob_planes, maxp, amax, entropy);
  return arco_launch_status();
}

int arco_label_onehot(const int64_t* lab, long M, int C, long P, int64_t* out, void* stream) {
  hipLaunchKernelGGL(onehot_kernel, dim3(gl_grid(M)), dim3(256), 0, as_stream(stream), lab, M, C, P, out);
  return arco_launch_status();
}

// TPS grid: rep [HW][NR], mapping [B][NR][2] -> grid [B][HW][2]   (NR <= 32)
int arco_tps_grid(const float* rep, const float* mapping, int B, long HW, int NR, float* grid, void* stream) {
  ARCO_CHECK_ARG(rep && mapping && grid && B > 0 && HW > 0 && NR > 0 && NR <= 32 && (HW + 255) / 256 < (1l << 31));
  hipLaunchKernelGGL(tps_grid_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)((B + TPS_BG - 1) / TPS_BG)), dim3(256), 0, as_stream(stream), rep,
                     mapping, B, HW, NR, grid);
  return arco_launch_status();
}
int arco_grid_sample_fwd(const float* X, long ldx, int NB, int H, int W, int D3, int C, const float* grid, int Ho, int Wo,
                         int border, float* Y, long ldy, void* stream) {
  ARCO_CHECK_ARG(X && grid && Y && NB > 0 && H > 0 && W > 0 && D3 > 0 && C > 0 && Ho > 0 && Wo > 0);
  hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(gl_grid((long)NB * Ho * Wo * D3 * C)), dim3(256), 0, as_stream(stream), X, ldx,
                     NB, H, W, D3, C, grid, Ho, Wo, border, Y, ldy);
  return arco_launch_status();
}
/* AdvMorph fields (adv_morph.py:184-207, 260-307, 445-532), channels-last [B, H, W, 2] */
int arco_field_axpb(const float* in, float alpha, float beta, const float* in2, float gamma, int B, int H, int W, int clamp,
                    float* out, void* stream) {
  ARCO_CHECK_ARG(out && B > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(field_axpb_kernel, dim3(gl_grid((long)B * H * W * 2)), dim3(256), 0, as_stream(stream), in, alpha, beta, in2,
                     gamma, B, H, W, clamp, out);
  return arco_launch_status();
}
int arco_field_smooth(const float* in, int B, int H, int W, int C, int ks, const float* weights_host, float* out, void* stream) {
  ARCO_CHECK_ARG(in && out && weights_host && ks >= 1 && ks <= 9 && (ks & 1) && B > 0 && H > 0 && W > 0 && C > 0);
  SmoothW sw;
  for (int i = 0; i < ks * ks; ++i) sw.w[i] = weights_host[i];
  hipLaunchKernelGGL(field_smooth_kernel, dim3(gl_grid((long)B * H * W * C)), dim3(256), 0, as_stream(stream), in, B, H, W, C, ks, sw, out);
  return arco_launch_status();
}
int arco_field_resize(const float* in, int B, int h, int w, int C, int H, int W, float* out, void* stream) {
  ARCO_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0);
  hipLaunchKernelGGL(field_resize_kernel, dim3(gl_grid((long)B * H * W * C)), dim3(256), 0, as_stream(stream), in, B, h, w, C, H, W, out);
  return arco_launch_status();
}
// ws: arco_loss_slabs(B) * 2 * B + B doubles; out[0] = loss_eqv
int arco_eqv_loss_fwd(const float* P_, long ldp, const float* Q_, long ldq, const float* mask, int B, long P, int C, double* ws,
                      float* out, void* stream) {
  ARCO_CHECK_ARG(P_ && Q_ && mask && ws && out && B > 0 && P > 0 && C >= 1);
  const int nblk = unsup_nblk(B);
  hipLaunchKernelGGL(eqv_loss_partial_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), P_, ldp, Q_, ldq, mask, P, C, ws);
  hipLaunchKernelGGL(eqv_loss_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), ws, B, nblk, ws + (long)nblk * 2 * B, out);
  return arco_launch_status();
}
int arco_eqv_loss_bwd(const float* P_, long ldp, const float* Q_, long ldq, const float* mask, int B, long P, int C,
                      const double* ws, const float* g, float* dP, long ldo, void* stream) {
  hipLaunchKernelGGL(eqv_loss_bwd_kernel, dim3(gl_grid((long)B * P)), dim3(256), 0, as_stream(stream), P_, ldp, Q_, ldq, mask, P,
                     (long)B * P, C, B, ws + (long)unsup_nblk(B) * 2 * B, g, dP, ldo);
  return arco_launch_status();
}

// ---- A  mixing strategies of the unlabeled stream (augment.py:284-313 generate_unsup_data) ----
// desc[i] = {y0, y1, x0, x1, z0, z1, sel_lo, sel_hi}: the zero box of the cutout mask (augment.py:230-244; volumes
// augment_3d.py:182-198, Z = 1 and z = [0, 1) in 2-D) for modes 0/1, or the 64-bit set of selected labels
// (generate_class_mask, :247-252) for mode 2.  mask = 1 outside the box / where the
// pixel's label is selected.   mode 0 (cutmix) and 2 (classmix): out_i = mask ? src_i : src_{(i+1) % B}
// mode 1 (cutout): data, logits *= mask; target = -1 where mask == 0.       data is NC[spatial] (the loader's layout).
struct MixDescs { int d[32][8]; };     // by value in the kernel arguments: no H2D copy, no host synchronisation
__global__ void mix_unsup_kernel(const float* __restrict__ data, int Cimg, const int64_t* __restrict__ target,
                                 const float* __restrict__ logits, int B, int H, int W, int Z, MixDescs dd, int i0, int nimg, int mode,
                                 float* __restrict__ odata, int64_t* __restrict__ otarget, float* __restrict__ ologits) {
  const long HW = (long)H * W * Z, n = (long)nimg * HW;
  for (long tt = (long)blockIdx.x * blockDim.x + threadIdx.x; tt < n; tt += (long)gridDim.x * blockDim.x) {
    const int il = tt / HW, i = i0 + il; const long p = tt - (long)il * HW, t = (long)i * HW + p;
    const int z = p % Z; const long q = p / Z; const int y = q / W, x = q - (long)y * W;
    const int* d = dd.d[il];
    bool keep;
    if (mode == 2) {
      const int64_t lab = target[t];
      const unsigned long long sel = ((unsigned long long)(unsigned)d[7] << 32) | (unsigned)d[6];
      keep = lab >= 0 && lab < 64 && ((sel >> lab) & 1ull);
    } else keep = !(y >= d[0] && y < d[1] && x >= d[2] && x < d[3] && z >= d[4] && z < d[5]);
    if (mode == 1) {
      for (int c = 0; c < Cimg; ++c) { const long o = ((long)i * Cimg + c) * HW + p; odata[o] = keep ? data[o] : 0.f; }
      otarget[t] = keep ? target[t] : -1;
      ologits[t] = keep ? logits[t] : 0.f;
    } else {
      const int j = keep ? i : (i + 1) % B;
      for (int c = 0; c < Cimg; ++c) odata[((long)i * Cimg + c) * HW + p] = data[((long)j * Cimg + c) * HW + p];
      otarget[t] = target[(long)j * HW + p];
      ologits[t] = logits[(long)j * HW + p];
    }
  }
}
// presence[i] = 64-bit set of the labels occurring in target[i] (torch.unique per image, labels in [0, 64))
__global__ void label_presence_kernel(const int64_t* __restrict__ target, long HW, unsigned long long* __restrict__ presence) {
  unsigned long long m = 0;
  const int64_t* t = target + (long)blockIdx.y * HW;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    const int64_t l = t[p];
    if (l >= 0 && l < 64) m |= 1ull << l;
  }
  for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
  if ((threadIdx.x & 63) == 0 && m) atomicOr(&presence[blockIdx.y], m);
}
int arco_mix_unsup(const float* data, int Cimg, const int64_t* target, const float* logits, int B, int H, int W, int Z,
                   const int* desc_host, int mode, float* odata, int64_t* otarget, float* ologits, void* stream) {
  ARCO_CHECK_ARG(data && target && logits && desc_host && odata && otarget && ologits && B > 0 && H > 0 && W > 0 && Z > 0 &&
                 Cimg > 0 && mode >= 0 && mode <= 2);
  for (int i0 = 0; i0 < B; i0 += 32) {
    const int nimg = B - i0 < 32 ? B - i0 : 32;
    MixDescs dd;
    memcpy(dd.d, desc_host + 8l * i0, sizeof(int) * 8 * nimg);
    hipLaunchKernelGGL(mix_unsup_kernel, dim3(gl_grid((long)nimg * H * W * Z)), dim3(256), 0, as_stream(stream), data, Cimg, target,
                       logits, B, H, W, Z, dd, i0, nimg, mode, odata, otarget, ologits);
  }
  return arco_launch_status();
}
int arco_label_presence(const int64_t* target, int B, long HW, uint64_t* presence, void* stream) {
  ARCO_CHECK_ARG(target && presence && B > 0 && HW > 0);
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(presence, 0, (size_t)B * 8, st) != hipSuccess) return -2;
  long g = (HW + 255) / 256; if (g > 64) g = 64;
  hipLaunchKernelGGL(label_presence_kernel, dim3((unsigned)g, B), dim3(256), 0, st, target, HW,
                     reinterpret_cast<unsigned long long*>(presence));
  return arco_launch_status();
}

// ---- V  3-D sliding-window evaluation (test_util.py:139-211) ----
// score[c][x][y][z] += prob[c][i][j][k] over the window at (xs, ys, zs); cnt += 1.  One thread per window voxel; windows are
// accumulated by successive launches on one stream, i.e. in the reference's (x, y, z) loop order -> same fp32 sums.
__global__ void window_accumulate_kernel(const float* __restrict__ prob, int C, int px, int py, int pz, float* __restrict__ score,
                                         float* __restrict__ cnt, long hh, long dd, long vol, int xs, int ys, int zs) {
  const long pv = (long)px * py * pz;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pv; i += (long)gridDim.x * blockDim.x) {
    const int k = i % pz; const long r = i / pz; const int j = r % py, ii = r / py;
    const long o = ((long)(xs + ii) * hh + (ys + j)) * dd + (zs + k);
    for (int c = 0; c < C; ++c) score[c * vol + o] += prob[c * pv + i];
    cnt[o] += 1.f;
  }
}
// score /= cnt (in place, fp32 like the numpy division); label = first argmax over classes (np.argmax)
__global__ void score_finalize_kernel(float* __restrict__ score, const float* __restrict__ cnt, int C, long vol,
                                      int64_t* __restrict__ label) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < vol; i += (long)gridDim.x * blockDim.x) {
    const float n = cnt[i];
    float best = 0.f; int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float v = score[c * vol + i] / n;
      score[c * vol + i] = v;
      if (c == 0 || v > best) { best = v; arg = c; }
    }
    label[i] = arg;
  }
}
int arco_window_accumulate(const float* prob, int C, int px, int py, int pz, float* score, float* cnt, int ww, int hh, int dd,
                           int xs, int ys, int zs, void* stream) {
  ARCO_CHECK_ARG(prob && score && cnt && C > 0 && px > 0 && py > 0 && pz > 0 && xs >= 0 && ys >= 0 && zs >= 0 &&
                 xs + px <= ww && ys + py <= hh && zs + pz <= dd);
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(gl_grid((long)px * py * pz)), dim3(256), 0, as_stream(stream), prob, C, px, py,
                     pz, score, cnt, (long)hh, (long)dd, (long)ww * hh * dd, xs, ys, zs);
  return arco_launch_status();
}
int arco_score_finalize(float* score, const float* cnt, int C, long vol, int64_t* label, void* stream) {
  ARCO_CHECK_ARG(score && cnt && label && C > 0 && vol > 0);
  hipLaunchKernelGGL(score_finalize_kernel, dim3(gl_grid(vol)), dim3(256), 0, as_stream(stream), score, cnt, C, vol, label);
  return arco_launch_status();
}

// out: C*3 int64 counters {pred, gt, both} per class (zeroed here)
int arco_overlap_counts(const int64_t* pred, const int64_t* gt, long n, int C, int64_t* out, void* stream) {
  ARCO_CHECK_ARG(pred && gt && out && n > 0 && C >= 1 && C <= GL_MAXC);
  (void)hipMemsetAsync(out, 0, sizeof(int64_t) * 3 * C, as_stream(stream));
  int g = gl_grid(n); if (g > 1024) g = 1024;
  hipLaunchKernelGGL(overlap_counts_kernel, dim3(g), dim3(256), 0, as_stream(stream), pred, gt, n, C,
                     reinterpret_cast<unsigned long long*>(out));
  return arco_launch_status();
}

// supervised CE + Dice: ws = arco_seg_ws_doubles(...) doubles; out[0] = CE, out[1] = Dice
long arco_seg_ws_doubles(long M, int C, int B) { return 1024l * (2 + 3 * C) + (2 + 3 * C) + 1024l * 4 * (B > 0 ? B : 1) + B + 1; }
int arco_sup_loss_fwd(const float* X, long ld, long M, int C, const int64_t* lab, double* ws, float* out, void* stream) {
  ARCO_CHECK_ARG(C >= 1 && C <= GL_MAXC && M > 0);
  int nblk = gl_grid(M); if (nblk > 1024) nblk = 1024;
  double* sums = ws + 1024l * (2 + 3 * C);
  if (C <= 4) hipLaunchKernelGGL(sup_loss_partial_kernel<4>, dim3(nblk), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, ws);
  else if (C <= 8) hipLaunchKernelGGL(sup_loss_partial_kernel<8>, dim3(nblk), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, ws);
  else hipLaunchKernelGGL(sup_loss_partial_kernel<GL_MAXC>, dim3(nblk), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, ws);
  hipLaunchKernelGGL(sup_loss_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), ws, nblk, C, sums, out);
  return arco_launch_status();
}
int arco_sup_loss_bwd(const float* X, long ld, long M, int C, const int64_t* lab, const double* ws, const float* g_ce,
                      const float* g_dice, float* dX, long ldo, void* stream) {
  const double* sums = ws + 1024l * (2 + 3 * C);
  if (C <= 4) hipLaunchKernelGGL(sup_loss_bwd_kernel<4>, dim3(gl_grid(M)), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, sums, g_ce, g_dice, dX, ldo);
  else if (C <= 8) hipLaunchKernelGGL(sup_loss_bwd_kernel<8>, dim3(gl_grid(M)), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, sums, g_ce, g_dice, dX, ldo);
  else hipLaunchKernelGGL(sup_loss_bwd_kernel<GL_MAXC>, dim3(gl_grid(M)), dim3(256), 0, as_stream(stream), X, ld, M, C, lab, sums, g_ce, g_dice, dX, ldo);
  return arco_launch_status();
}
// DiceLoss on probability rows (utils/losses.py:173-209): ws >= 1024*3*C + 3*C doubles (arco_seg_ws_doubles covers it);
// wgt = per-class weights or null; out[0] = loss
int arco_dice_probs_fwd(const float* Pm, long ld, long M, int C, const int64_t* lab, const float* wgt, double* ws, float* out,
                        void* stream) {
  ARCO_CHECK_ARG(Pm && lab && ws && out && C >= 1 && C <= GL_MAXC && M > 0 && ld >= C);
  int nblk = gl_grid(M); if (nblk > 1024) nblk = 1024;
  double* sums = ws + 1024l * 3 * C;
  if (C <= 4) hipLaunchKernelGGL(dice_probs_partial_kernel<4>, dim3(nblk), dim3(256), 0, as_stream(stream), Pm, ld, M, C, lab, ws);
  else if (C <= 8) hipLaunchKernelGGL(dice_probs_partial_kernel<8>, dim3(nblk), dim3(256), 0, as_stream(stream), Pm, ld, M, C, lab, ws);
  else hipLaunchKernelGGL(dice_probs_partial_kernel<GL_MAXC>, dim3(nblk), dim3(256), 0, as_stream(stream), Pm, ld, M, C, lab, ws);
  hipLaunchKernelGGL(dice_probs_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), ws, nblk, C, wgt, sums, out);
  return arco_launch_status();
}
int arco_dice_probs_bwd(const float* Pm, long ld, long M, int C, const int64_t* lab, const float* wgt, const double* ws,
                        const float* g, float* dP, long ldo, void* stream) {
  ARCO_CHECK_ARG(Pm && lab && ws && g && dP && C >= 1 && C <= GL_MAXC && M > 0);
  hipLaunchKernelGGL(dice_probs_bwd_kernel, dim3(gl_grid(M * C)), dim3(256), 0, as_stream(stream), Pm, ld, M, C, lab,
                     ws + 1024l * 3 * C, wgt, g, dP, ldo);
  return arco_launch_status();
}
// unsupervised weighted CE: B images of P pixels; ws >= arco_loss_slabs(B) * 4 * B + B + 1 doubles; out[0] = loss
long arco_loss_slabs(int B) { return unsup_nblk(B); }
int arco_unsup_loss_fwd(const float* X, long ld, int B, long P, int C, const int64_t* lab, const float* conf, float thr,
                        double* ws, float* out, void* stream) {
  ARCO_CHECK_ARG(C >= 1 && C <= GL_MAXC && B > 0 && P > 0);
  const int nblk = unsup_nblk(B);
  hipLaunchKernelGGL(unsup_loss_partial_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), X, ld, P, C, lab, conf, thr, ws);
  hipLaunchKernelGGL(unsup_loss_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), ws, B, nblk, ws + (long)nblk * 4 * B, out);
  return arco_launch_status();
}
int arco_unsup_loss_bwd(const float* X, long ld, int B, long P, int C, const int64_t* lab, const double* ws, const float* g,
                        float* dX, long ldo, void* stream) {
  hipLaunchKernelGGL(unsup_loss_bwd_kernel, dim3(gl_grid((long)B * P)), dim3(256), 0, as_stream(stream), X, ld, P, (long)B * P, C,
                     lab, ws + (long)unsup_nblk(B) * 4 * B, B, g, dX, ldo);
  return arco_launch_status();
}

long arco_sel_state_bytes() { return (long)sizeof(SelState); }
// byte offsets of the two fields a data-parallel run sums over ranks between the phases below
long arco_sel_state_offset(int field) { return field == 0 ? (long)offsetof(SelState, n_valid) : (long)offsetof(SelState, hist); }

// The same selection in phases, so that a data-parallel caller can all-reduce the valid count (after phase 0) and the
// four 256-bin histograms (after every phase 2) across ranks and every rank arrives at the percentiles of the GLOBAL
// batch.  phase 0: clear + count valid, 1: ranks from the count, 2: histogram of digit `pass`,
// 3: pick the bucket of digit `pass`, 4: masks.
int arco_entropy_masks_phase(int phase, int pass, const float* ent, const int64_t* lab_l, const int64_t* lab_u, long n_l,
                             long n_u, double q_lo, double q_hi, void* state, float* low, float* high, void* stream) {
  ARCO_CHECK_ARG(n_u > 0 && state && phase >= 0 && phase <= 4 && pass >= 0 && pass < 4);
  hipStream_t st = as_stream(stream);
  SelState* s = reinterpret_cast<SelState*>(state);
  const unsigned g = gl_grid(n_u) > 256 ? 256 : gl_grid(n_u);
  switch (phase) {
    case 0:
      (void)hipMemsetAsync(s, 0, sizeof(SelState), st);
      hipLaunchKernelGGL(sel_count_kernel, dim3(g), dim3(256), 0, st, lab_u, n_u, s);
      break;
    case 1: hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, st, s, q_lo, q_hi); break;
    case 2: hipLaunchKernelGGL(sel_hist_kernel, dim3(g), dim3(256), 0, st, ent, lab_u, n_u, pass, s); break;
    case 3: hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, st, pass, s); break;
    default:
      hipLaunchKernelGGL(entropy_masks_kernel, dim3(gl_grid(n_l + n_u)), dim3(256), 0, st, ent, lab_l, lab_u, n_l, n_u, s, low, high);
  }
  return arco_launch_status();
}

// masks from the entropy percentiles q_lo / q_hi over pixels with lab_u >= 0; state = arco_sel_state_bytes() scratch
int arco_entropy_masks(const float* ent, const int64_t* lab_l, const int64_t* lab_u, long n_l, long n_u, double q_lo,
                       double q_hi, void* state, float* low, float* high, void* stream) {
  ARCO_CHECK_ARG(n_u > 0 && state);
  hipStream_t st = as_stream(stream);
  SelState* s = reinterpret_cast<SelState*>(state);
  (void)hipMemsetAsync(s, 0, sizeof(SelState), st);
  hipLaunchKernelGGL(sel_count_kernel, dim3(gl_grid(n_u) > 256 ? 256 : gl_grid(n_u)), dim3(256), 0, st, lab_u, n_u, s);
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, st, s, q_lo, q_hi);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(gl_grid(n_u) > 256 ? 256 : gl_grid(n_u)), dim3(256), 0, st, ent, lab_u, n_u, pass, s);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, st, pass, s);
  }
  hipLaunchKernelGGL(entropy_masks_kernel, dim3(gl_grid(n_l + n_u)), dim3(256), 0, st, ent, lab_l, lab_u, n_l, n_u, s,
                     low, high);
  return arco_launch_status();
}

}  // extern "C"
