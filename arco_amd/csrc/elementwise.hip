// HBM-bound channels-last passes around the convs: train-mode BatchNorm statistics /
// apply / backward fused with (Leaky)ReLU and dropout (unetWithArgs.py:36-44,
// vnetWithArgs.py:16-25), 2x2 max-pool, align_corners bilinear resize
// (model_2D.py:43-52, unetWithArgs.py:74-75), channel-slice copies (the cat's),
// flat-buffer SGD-Nesterov and EMA (train_arco_2d.py:248,306-308,432; model_2D.py:176-182).
// All kernels: float4 per lane along the channel axis, grid-stride over pixels.
#include "common.h"
#include "igemm_args.h"

// (pcg_hash / drop_keep: common.h - the consumer-side activation of the convolution loaders draws the same masks)

// ---- BN statistics finalize: block partial (sum, sumsq) -> mean, istd, running stats.
//      One 64-lane wave per channel; fp64 tree over the partial slabs.
// Per-thread partial sums of two slab rows (stride NT): eight independent loads of each row in flight per trip - the
// rows of the shallow levels have 2048-8192 slabs, one load per trip made these few-block kernels a chain of DRAM latencies.
template <int NT>
__device__ __forceinline__ void slab_sums2(const float* __restrict__ pa, const float* __restrict__ pb, int n, double& a, double& b) {
  int i = threadIdx.x;
  for (; i + 7 * NT < n; i += 8 * NT) {
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { va[u] = pa[i + u * NT]; vb[u] = pb[i + u * NT]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
  }
  for (; i < n; i += NT) { a += (double)pa[i]; b += (double)pb[i]; }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ ssum, const float* __restrict__ ssq, int nblk, int C,
                                   double count, float eps, float momentum, float* __restrict__ mean,
                                   float* __restrict__ istd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked, int groups,
                                   int defer_from, float* __restrict__ deferred) {
  // groups > 1: the batch is `groups` independent BN batches (e.g. the labelled and the unlabelled half of one
  // launch); group g owns the slabs [g*nblk/groups, ...), has `count` elements per channel, gets mean/istd row g,
  // and the running statistics receive the groups' updates one after the other (as separate forwards would).
  // One 4-wave block per channel (the 3-D levels have up to 31 360 slabs per channel), fixed summation order.
  // deferred != null: groups g >= defer_from do NOT touch the running statistics; their update terms (mean, unbiased
  // variance) go to deferred[g - defer_from][2][C] (+ a "pending" flag behind them) and are applied later by
  // bn_apply_deferred_kernel - the trainer runs forwards in a different order than the reference, the momentum
  // updates must still land in the reference's order.
  __shared__ double sh[2][4];
  const int c = blockIdx.x, npg = nblk / groups, w = threadIdx.x >> 6;
  if (c == 0 && threadIdx.x == 0 && num_batches_tracked) num_batches_tracked[0] += groups;
  for (int g = 0; g < groups; ++g) {
    double s = 0.0, q = 0.0;
    slab_sums2<256>(ssum + (long)c * nblk + g * npg, ssq + (long)c * nblk + g * npg, npg, s, q);
    s = wave_sum_d(s); q = wave_sum_d(q);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
      s = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]); q = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
      const double m = s / count;
      double var = q / count - m * m;
      if (var < 0.0) var = 0.0;
      mean[(long)g * C + c] = (float)m;
      istd[(long)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        if (deferred && g >= defer_from) {
          deferred[((long)(g - defer_from) * 2 + 0) * C + c] = (float)m;
          deferred[((long)(g - defer_from) * 2 + 1) * C + c] = (float)unb;
          if (c == 0) deferred[(long)(groups - defer_from) * 2 * C] = 1.f;
        } else {
          running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
          running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
        }
      }
    }
  }
}

// ---- postponed running-statistics updates of many BN layers in one launch (one block per layer)
struct BnDeferDesc { float* running_mean; float* running_var; float* deferred; int C, n; float momentum; int pad; };
__global__ void bn_apply_deferred_kernel(const BnDeferDesc* __restrict__ desc) {
  const BnDeferDesc d = desc[blockIdx.x];
  float* flag = d.deferred + (long)d.n * 2 * d.C;
  const bool pending = *flag != 0.f;
  __syncthreads();
  if (!pending) return;
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    double rm = d.running_mean[c], rv = d.running_var[c];
    for (int g = 0; g < d.n; ++g) {
      rm = (float)((1.0 - d.momentum) * rm + d.momentum * (double)d.deferred[((long)g * 2 + 0) * d.C + c]);
      rv = (float)((1.0 - d.momentum) * rv + d.momentum * (double)d.deferred[((long)g * 2 + 1) * d.C + c]);
    }
    d.running_mean[c] = (float)rm; d.running_var[c] = (float)rv;
  }
  if (threadIdx.x == 0) *flag = 0.f;
}

// ---- generic per-channel (sum, sumsq) partials of a channels-last tensor (for
//      tensors not produced by the conv kernel, e.g. the V-Net k2s2 convs)
// Block-level per-channel totals of (s, q): thread (tq = tid % q4, tr = tid / q4) holds partial sums of the
// channel quad tq.  Power-of-two q4 <= 64: the lanes of a wave that share a quad differ only in lane bits >= q4 ->
// xor-shuffle tree, then a 4-wave sum through LDS (the serial rstep-long loop it replaces dominated the kernel
// for C = 16: 64 dependent LDS reads by 16 threads).  Other q4: the serial loop.
__device__ __forceinline__ void block_channel_totals(f32x4 s, f32x4 q, int C, float* red /*>= 2048 floats*/,
                                                     float* __restrict__ out_s, float* __restrict__ out_q, int nblk) {
  const int q4 = C / 4, rstep = 256 / q4;
  if ((q4 & (q4 - 1)) == 0 && q4 <= 64) {
    for (int o = q4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += __shfl_xor(s[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < q4) {
      *reinterpret_cast<f32x4*>(&red[w * C + 4 * lane]) = s;
      *reinterpret_cast<f32x4*>(&red[1024 + w * C + 4 * lane]) = q;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      out_s[(long)c * nblk + blockIdx.x] = (red[c] + red[C + c]) + (red[2 * C + c] + red[3 * C + c]);
      out_q[(long)c * nblk + blockIdx.x] = (red[1024 + c] + red[1024 + C + c]) + (red[1024 + 2 * C + c] + red[1024 + 3 * C + c]);
    }
    return;
  }
  *reinterpret_cast<f32x4*>(&red[threadIdx.x * 4]) = s;
  *reinterpret_cast<f32x4*>(&red[1024 + threadIdx.x * 4]) = q;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int qq = c / 4, e = c % 4;
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rstep; ++r) { a += red[(r * q4 + qq) * 4 + e]; b += red[1024 + (r * q4 + qq) * 4 + e]; }
    out_s[(long)c * nblk + blockIdx.x] = a; out_q[(long)c * nblk + blockIdx.x] = b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void chan_stats_kernel(const T* __restrict__ X, long ldx, long M, int C,
                                                        float* __restrict__ ssum, float* __restrict__ ssq, int nblk) {
  const int q4 = C / 4, tq = threadIdx.x % q4, tr = threadIdx.x / q4, rstep = 256 / q4;
  // blockIdx.y = BN group: rows [g*M, (g+1)*M), slabs [g*nblk, (g+1)*nblk) of the gridDim.y*nblk slabs per channel
  X += (long)blockIdx.y * M * ldx;
  const long rpb = (M + nblk - 1) / nblk;
  const long r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (tr < rstep)
    for (long r = r0 + tr; r < r1; r += rstep) {
      const f32x4 v = ld4f(X + r * ldx + 4 * tq);
      s += v; q += v * v;
    }
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][256][4]
  block_channel_totals(s, q, C, red, ssum + (long)blockIdx.y * nblk, ssq + (long)blockIdx.y * nblk, nblk * gridDim.y);
}

// depth-to-space folded into the BatchNorm passes of UpsamplingDeconvBlock (vnetWithArgs.py:94-118): the GEMM form of the k2 s2 transposed
// conv leaves Y[(n, x, y, z)][tap * C + c]; read as rows (voxel, tap) of C channels it IS the pre-activation, only in another row order -
// statistics do not care, and the apply pass writes row (voxel, tap) to voxel (n, 2x+dx, 2y+dy, 2z+dz) of the activation (the backward
// reads dA from there): no separate depth-to-space / space-to-depth pass in either direction.  tap = dx*4 + dy*2 + dz as in s2d3_kernel.
struct D2S { int X2, Y2, Z2; };
__device__ __forceinline__ long d2s_row(long r, const D2S& d) {
  const long q = r >> 3; const int tap = (int)(r & 7);
  const int z = (int)(q % d.Z2); long t = q / d.Z2; const int y = (int)(t % d.Y2); t /= d.Y2; const int x = (int)(t % d.X2); const long n = t / d.X2;
  return ((n * (2 * d.X2) + 2 * x + (tap >> 2)) * (2 * d.Y2) + 2 * y + ((tap >> 1) & 1)) * (long)(2 * d.Z2) + 2 * z + (tap & 1);
}
// the inverse: activation voxel row -> (voxel, tap) row of Y.  The forward apply walks the OUTPUT rows (contiguous stores, gathered loads:
// scattered 8- / 16-byte stores measured slower than the depth-to-space pass they replace on f16 rows)
__device__ __forceinline__ long d2s_src_row(long ro, const D2S& d) {
  const int zf = (int)(ro % (2 * d.Z2)); long t = ro / (2 * d.Z2); const int yf = (int)(t % (2 * d.Y2)); t /= 2 * d.Y2;
  const int xf = (int)(t % (2 * d.X2)); const long n = t / (2 * d.X2);
  const int tap = (xf & 1) * 4 + (yf & 1) * 2 + (zf & 1);
  return ((((n * d.X2 + (xf >> 1)) * d.Y2 + (yf >> 1)) * (long)d.Z2 + (zf >> 1)) << 3) + tap;
}

// ---- BN apply + LeakyReLU(slope) + dropout:  a = drop(lrelu((z-mean)*istd*gamma+beta))
//      drop_mode 0: none, 1: per element (nn.Dropout), 2: per (image, channel) (nn.Dropout3d)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ Z, long ldz, long M, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ istd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float slope, int drop_mode, float p, uint64_t seed_, long P,
                                                        T* __restrict__ Aout, long lda,
                                                        const uint64_t* __restrict__ seed_dev,
                                                        const T* __restrict__ R = nullptr, long ldr = 0, D2S ds = D2S{0, 0, 0}) {
  // R != nullptr: a = drop(lrelu(BN(z))) + R  (the skip addition of the V-Net decoder, vnetWithArgs.py:224-236, in the apply pass)
  // ds.X2 != 0: the loop runs over the rows (voxels) of A / R, the depth-to-space'd tensor; Z is gathered from the (voxel, tap) rows of the
  // GEMM-form transposed conv (d2s_src_row)
  const int q4 = C / 4;
  // blockIdx.y = BN group: rows [g*M, (g+1)*M) of the tensor with parameter row g (M = rows per group)
  const long row0 = (long)blockIdx.y * M;
  const T* const Z0 = Z;
  Z += row0 * ldz; Aout += row0 * lda;
  if (R) R += row0 * ldr;
  if (mean) { mean += (long)blockIdx.y * C; istd += (long)blockIdx.y * C; }
  const long tot = M * q4;
  const float keep_scale = drop_mode ? 1.0f / (1.0f - p) : 1.0f;
  // graph-captured forwards read a per-replay salt from device memory (fresh masks on every replay)
  const uint64_t seed = seed_dev ? seed_ ^ (seed_dev[0] * 0x9E3779B97F4A7C15ull) : seed_;
  if constexpr (sizeof(T) == 2) {
    // f16 storage: EIGHT channels per thread - one 16-byte access per row where the quad form moves 8 bytes (round 4: the f16
    // apply passes ran at 3.3 TB/s against the fp32 ones' 4-5; same arithmetic per element, results identical)
    const int q8 = C / 8;
    if (mean && (C & 7) == 0 && 256 % q8 == 0 && ((ldz | lda | ldr) & 7) == 0 && al16(Z0) && al16(Aout) && al16(R)) {
      const long gt = (long)blockIdx.x * 256 + threadIdx.x, rstride = (long)gridDim.x * 256 / q8;
      const int c = (int)(gt % q8) * 8;
      const f32x8 mu = ld8p(mean + c), is = ld8p(istd + c), ga = ld8p(gamma + c), be = ld8p(beta + c);
      for (long rb = gt / q8; rb < M; rb += 4 * rstride) {
        f32x8 z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long r = rb + u * rstride;
          if (r < M) z[u] = ld8f(ds.X2 ? Z0 + d2s_src_row(r + row0, ds) * ldz + c : Z + r * ldz + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long r = rb + u * rstride;
          if (r < M) {
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y = (z[u][e] - mu[e]) * is[e] * ga[e] + be[e];
              y = y >= 0.f ? y : y * slope;
              if (drop_mode == 1) y = drop_keep(seed, (uint64_t)((r + row0) * C + c + e), p) ? y * keep_scale : 0.f;
              else if (drop_mode == 2) y = drop_keep(seed, (uint64_t)(((r + row0) / P) * C + c + e), p) ? y * keep_scale : 0.f;
              o[e] = y;
            }
            if (R) o += ld8f(R + r * ldr + c);
            st8f(Aout + r * lda + c, o);
          }
        }
      }
      return;
    }
  }
  if (mean && 256 % q4 == 0) {
    // a thread keeps ONE channel quad for the whole sweep (the grid stride is a multiple of q4): the four
    // per-channel parameters are loaded once, and four rows are in flight per trip
    const long gt = (long)blockIdx.x * 256 + threadIdx.x, rstride = (long)gridDim.x * 256 / q4;
    const int c = (int)(gt % q4) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(istd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    for (long rb = gt / q4; rb < M; rb += 4 * rstride) {
      f32x4 z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + u * rstride;
        if (r < M) z[u] = ld4f(ds.X2 ? Z0 + d2s_src_row(r + row0, ds) * ldz + c : Z + r * ldz + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + u * rstride;
        if (r < M) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float y = (z[u][e] - mu[e]) * is[e] * ga[e] + be[e];
            y = y >= 0.f ? y : y * slope;
            if (drop_mode == 1) y = drop_keep(seed, (uint64_t)((r + row0) * C + c + e), p) ? y * keep_scale : 0.f;
            else if (drop_mode == 2) y = drop_keep(seed, (uint64_t)(((r + row0) / P) * C + c + e), p) ? y * keep_scale : 0.f;
            o[e] = y;
          }
          if (R) o += ld4f(R + r * ldr + c);
          st4f(Aout + r * lda + c, o);
        }
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long r = i / q4; const int c = (int)(i - r * q4) * 4;
    const f32x4 z = ld4f(ds.X2 ? Z0 + d2s_src_row(r + row0, ds) * ldz + c : Z + r * ldz + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y = mean ? (z[e] - mean[c + e]) * istd[c + e] * gamma[c + e] + beta[c + e] : z[e];
      y = y >= 0.f ? y : y * slope;
      if (drop_mode == 1) y = drop_keep(seed, (uint64_t)((r + row0) * C + c + e), p) ? y * keep_scale : 0.f;
      else if (drop_mode == 2) y = drop_keep(seed, (uint64_t)(((r + row0) / P) * C + c + e), p) ? y * keep_scale : 0.f;
      o[e] = y;
    }
    if (R) o += ld4f(R + r * ldr + c);
    st4f(Aout + r * lda + c, o);
  }
}

// dy (gradient at the BN output) recomputed from z: dy = dA * dropmask * lrelu'(y)
__device__ __forceinline__ float bn_dy(float dA, float y, float slope, int drop_mode, float p, float keep_scale,
                                       uint64_t seed, uint64_t e) {
  float d = y >= 0.f ? dA : dA * slope;
  if (drop_mode) d = drop_keep(seed, e, p) ? d * keep_scale : 0.f;
  return d;
}

// ---- BN backward pass 1: per-channel partial sums of dy and dy*xhat
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(
    const T* __restrict__ dA, long ldd, const T* __restrict__ Z, long ldz, long M, int C,
    const float* __restrict__ mean, const float* __restrict__ istd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float slope, int drop_mode, float p, uint64_t seed_, long P,
    float* __restrict__ s_dy, float* __restrict__ s_dyx, int nblk, const uint64_t* __restrict__ seed_dev, D2S ds = D2S{0, 0, 0}) {
  const uint64_t seed = seed_dev ? seed_ ^ (seed_dev[0] * 0x9E3779B97F4A7C15ull) : seed_;
  // blockIdx.y = BN group (rows [g*M, (g+1)*M), parameter row g, slab block [g][C][nblk])
  const long row0 = (long)blockIdx.y * M;
  const T* const dA0 = dA;          // ds.X2 != 0: dA rows are voxels of the depth-to-space'd tensor (see D2S)
  dA += row0 * ldd; Z += row0 * ldz;
  mean += (long)blockIdx.y * C; istd += (long)blockIdx.y * C;
  s_dy += (long)blockIdx.y * C * nblk; s_dyx += (long)blockIdx.y * C * nblk;
  const int q4 = C / 4, tq = threadIdx.x % q4, tr = threadIdx.x / q4, rstep = 256 / q4;
  const long rpb = (M + nblk - 1) / nblk;
  const long r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
  const float keep_scale = drop_mode ? 1.0f / (1.0f - p) : 1.0f;
  f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  const int c = 4 * tq;
  if (tr < rstep) {
    f32x4 mu, is, ga, be;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = mean[c + e]; is[e] = istd[c + e]; ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
    // 4 rows per trip: 8 independent 16-byte loads in flight per thread
    for (long rb = r0 + tr; rb < r1; rb += 4l * rstep) {
      f32x4 z[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + (long)u * rstep;
        if (r < r1) {
          z[u] = ld4f(Z + r * ldz + c);
          d[u] = ld4f(ds.X2 ? dA0 + d2s_row(r + row0, ds) * ldd + c : dA + r * ldd + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + (long)u * rstep;
        if (r < r1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (z[u][e] - mu[e]) * is[e];
            const float y = xh * ga[e] + be[e];
            const uint64_t ei = drop_mode == 2 ? (uint64_t)(((r + row0) / P) * C + c + e) : (uint64_t)((r + row0) * C + c + e);
            const float dy = bn_dy(d[u][e], y, slope, drop_mode, p, keep_scale, seed, ei);
            s[e] += dy; q[e] += dy * xh;
          }
        }
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) float red[];
  block_channel_totals(s, q, C, red, s_dy, s_dyx, nblk);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ s_dy, const float* __restrict__ s_dyx, int nblk, int C,
                                       float* __restrict__ sums /*[G][2][C]*/,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int groups) {
  __shared__ double sh[2][4];
  const int c = blockIdx.x, w = threadIdx.x >> 6;
  double ga = 0.0, gb = 0.0;
  for (int g = 0; g < groups; ++g) {
    double a = 0.0, b = 0.0;
    slab_sums2<256>(s_dy + ((long)g * C + c) * nblk, s_dyx + ((long)g * C + c) * nblk, nblk, a, b);
    a = wave_sum_d(a); b = wave_sum_d(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; }
    __syncthreads();
    a = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]); b = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    if (threadIdx.x == 0) { sums[(2l * g) * C + c] = (float)a; sums[(2l * g + 1) * C + c] = (float)b; }   // this group's sums (apply pass)
    if (groups == 1) { ga = a; gb = b; } else { ga += (double)(float)a; gb += (double)(float)b; }   // = two separate backward passes
  }
  if (threadIdx.x != 0) return;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)ga : (float)ga;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)gb : (float)gb;
}

// ---- BN backward pass 2: dz = gamma*istd*(dy - sum_dy/n - xhat*sum_dyx/n)
//      sums are read from `sums` = [dbeta_this_call | dgamma_this_call] (not the accumulated grads)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const T* __restrict__ dA, long ldd, const T* __restrict__ Z, long ldz, long M, int C,
    const float* __restrict__ mean, const float* __restrict__ istd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float slope, int drop_mode, float p, uint64_t seed_, long P,
    const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx, float inv_count,
    T* __restrict__ dZ, long ldo, const uint64_t* __restrict__ seed_dev, int gn, D2S ds = D2S{0, 0, 0}) {
  // gn != 0 (GroupNorm / InstanceNorm: statistics per sample over a SET of channels): sum_dy / sum_dyx hold the set's sums
  // of gamma*dy and gamma*dy*xhat, and dz = istd * (gamma*dy - sum_dy/n - xhat*sum_dyx/n); gamma varies inside a set
  const uint64_t seed = seed_dev ? seed_ ^ (seed_dev[0] * 0x9E3779B97F4A7C15ull) : seed_;
  const long row0 = (long)blockIdx.y * M;                 // blockIdx.y = BN group
  const T* const dA0 = dA;          // ds.X2 != 0: dA rows are voxels of the depth-to-space'd tensor (see D2S)
  dA += row0 * ldd; Z += row0 * ldz; dZ += row0 * ldo;
  if (mean) { mean += (long)blockIdx.y * C; istd += (long)blockIdx.y * C; sum_dy += 2l * blockIdx.y * C; sum_dyx += 2l * blockIdx.y * C; }
  const int q4 = C / 4;
  const long tot = M * q4;
  const float keep_scale = drop_mode ? 1.0f / (1.0f - p) : 1.0f;
  if constexpr (sizeof(T) == 2) {       // f16 storage: eight channels per thread, 16-byte accesses (see bn_act_fwd_kernel)
    const int q8 = C / 8;
    if (mean && (C & 7) == 0 && 256 % q8 == 0 && ((ldd | ldz | ldo) & 7) == 0 && al16(dA0) && al16(Z) && al16(dZ)) {
      const long gt = (long)blockIdx.x * 256 + threadIdx.x, rstride = (long)gridDim.x * 256 / q8;
      const int c = (int)(gt % q8) * 8;
      const f32x8 mu = ld8p(mean + c), is = ld8p(istd + c), ga = ld8p(gamma + c), be = ld8p(beta + c);
      const f32x8 sd = ld8p(sum_dy + c), sx = ld8p(sum_dyx + c);
      for (long rb = gt / q8; rb < M; rb += 2 * rstride) {
        f32x8 z[2], d[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const long r = rb + u * rstride;
          if (r < M) { z[u] = ld8f(Z + r * ldz + c); d[u] = ld8f(ds.X2 ? dA0 + d2s_row(r + row0, ds) * ldd + c : dA + r * ldd + c); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const long r = rb + u * rstride;
          if (r < M) {
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint64_t ei = drop_mode == 2 ? (uint64_t)(((r + row0) / P) * C + c + e) : (uint64_t)((r + row0) * C + c + e);
              const float xh = (z[u][e] - mu[e]) * is[e];
              const float y = xh * ga[e] + be[e];
              const float dy = bn_dy(d[u][e], y, slope, drop_mode, p, keep_scale, seed, ei);
              o[e] = gn ? is[e] * (ga[e] * dy - sd[e] * inv_count - xh * sx[e] * inv_count)
                        : ga[e] * is[e] * (dy - sd[e] * inv_count - xh * sx[e] * inv_count);
            }
            st8f(dZ + r * ldo + c, o);
          }
        }
      }
      return;
    }
  }
  if (mean && 256 % q4 == 0) {          // one channel quad per thread, parameters hoisted, two rows in flight
    const long gt = (long)blockIdx.x * 256 + threadIdx.x, rstride = (long)gridDim.x * 256 / q4;
    const int c = (int)(gt % q4) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(istd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 sd = *reinterpret_cast<const f32x4*>(sum_dy + c), sx = *reinterpret_cast<const f32x4*>(sum_dyx + c);
    for (long rb = gt / q4; rb < M; rb += 2 * rstride) {
      f32x4 z[2], d[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long r = rb + u * rstride;
        if (r < M) { z[u] = ld4f(Z + r * ldz + c); d[u] = ld4f(ds.X2 ? dA0 + d2s_row(r + row0, ds) * ldd + c : dA + r * ldd + c); }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long r = rb + u * rstride;
        if (r < M) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint64_t ei = drop_mode == 2 ? (uint64_t)(((r + row0) / P) * C + c + e) : (uint64_t)((r + row0) * C + c + e);
            const float xh = (z[u][e] - mu[e]) * is[e];
            const float y = xh * ga[e] + be[e];
            const float dy = bn_dy(d[u][e], y, slope, drop_mode, p, keep_scale, seed, ei);
            o[e] = gn ? is[e] * (ga[e] * dy - sd[e] * inv_count - xh * sx[e] * inv_count)
                      : ga[e] * is[e] * (dy - sd[e] * inv_count - xh * sx[e] * inv_count);
          }
          st4f(dZ + r * ldo + c, o);
        }
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long r = i / q4; const int c = (int)(i - r * q4) * 4;
    const f32x4 z = ld4f(Z + r * ldz + c);
    const f32x4 d = ld4f(ds.X2 ? dA0 + d2s_row(r + row0, ds) * ldd + c : dA + r * ldd + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint64_t ei = drop_mode == 2 ? (uint64_t)(((r + row0) / P) * C + c + e) : (uint64_t)((r + row0) * C + c + e);
      if (mean) {
        const float xh = (z[e] - mean[c + e]) * istd[c + e];
        const float y = xh * gamma[c + e] + beta[c + e];
        const float dy = bn_dy(d[e], y, slope, drop_mode, p, keep_scale, seed, ei);
        o[e] = gn ? istd[c + e] * (gamma[c + e] * dy - sum_dy[c + e] * inv_count - xh * sum_dyx[c + e] * inv_count)
                  : gamma[c + e] * istd[c + e] * (dy - sum_dy[c + e] * inv_count - xh * sum_dyx[c + e] * inv_count);
      } else {
        o[e] = bn_dy(d[e], z[e], slope, drop_mode, p, keep_scale, seed, ei);
      }
    }
    st4f(dZ + r * ldo + c, o);
  }
}

// ---- 2x2 max pool (nn.MaxPool2d(2), unetWithArgs.py:55-58), channels-last
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ X, long ldx, int NB, int H, int W,
                                                          int C, float* __restrict__ Y, long ldy) {
  const int q4 = C / 4, Ho = H / 2, Wo = W / 2;
  const long tot = (long)NB * Ho * Wo * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xo = r % Wo; r /= Wo; const int yo = r % Ho; const long n = r / Ho;
    const float* b = X + (((n * H) + 2 * yo) * W + 2 * xo) * ldx + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b), v01 = *reinterpret_cast<const f32x4*>(b + ldx);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + (long)W * ldx), v11 = *reinterpret_cast<const f32x4*>(b + (long)W * ldx + ldx);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
    *reinterpret_cast<f32x4*>(Y + (((n * Ho) + yo) * Wo + xo) * ldy + c) = o;
  }
}
// BN apply + LeakyReLU of a block's last stage AND the 2x2 max-pool of the result in one pass (the ConvBlock output feeds
// the next DownBlock's nn.MaxPool2d and, unpooled, the decoder's skip: unetWithArgs.py:55-58,109-116): a thread owns a
// 2x2 pixel window x 4 channels - four Z loads, four A stores, one pooled store; the separate pooling pass re-read A.
// Same arithmetic per element as bn_act_fwd_kernel (no dropout on this stage) / maxpool2_fwd_kernel.
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const float* __restrict__ Z, long ldz, int NB, int H, int W, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ istd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float slope, float* __restrict__ Aout, long lda,
                                                             float* __restrict__ Pout, long ldp, int img_per_group) {
  const int q4 = C / 4, Ho = H / 2, Wo = W / 2;
  const long tot = (long)NB * Ho * Wo * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xo = r % Wo; r /= Wo; const int yo = r % Ho; const long n = r / Ho;
    const int g = (int)(n / img_per_group);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (long)g * C + c), is = *reinterpret_cast<const f32x4*>(istd + (long)g * C + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    const long row = ((n * H) + 2 * yo) * W + 2 * xo;
    const long rows[4] = {row, row + 1, row + W, row + W + 1};
    f32x4 z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) z[u] = *reinterpret_cast<const f32x4*>(Z + rows[u] * ldz + c);
    f32x4 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = (z[u][e] - mu[e]) * is[e] * ga[e] + be[e];
        o[u][e] = y >= 0.f ? y : y * slope;
      }
      *reinterpret_cast<f32x4*>(Aout + rows[u] * lda + c) = o[u];
    }
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(o[0][e], o[1][e]), fmaxf(o[2][e], o[3][e]));
    *reinterpret_cast<f32x4*>(Pout + (((n * Ho) + yo) * Wo + xo) * ldp + c) = m;
  }
}
// gradient goes to the first maximum in window scan order (torch's saved argmax)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ X, long ldx, int NB, int H, int W,
                                                          int C, const float* __restrict__ dY, long ldy,
                                                          float* __restrict__ dX, long ldo,
                                                          const float* __restrict__ add, long lda) {
  const int q4 = C / 4, Ho = H / 2, Wo = W / 2;
  const long tot = (long)NB * Ho * Wo * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xo = r % Wo; r /= Wo; const int yo = r % Ho; const long n = r / Ho;
    const long p00 = ((n * H) + 2 * yo) * W + 2 * xo;
    const long off[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(X + off[k] * ldx + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dY + (((n * Ho) + yo) * Wo + xo) * ldy + c);
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int best = 0; float bv = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k) if (v[k][e] > bv) { bv = v[k][e]; best = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] = k == best ? g[e] : 0.f;
    }
    if (add) {           // + the gradient of the other consumer of x (the decoder's skip connection): no separate add kernel
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += *reinterpret_cast<const f32x4*>(add + off[k] * lda + c);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dX + off[k] * ldo + c) = o[k];
  }
}

// ---- bilinear resize, align_corners=True (torch upsample_bilinear2d index math in fp32)
// (ac_src: igemm_args.h)
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ X, long ldx, int NB, int Hi, int Wi,
                                                          int C, int Ho, int Wo, float* __restrict__ Y, long ldy) {
  const int q4 = C / 4;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long tot = (long)NB * Ho * Wo * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xo = r % Wo; r /= Wo; const int yo = r % Ho; const long n = r / Ho;
    int y0, y1, x0, x1; float ly, lx;
    ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = X + (n * Hi) * (long)Wi * ldx + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((long)y0 * Wi + x0) * ldx);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((long)y0 * Wi + x1) * ldx);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((long)y1 * Wi + x0) * ldx);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((long)y1 * Wi + x1) * ldx);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(Y + (((n * Ho) + yo) * Wo + xo) * ldy + c) = o;
  }
}
// adjoint as a gather (no atomics): input pixel (yi, xi) collects from the output pixels that
// reference it; candidate output ranges are bounded by the scale and verified with ac_src.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dY, long ldy, int NB, int Hi, int Wi,
                                                          int C, int Ho, int Wo, float* __restrict__ dX, long ldx,
                                                          int accumulate) {
  const int q4 = C / 4;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const float ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
  const long tot = (long)NB * Hi * Wi * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xi = r % Wi; r /= Wi; const int yi = r % Hi; const long n = r / Hi;
    int ya = sh > 0.f ? (int)floorf((float)(yi - 1) * ish) - 1 : 0, yb = sh > 0.f ? (int)ceilf((float)(yi + 1) * ish) + 1 : Ho - 1;
    int xa = sw > 0.f ? (int)floorf((float)(xi - 1) * isw) - 1 : 0, xb = sw > 0.f ? (int)ceilf((float)(xi + 1) * isw) + 1 : Wo - 1;
    ya = max(ya, 0); yb = min(yb, Ho - 1); xa = max(xa, 0); xb = min(xb, Wo - 1);
    f32x4 acc = {0, 0, 0, 0};
    for (int yo = ya; yo <= yb; ++yo) {
      int y0, y1; float ly; ac_src(yo, sh, Hi, y0, y1, ly);
      float wy = 0.f;
      if (y0 == yi) wy += 1.f - ly;
      if (y1 == yi) wy += ly;
      if (wy == 0.f && !(y0 == yi || y1 == yi)) continue;
      for (int xo = xa; xo <= xb; ++xo) {
        int x0, x1; float lx; ac_src(xo, sw, Wi, x0, x1, lx);
        float wx = 0.f;
        if (x0 == xi) wx += 1.f - lx;
        if (x1 == xi) wx += lx;
        if (!(x0 == xi || x1 == xi)) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dY + (((n * Ho) + yo) * Wo + xo) * ldy + c);
        const float w = wy * wx;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * g[e];
      }
    }
    float* dst = dX + (((n * Hi) + yi) * Wi + xi) * ldx + c;
    if (accumulate) { const f32x4 old = *reinterpret_cast<const f32x4*>(dst); acc += old; }
    *reinterpret_cast<f32x4*>(dst) = acc;
  }
}

// ---- anchor-row gather of cat(bilinear_up(lo), hi) and its adjoint (row-sparse student head).
// For anchor j with high-res pixel id pix[j] = (n, y, x):  X[j][0..Clo) = align_corners bilinear
// sample of `lo` at (y, x) (same fp32 index math / lerp order as bilinear_fwd_kernel, so rows are
// bit-identical to the dense upsample+cat), X[j][Clo..Clo+Chi) = hi[pix[j]].
__global__ __launch_bounds__(256) void gather_upcat_rows_kernel(const float* __restrict__ lo, long ldlo, int Clo, int Hi, int Wi,
                                                               const float* __restrict__ hi, long ldhi, int Chi, int Ho, int Wo,
                                                               const int64_t* __restrict__ pix, long n, float* __restrict__ X, long ldx) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const long p = pix[j];
  const long img = p / ((long)Ho * Wo); const int rem = (int)(p - img * (long)Ho * Wo);
  const int yo = rem / Wo, xo = rem - yo * Wo;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int y0, y1, x0, x1; float ly, lx;
  ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* b = lo + (img * Hi) * (long)Wi * ldlo;
  const float* r00 = b + ((long)y0 * Wi + x0) * ldlo; const float* r01 = b + ((long)y0 * Wi + x1) * ldlo;
  const float* r10 = b + ((long)y1 * Wi + x0) * ldlo; const float* r11 = b + ((long)y1 * Wi + x1) * ldlo;
  float* o = X + j * ldx;
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(r00 + c), v01 = *reinterpret_cast<const f32x4*>(r01 + c);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(r10 + c), v11 = *reinterpret_cast<const f32x4*>(r11 + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(o + c) = r;
  }
  const float* h = hi + p * ldhi;
  for (int c = lane * 4; c < Chi; c += 256) *reinterpret_cast<f32x4*>(o + Clo + c) = *reinterpret_cast<const f32x4*>(h + c);
}
__global__ __launch_bounds__(256) void scatter_upcat_rows_kernel(const float* __restrict__ dX, long ldx, const int64_t* __restrict__ pix, long n,
                                                                float* __restrict__ dlo, long ldlo, int Clo, int Hi, int Wi,
                                                                float* __restrict__ dhi, long ldhi, int Chi, int Ho, int Wo) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const long p = pix[j];
  const long img = p / ((long)Ho * Wo); const int rem = (int)(p - img * (long)Ho * Wo);
  const int yo = rem / Wo, xo = rem - yo * Wo;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int y0, y1, x0, x1; float ly, lx;
  ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  float* b = dlo + (img * Hi) * (long)Wi * ldlo;
  float* r00 = b + ((long)y0 * Wi + x0) * ldlo; float* r01 = b + ((long)y0 * Wi + x1) * ldlo;
  float* r10 = b + ((long)y1 * Wi + x0) * ldlo; float* r11 = b + ((long)y1 * Wi + x1) * ldlo;
  const float* g = dX + j * ldx;
  for (int c = lane; c < Clo; c += 64) {
    const float v = g[c];
    atomicAdd(r00 + c, hy * hx * v); atomicAdd(r01 + c, hy * lx * v);
    atomicAdd(r10 + c, ly * hx * v); atomicAdd(r11 + c, ly * lx * v);
  }
  float* h = dhi + p * ldhi;
  for (int c = lane; c < Chi; c += 64) atomicAdd(h + c, g[Clo + c]);
}

// ---- second level of the row-sparse head: the 4 low-res neighbours of each anchor as explicit rows.
// nb4[4j+t] = low-res pixel id of neighbour t (00,01,10,11) of high-res pixel pix[j]; lylx[j] = (ly, lx)
__global__ void up_neighbors_kernel(const int64_t* __restrict__ pix, long n, int Hi, int Wi, int Ho, int Wo,
                                    int64_t* __restrict__ nb4, float* __restrict__ lylx) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long p = pix[j];
  const long img = p / ((long)Ho * Wo); const int rem = (int)(p - img * (long)Ho * Wo);
  const int yo = rem / Wo, xo = rem - yo * Wo;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int y0, y1, x0, x1; float ly, lx;
  ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
  const long b = img * (long)Hi * Wi;
  nb4[4 * j + 0] = b + (long)y0 * Wi + x0; nb4[4 * j + 1] = b + (long)y0 * Wi + x1;
  nb4[4 * j + 2] = b + (long)y1 * Wi + x0; nb4[4 * j + 3] = b + (long)y1 * Wi + x1;
  lylx[2 * j] = ly; lylx[2 * j + 1] = lx;
}
// X[j][0..Clo) = hy*(hx*V[4j]+lx*V[4j+1]) + ly*(hx*V[4j+2]+lx*V[4j+3])  (same order as bilinear_fwd_kernel);
// X[j][Clo..Clo+Chi) = hi[pix[j]]
__global__ __launch_bounds__(256) void lerp4_cat_rows_kernel(const float* __restrict__ V, long ldv, int Clo,
                                                            const float* __restrict__ lylx, const float* __restrict__ hi,
                                                            long ldhi, int Chi, const int64_t* __restrict__ pix, long n,
                                                            float* __restrict__ X, long ldx) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float ly = lylx[2 * j], lx = lylx[2 * j + 1], hy = 1.f - ly, hx = 1.f - lx;
  const float* v0 = V + (4 * j) * ldv;
  float* o = X + j * ldx;
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(v0 + c), v01 = *reinterpret_cast<const f32x4*>(v0 + ldv + c);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(v0 + 2 * ldv + c), v11 = *reinterpret_cast<const f32x4*>(v0 + 3 * ldv + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(o + c) = r;
  }
  const float* h = hi + pix[j] * ldhi;
  for (int c = lane * 4; c < Chi; c += 256) *reinterpret_cast<f32x4*>(o + Clo + c) = *reinterpret_cast<const f32x4*>(h + c);
}
// adjoint: dV[4j+t] = w_t * dX[j][0..Clo) ; dhi[pix[j]] += dX[j][Clo..)
__global__ __launch_bounds__(256) void lerp4_cat_rows_bwd_kernel(const float* __restrict__ dX, long ldx, int Clo,
                                                                const float* __restrict__ lylx, const int64_t* __restrict__ pix,
                                                                long n, float* __restrict__ dV, long ldv,
                                                                float* __restrict__ dhi, long ldhi, int Chi) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float ly = lylx[2 * j], lx = lylx[2 * j + 1], hy = 1.f - ly, hx = 1.f - lx;
  const float* g = dX + j * ldx;
  float* v0 = dV + (4 * j) * ldv;
  // The two MIXED weights are single v_mul_f32 instructions on purpose.  Left to the compiler they became
  // `v_pk_mul_f32 v[16:17], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]` - a packed-fp32 instruction whose LOW result takes
  // the HIGH dword of src1 - and gfx950 computes that low result with src1.hi read as 0 in lanes 48-63 whenever another wave of
  // the SIMD is executing a K=32 MFMA (v_mfma_f32_16x16x32_bf16 / _f16): the non-reproducible gradient of round 4 (this kernel
  // beside the warped pass's backward graph on the second queue; torch-free reproducer tools/debug/pkmul_repro.hip, form sweep
  // tools/debug/pkmul_sweep.hip, profiles/r05_notes.md section 1).  tests/test_isa_lint.py keeps the form out of the library.
  const float w00 = hy * hx, w11 = ly * lx;
  float w01, w10;
  asm("v_mul_f32 %0, %1, %2" : "=v"(w01) : "v"(hy), "v"(lx));
  asm("v_mul_f32 %0, %1, %2" : "=v"(w10) : "v"(ly), "v"(hx));
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(g + c);
    *reinterpret_cast<f32x4*>(v0 + c) = d * w00;
    *reinterpret_cast<f32x4*>(v0 + ldv + c) = d * w01;
    *reinterpret_cast<f32x4*>(v0 + 2 * ldv + c) = d * w10;
    *reinterpret_cast<f32x4*>(v0 + 3 * ldv + c) = d * w11;
  }
  float* h = dhi + pix[j] * ldhi;
  for (int c = lane; c < Chi; c += 64) atomicAdd(h + c, g[Clo + c]);
}

// ---- 3-D: space-to-depth / depth-to-space (factor 2) for the k2s2 (transposed) convs of the V-Net
// (vnetWithArgs.py:67-118): a k=2,s=2 Conv3d is a GEMM over the 8*C channels of the packed tensor.
// dir 0: P[q][tap*C + c] = V[(n, 2x+dx, 2y+dy, 2z+dz)][c] ; dir 1: the inverse scatter.  tap = dx*4+dy*2+dz
__global__ __launch_bounds__(256) void s2d3_kernel(float* __restrict__ V, long ldv, int NV, int X2, int Y2, int Z2, int C,
                                                  float* __restrict__ P, long ldp, int dir) {
  const int q4 = C / 4;
  const long tot = (long)NV * X2 * Y2 * Z2 * 8 * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int tap = r % 8; r /= 8;
    const int z = r % Z2; r /= Z2; const int y = r % Y2; r /= Y2; const int x = r % X2; const long n = r / X2;
    const long q = ((n * X2 + x) * Y2 + y) * (long)Z2 + z;
    const long v = (((n * (2 * X2) + 2 * x + (tap >> 2)) * (2 * Y2) + 2 * y + ((tap >> 1) & 1)) * (long)(2 * Z2) + 2 * z + (tap & 1));
    float* pv = V + v * ldv + c; float* pp = P + q * ldp + tap * C + c;
    if (dir == 0) *reinterpret_cast<f32x4*>(pp) = *reinterpret_cast<const f32x4*>(pv);
    else *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(pp);
  }
}

// V = depth_to_space(P) + ADD in one pass: the gradient of an activation that feeds BOTH a DownsamplingConvBlock (through space-to-depth)
// and the decoder's skip connection (vnetWithArgs.py:186-201,224-236) - instead of the inverse permutation followed by an add
template <typename T>
__global__ __launch_bounds__(256) void d2s3_add_kernel(const T* __restrict__ P, long ldp, int NV, int X2, int Y2, int Z2, int C,
                                                      const T* __restrict__ ADD, long lda, T* __restrict__ V, long ldv) {
  const int q4 = C / 4;
  const long tot = (long)NV * X2 * Y2 * Z2 * 8 * q4;
  const D2S ds{X2, Y2, Z2};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {      // over OUTPUT voxels: contiguous stores
    const int c = (int)(i % q4) * 4; const long ro = i / q4;
    const long r = d2s_src_row(ro, ds);                        // (voxel, tap) row of P: P[q][tap * C + c] at q = r >> 3, tap = r & 7
    const f32x4 v = ld4f(P + (r >> 3) * ldp + (r & 7) * C + c) + ld4f(ADD + ro * lda + c);
    st4f(V + ro * ldv + c, v);
  }
}

// ---- trilinear resize, align_corners=True (nn.Upsample(mode='trilinear'), model_3D.py:46-58)
__global__ __launch_bounds__(256) void trilinear_fwd_kernel(const float* __restrict__ X, long ldx, int NV, int Di, int Hi, int Wi,
                                                           int C, int Do, int Ho, int Wo, float* __restrict__ Y, long ldy) {
  const int q4 = C / 4;
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long tot = (long)NV * Do * Ho * Wo * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xo = r % Wo; r /= Wo; const int yo = r % Ho; r /= Ho; const int zo = r % Do; const long n = r / Do;
    int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
    ac_src(zo, sd, Di, z0, z1, lz); ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
    const float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = X + (n * Di) * (long)Hi * Wi * ldx + c;
#define TL(zz, yy, xx) (*reinterpret_cast<const f32x4*>(b + (((long)(zz) * Hi + (yy)) * Wi + (xx)) * ldx))
    const f32x4 v000 = TL(z0, y0, x0), v001 = TL(z0, y0, x1), v010 = TL(z0, y1, x0), v011 = TL(z0, y1, x1);
    const f32x4 v100 = TL(z1, y0, x0), v101 = TL(z1, y0, x1), v110 = TL(z1, y1, x0), v111 = TL(z1, y1, x1);
#undef TL
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = tl_blend1(v000[e], v001[e], v010[e], v011[e], v100[e], v101[e], v110[e], v111[e], hx, lx, hy, ly, hz, lz);
    *reinterpret_cast<f32x4*>(Y + (((n * Do + zo) * Ho + yo) * (long)Wo + xo) * ldy + c) = o;
  }
}
__device__ __forceinline__ void ac_range(int i, float s, int out_size, int& a, int& b) {
  if (s > 0.f) { const float is = 1.f / s; a = (int)floorf((float)(i - 1) * is) - 1; b = (int)ceilf((float)(i + 1) * is) + 1; }
  else { a = 0; b = out_size - 1; }
  a = max(a, 0); b = min(b, out_size - 1);
}
__device__ __forceinline__ float ac_weight(int o, float s, int in_size, int i) {
  int i0, i1; float l; ac_src(o, s, in_size, i0, i1, l);
  float w = 0.f;
  if (i0 == i) w += 1.f - l;
  if (i1 == i) w += l;
  return (i0 == i || i1 == i) ? w : -1.f;      // -1: output o does not reference input i
}
__global__ __launch_bounds__(256) void trilinear_bwd_kernel(const float* __restrict__ dY, long ldy, int NV, int Di, int Hi, int Wi,
                                                           int C, int Do, int Ho, int Wo, float* __restrict__ dX, long ldx) {
  const int q4 = C / 4;
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long tot = (long)NV * Di * Hi * Wi * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int c = (int)(i % q4) * 4; long r = i / q4;
    const int xi = r % Wi; r /= Wi; const int yi = r % Hi; r /= Hi; const int zi = r % Di; const long n = r / Di;
    int za, zb, ya, yb, xa, xb;
    ac_range(zi, sd, Do, za, zb); ac_range(yi, sh, Ho, ya, yb); ac_range(xi, sw, Wo, xa, xb);
    f32x4 acc = {0, 0, 0, 0};
    for (int zo = za; zo <= zb; ++zo) {
      const float wz = ac_weight(zo, sd, Di, zi); if (wz < 0.f) continue;
      for (int yo = ya; yo <= yb; ++yo) {
        const float wy = ac_weight(yo, sh, Hi, yi); if (wy < 0.f) continue;
        for (int xo = xa; xo <= xb; ++xo) {
          const float wx = ac_weight(xo, sw, Wi, xi); if (wx < 0.f) continue;
          const f32x4 g = *reinterpret_cast<const f32x4*>(dY + (((n * Do + zo) * Ho + yo) * (long)Wo + xo) * ldy + c);
          const float w = wz * wy * wx;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w * g[e];
        }
      }
    }
    *reinterpret_cast<f32x4*>(dX + (((n * Di + zi) * Hi + yi) * (long)Wi + xi) * ldx + c) = acc;
  }
}

// ---- 3-D anchor-row gather of cat(trilinear_up(lo), hi) and its adjoint (row-sparse head of FeatureExtractor_3d,
// model_3D.py:52-55).  Same fp32 index math / lerp order as trilinear_fwd_kernel -> rows bit-identical to the dense path.
template <typename TH>
__global__ __launch_bounds__(256) void gather_upcat_rows3d_kernel(const float* __restrict__ lo, long ldlo, int Clo, int Di, int Hi, int Wi,
                                                                 const TH* __restrict__ hi, long ldhi, int Chi, int Do, int Ho, int Wo,
                                                                 const int64_t* __restrict__ pix, long n, float* __restrict__ X, long ldx) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const long p = pix[j];
  const long vol = (long)Do * Ho * Wo;
  const long img = p / vol; long rem = p - img * vol;
  const int zo = (int)(rem / ((long)Ho * Wo)); rem -= (long)zo * Ho * Wo;
  const int yo = (int)(rem / Wo), xo = (int)(rem - (long)yo * Wo);
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
  ac_src(zo, sd, Di, z0, z1, lz); ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
  const float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
  const float* b = lo + (img * Di) * (long)Hi * Wi * ldlo;
#define GP(zz, yy, xx) (b + (((long)(zz) * Hi + (yy)) * Wi + (xx)) * ldlo)
  const float *p000 = GP(z0, y0, x0), *p001 = GP(z0, y0, x1), *p010 = GP(z0, y1, x0), *p011 = GP(z0, y1, x1);
  const float *p100 = GP(z1, y0, x0), *p101 = GP(z1, y0, x1), *p110 = GP(z1, y1, x0), *p111 = GP(z1, y1, x1);
#undef GP
  float* o = X + j * ldx;
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 v000 = *reinterpret_cast<const f32x4*>(p000 + c), v001 = *reinterpret_cast<const f32x4*>(p001 + c);
    const f32x4 v010 = *reinterpret_cast<const f32x4*>(p010 + c), v011 = *reinterpret_cast<const f32x4*>(p011 + c);
    const f32x4 v100 = *reinterpret_cast<const f32x4*>(p100 + c), v101 = *reinterpret_cast<const f32x4*>(p101 + c);
    const f32x4 v110 = *reinterpret_cast<const f32x4*>(p110 + c), v111 = *reinterpret_cast<const f32x4*>(p111 + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      r[e] = tl_blend1(v000[e], v001[e], v010[e], v011[e], v100[e], v101[e], v110[e], v111[e], hx, lx, hy, ly, hz, lz);
    *reinterpret_cast<f32x4*>(o + c) = r;
  }
  const TH* h = hi + p * ldhi;
  for (int c = lane * 4; c < Chi; c += 256) *reinterpret_cast<f32x4*>(o + Clo + c) = ld4f(h + c);
}
// ---- the level below, lazily too (head.LazyHead3dL3Fn): V holds the eight corner rows of every sampled voxel (rows 8 j + k, corner order
// z, y, x with x fastest - arco_corner_rows3d), each already evaluated through its own layer;
// X[j][0..Clo) = their trilinear blend in the gather's index / lerp arithmetic (-> the values the dense map would have handed to
// gather_upcat_rows3d_kernel, up to the rounding of the rows themselves), X[j][Clo..Clo+Chi) = hi[pix[j]]
template <typename TH>
__global__ __launch_bounds__(256) void lerp8_cat_rows3d_kernel(const float* __restrict__ V, long ldv, int Clo, int Di, int Hi, int Wi,
                                                              const TH* __restrict__ hi, long ldhi, int Chi, int Do, int Ho, int Wo,
                                                              const int64_t* __restrict__ pix, long n, float* __restrict__ X, long ldx) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const long p = pix[j];
  const long vol = (long)Do * Ho * Wo;
  const long img = p / vol; long rem = p - img * vol;
  const int zo = (int)(rem / ((long)Ho * Wo)); rem -= (long)zo * Ho * Wo;
  const int yo = (int)(rem / Wo), xo = (int)(rem - (long)yo * Wo);
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
  ac_src(zo, sd, Di, z0, z1, lz); ac_src(yo, sh, Hi, y0, y1, ly); ac_src(xo, sw, Wi, x0, x1, lx);
  const float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
  const float* v = V + (8 * j) * ldv;
  float* o = X + j * ldx;
  for (int c = lane * 4; c < Clo; c += 256) {
    f32x4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const f32x4*>(v + k * ldv + c);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = tl_blend1(q[0][e], q[1][e], q[2][e], q[3][e], q[4][e], q[5][e], q[6][e], q[7][e], hx, lx, hy, ly, hz, lz);
    *reinterpret_cast<f32x4*>(o + c) = r;
  }
  const TH* h = hi + p * ldhi;
  for (int c = lane * 4; c < Chi; c += 256) *reinterpret_cast<f32x4*>(o + Clo + c) = ld4f(h + c);
}
// adjoint of the blend: dV[8 j + k] = w8[8 j + k] * dX[j][0..Clo)   (w8: arco_corner_rows3d's weights; plain stores)
__global__ __launch_bounds__(256) void lerp8_rows3d_bwd_kernel(const float* __restrict__ dX, long ldx, int Clo, const float* __restrict__ w8,
                                                              long n, float* __restrict__ dV, long ldv) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const float* g = dX + j * ldx;
  float* v = dV + (8 * j) * ldv;
  float wk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) wk[k] = w8[8 * j + k];
  for (int c = lane * 4; c < Clo; c += 256) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(g + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) asm("v_mul_f32 %0, %1, %2" : "=v"(r[e]) : "v"(d[e]), "v"(wk[k]));      // (never a packed multiply: see lerp4_cat_rows_bwd_kernel)
      *reinterpret_cast<f32x4*>(v + k * ldv + c) = r;
    }
  }
}
__global__ __launch_bounds__(256) void scatter_upcat_rows3d_kernel(const float* __restrict__ dX, long ldx, const int64_t* __restrict__ pix, long n,
                                                                  float* __restrict__ dlo, long ldlo, int Clo, int Di, int Hi, int Wi,
                                                                  float* __restrict__ dhi, long ldhi, int Chi, int Do, int Ho, int Wo) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const long p = pix[j];
  const long vol = (long)Do * Ho * Wo;
  const long img = p / vol; long rem = p - img * vol;
  const int zo = (int)(rem / ((long)Ho * Wo)); rem -= (long)zo * Ho * Wo;
  const int yo = (int)(rem / Wo), xo = (int)(rem - (long)yo * Wo);
  const float sd = Do > 1 ? (float)(Di - 1) / (float)(Do - 1) : 0.f, sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f,
              sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int zz[2], yy[2], xx[2]; float lz, ly, lx;
  ac_src(zo, sd, Di, zz[0], zz[1], lz); ac_src(yo, sh, Hi, yy[0], yy[1], ly); ac_src(xo, sw, Wi, xx[0], xx[1], lx);
  const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
  float* b = dlo + (img * Di) * (long)Hi * Wi * ldlo;
  const float* g = dX + j * ldx;
  for (int c = lane; c < Clo; c += 64) {
    const float v = g[c];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
          atomicAdd(b + (((long)zz[a] * Hi + yy[bb]) * Wi + xx[d]) * ldlo + c, wz[a] * wy[bb] * wx[d] * v);
  }
  float* h = dhi + p * ldhi;
  for (int c = lane; c < Chi; c += 64) atomicAdd(h + c, g[Clo + c]);
}

// ---- strided channel-slice copy / add:  Y[r][0..C) (+)= X[r][0..C)
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ X, long ldx, long M, int C,
                                                       float* __restrict__ Y, long ldy, int accumulate) {
  const int q4 = C / 4;
  const long tot = M * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long r = i / q4; const int c = (int)(i - r * q4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
    float* d = Y + r * ldy + c;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(d);
    *reinterpret_cast<f32x4*>(d) = v;
  }
}

// NCHW plane layout <-> channels-last rows (API boundary conversion)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ X, int NB, int C, long P, float* __restrict__ Y, long ldy) {
  __shared__ float tile[32][33];
  const long n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {          // r: channel
    const int c = c0 + r; const long p = p0 + threadIdx.x;
    tile[r][threadIdx.x] = (c < C && p < P) ? X[(n * C + c) * P + p] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {          // r: pixel
    const long p = p0 + r; const int c = c0 + threadIdx.x;
    if (p < P && c < C) Y[(n * P + p) * ldy + c] = tile[threadIdx.x][r];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ X, long ldx, int NB, int C, long P, float* __restrict__ Y) {
  __shared__ float tile[32][33];
  const long n = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32; const int c0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {          // r: pixel
    const long p = p0 + r; const int c = c0 + threadIdx.x;
    tile[r][threadIdx.x] = (c < C && p < P) ? X[(n * P + p) * ldx + c] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {          // r: channel
    const int c = c0 + r; const long p = p0 + threadIdx.x;
    if (p < P && c < C) Y[(n * C + c) * P + p] = tile[threadIdx.x][r];
  }
}

// ---- flat-buffer optimiser steps
// torch.optim.SGD(momentum, weight_decay, nesterov=True): g += wd*p; buf = first ? g : mom*buf + g;
// p -= lr * (g + mom*buf)
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ buf, long n, float lr, float mom, float wd,
                                                          int first) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pv = p[i];
    const float gv = g[i] + wd * pv;
    const float b = (first & 1) ? gv : mom * buf[i] + gv;
    buf[i] = b;
    p[i] = pv - lr * ((first & 2) ? b : gv + mom * b);       // bit 1: plain momentum (torch.optim.SGD, nesterov=False)
  }
}
// k = k*m + q*(1-m)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ k, const float* __restrict__ q, long n, float m) {
  const float om = 1.0f - m;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) k[i] = k[i] * m + q[i] * om;
}

static inline int ew_grid(long work) {
  long g = (work + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- small glue kernels that replace chains of ATen launches in the step (profiles/r02_h: ~0.6 ms of fills / copies / adds)
// rows idx[0..n) of a [rows, ld] tensor set to zero over C channels: re-arms a persistent, zero-by-invariant gradient
// buffer after a sparse scatter (the row-sparse head's feature-map gradients) instead of a dense fill per step
template <typename T>
__global__ __launch_bounds__(256) void zero_rows_kernel(T* __restrict__ dst, long ld, int C, const int64_t* __restrict__ idx, long n) {
  const int q4 = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * q4; i += (long)gridDim.x * 256) {
    const long r = i / q4; const int c = (int)(i - r * q4) * 4;
    st4f(dst + idx[r] * ld + c, f32x4{0, 0, 0, 0});
  }
}
// rows idx[0..n) of an fp32 [rows, ld] gradient buffer -> the same rows of an f16 buffer, multiplied by the loss scale and saturated
// at the largest finite f16 (see cast_f2h_kernel): the row-sparse head's feature-map gradients enter the f16 region without a dense
// cast of the whole map.  Duplicate indices write the same value twice.
__global__ __launch_bounds__(256) void cast_rows_f2h_kernel(const float* __restrict__ src, long lds_, int C, const int64_t* __restrict__ idx,
                                                           long n, float scale, _Float16* __restrict__ dst, long ldd) {
  const int q4 = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * q4; i += (long)gridDim.x * 256) {
    const long r = i / q4; const int c = (int)(i - r * q4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + idx[r] * lds_ + c) * scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 65504.f ? 65504.f : (v[j] < -65504.f ? -65504.f : v[j]);
    st4f(dst + idx[r] * ldd + c, v);
  }
}
// W' = W + I of a square [n, n] 1x1-conv weight, written as its two column blocks lo [n, c] and hi [n, n - c] (the
// FeatureExtractor's residual folded into the weights, model_2D.py forward_lowres1/2): one launch for eye + add + 2 slices
__global__ __launch_bounds__(256) void fold_residual_kernel(const float* __restrict__ W, int n, int c, float* __restrict__ lo, float* __restrict__ hi) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / n), k = (int)(i - (long)r * n);
    const float v = W[i] + (r == k ? 1.f : 0.f);
    if (k < c) lo[(long)r * c + k] = v; else hi[(long)r * (n - c) + (k - c)] = v;
  }
}
// the reverse for the gradient: dW[r][k] = k < c ? dlo[r][k] : dhi[r][k - c]
__global__ __launch_bounds__(256) void unfold_residual_kernel(const float* __restrict__ dlo, const float* __restrict__ dhi, int n, int c, float* __restrict__ dW) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / n), k = (int)(i - (long)r * n);
    dW[i] = k < c ? (dlo ? dlo[(long)r * c + k] : 0.f) : (dhi ? dhi[(long)r * (n - c) + (k - c)] : 0.f);
  }
}
// out[0] = sum_i w_i * (*term_i)  /  grads[i] = w_i * g[0]: the step's loss combination (train_arco_2d.py:426) and its
// backward as one launch each instead of a chain of 0-d multiplies and adds
struct TermTable { const float* p[8]; float w[8]; int n; };
__global__ void combine_terms_kernel(TermTable t, float* __restrict__ out) {
  if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < t.n; ++i) s += t.w[i] * t.p[i][0]; out[0] = s; }
}
__global__ void combine_terms_bwd_kernel(TermTable t, const float* __restrict__ g, float* __restrict__ grads) {
  if ((int)threadIdx.x < t.n) grads[threadIdx.x] = t.w[threadIdx.x] * g[0];
}

// ---- GroupNorm / InstanceNorm statistics (vnetWithArgs.py:19-22 `normalization='groupnorm' | 'instancenorm'`): one
// normalisation set = (sample n, cpg consecutive channels, all voxels).  The per-(sample, channel) slab sums are the BN
// machinery's with groups = samples; this finalize sums a set's channels: mean / istd rows [N][C], the set's value in
// each of its channels, so that bn_act_fwd_kernel applies them unchanged.  grid (C / cpg, N), fp64, fixed order.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ ssum, const float* __restrict__ ssq, int nblk, int C,
                                                         int cpg, double count, float eps, float* __restrict__ mean,
                                                         float* __restrict__ istd) {
  __shared__ double sh[2][4];
  const int set = blockIdx.x, n = blockIdx.y, N = gridDim.y, npg = nblk / N, w = threadIdx.x >> 6;
  double s = 0.0, q = 0.0;
  for (int j = 0; j < cpg; ++j) {
    const int c = set * cpg + j;
    slab_sums2<256>(ssum + (long)c * nblk + n * npg, ssq + (long)c * nblk + n * npg, npg, s, q);
  }
  s = wave_sum_d(s); q = wave_sum_d(q);
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
  __syncthreads();
  if (threadIdx.x < cpg) {
    s = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]); q = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[(long)n * C + set * cpg + threadIdx.x] = (float)m;
    istd[(long)n * C + set * cpg + threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
// backward: sums [N][2][C] (per sample and channel: sum dy, sum dy*xhat) -> per set the sums of gamma*dy, gamma*dy*xhat,
// written back into every channel of the set (what bn_act_bwd_apply_kernel(gn = 1) reads).  One thread per (n, set).
__global__ void gn_merge_bwd_kernel(float* __restrict__ sums, const float* __restrict__ gamma, int N, int C, int cpg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, sets = C / cpg;
  if (i >= N * sets) return;
  const int n = i / sets, set = i - n * sets;
  float* s1 = sums + (2l * n) * C + set * cpg; float* s2 = sums + (2l * n + 1) * C + set * cpg;
  double a = 0.0, b = 0.0;
  for (int j = 0; j < cpg; ++j) { const double g = gamma ? (double)gamma[set * cpg + j] : 1.0; a += g * (double)s1[j]; b += g * (double)s2[j]; }
  for (int j = 0; j < cpg; ++j) { s1[j] = (float)a; s2[j] = (float)b; }
}

extern "C" {

int arco_bn_finalize(const float* ssum, const float* ssq, int nblk, int C, long count, float eps, float momentum,
                     float* mean, float* istd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     int groups, int defer_from, float* deferred, void* stream) {
  if (groups < 1) groups = 1;
  ARCO_CHECK_ARG(C > 0 && nblk > 0 && count > 0 && nblk % groups == 0 && count % groups == 0 &&
                 (!deferred || (defer_from >= 0 && defer_from < groups)));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), ssum, ssq, nblk, C,
                     (double)(count / groups), eps, momentum, mean, istd, running_mean, running_var, num_batches_tracked,
                     groups, defer_from, deferred);
  return arco_launch_status();
}
long arco_bn_defer_desc_bytes() { return (long)sizeof(BnDeferDesc); }
// desc: device array of n_layers BnDeferDesc {running_mean*, running_var*, deferred*, C, n, momentum, pad}
int arco_bn_apply_deferred(const void* desc, int n_layers, void* stream) {
  if (n_layers <= 0) return ARCO_OK;
  ARCO_CHECK_ARG(desc);
  hipLaunchKernelGGL(bn_apply_deferred_kernel, dim3(n_layers), dim3(256), 0, as_stream(stream), (const BnDeferDesc*)desc);
  return arco_launch_status();
}

// slabs for the per-channel reductions: 512 rows per block on big tensors, but never fewer than ~512 blocks while
// a block still has >= 16 rows (the deep levels have 2048-8192 rows x 128-256 channels: 4-16 blocks left the GPU idle)
int arco_chan_stats_blocks(long M) {
  long b = (M + 511) / 512;
  long fill = (M + 15) / 16; if (fill > 512) fill = 512;
  if (b < fill) b = fill;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C++" {
template <typename T>
static int chan_stats_impl(const T* X, long ldx, long M, int C, float* ssum, float* ssq, int groups, void* stream) {
  if (groups < 1) groups = 1;
  ARCO_CHECK_ARG(C > 0 && (C & 3) == 0 && C <= 1024 && (ldx & 3) == 0 && M % groups == 0);
  const int nblk = arco_chan_stats_blocks(M / groups);            // slabs per group; ssum/ssq: [C][groups*nblk]
  hipLaunchKernelGGL(chan_stats_kernel<T>, dim3(nblk, groups), dim3(256), 2048 * sizeof(float), as_stream(stream), X, ldx,
                     M / groups, C, ssum, ssq, nblk);
  return arco_launch_status();
}
}  // extern "C++"
int arco_chan_stats(const float* X, long ldx, long M, int C, float* ssum, float* ssq, int groups, void* stream) {
  return chan_stats_impl<float>(X, ldx, M, C, ssum, ssq, groups, stream);
}
// ... of an f16 tensor (f16 activation storage; statistics stay fp32)
int arco_chan_stats_h(const void* X, long ldx, long M, int C, float* ssum, float* ssq, int groups, void* stream) {
  return chan_stats_impl<_Float16>(reinterpret_cast<const _Float16*>(X), ldx, M, C, ssum, ssq, groups, stream);
}

extern "C++" {
template <typename T>
static int bn_act_fwd_impl(const T* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                           const float* beta, float slope, int drop_mode, float p, uint64_t seed, long P, T* A, long lda,
                           const uint64_t* seed_dev, int groups, void* stream, const T* R = nullptr, long ldr = 0, D2S ds = D2S{0, 0, 0}) {
  if (groups < 1) groups = 1;
  ARCO_CHECK_ARG(C > 0 && (C & 3) == 0 && (ldz & 3) == 0 && (lda & 3) == 0 && p < 1.0f && M % groups == 0 && (!R || (ldr & 3) == 0));
  const long Mg = M / groups;                 // mean / istd: [groups][C]; rows [g*Mg, (g+1)*Mg) use row g
  hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(Mg * (C / 4)), groups), dim3(256), 0, as_stream(stream), Z, ldz, Mg, C, mean,
                     istd, gamma, beta, slope, p > 0.f ? drop_mode : 0, p, seed, P, A, lda, seed_dev, R, ldr, ds);
  return arco_launch_status();
}
}  // extern "C++"
int arco_bn_act_fwd(const float* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                    const float* beta, float slope, int drop_mode, float p, uint64_t seed, long P, float* A, long lda,
                    const uint64_t* seed_dev, int groups, void* stream) {
  return bn_act_fwd_impl<float>(Z, ldz, M, C, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, A, lda, seed_dev, groups, stream);
}
// A = drop(lrelu(BN(Z))) + R: the apply pass with the decoder's skip addition (vnetWithArgs.py:224-236 `block_x_up(x) + skip`)
int arco_bn_act_add_fwd(const float* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                        const float* beta, float slope, const float* R, long ldr, float* A, long lda, int groups, void* stream) {
  ARCO_CHECK_ARG(R != nullptr);
  return bn_act_fwd_impl<float>(Z, ldz, M, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1, A, lda, nullptr, groups, stream, R, ldr);
}
int arco_bn_act_add_fwd_h(const void* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                          const float* beta, float slope, const void* R, long ldr, void* A, long lda, int groups, void* stream) {
  ARCO_CHECK_ARG(R != nullptr);
  return bn_act_fwd_impl<_Float16>(reinterpret_cast<const _Float16*>(Z), ldz, M, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1,
                                   reinterpret_cast<_Float16*>(A), lda, nullptr, groups, stream, reinterpret_cast<const _Float16*>(R), ldr);
}
// UpsamplingDeconvBlock's BatchNorm + ReLU straight from the GEMM-form transposed conv (see D2S): Y rows = (voxel of the NV x X2 x Y2 x Z2
// grid, tap) of C channels (Y = [voxels][8 C] dense), A / R = the [NV, 2 X2, 2 Y2, 2 Z2] activation / skip tensor; M8 = 8 * voxels
int arco_bn_act_d2s_fwd(const float* Y, long M8, int C, const float* mean, const float* istd, const float* gamma, const float* beta,
                        float slope, const float* R, long ldr, float* A, long lda, int X2, int Y2, int Z2, int groups, void* stream) {
  ARCO_CHECK_ARG(X2 > 0 && Y2 > 0 && Z2 > 0 && M8 % (8l * X2 * Y2 * Z2) == 0);
  return bn_act_fwd_impl<float>(Y, C, M8, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1, A, lda, nullptr, groups, stream, R, ldr, D2S{X2, Y2, Z2});
}
int arco_bn_act_d2s_fwd_h(const void* Y, long M8, int C, const float* mean, const float* istd, const float* gamma, const float* beta,
                          float slope, const void* R, long ldr, void* A, long lda, int X2, int Y2, int Z2, int groups, void* stream) {
  ARCO_CHECK_ARG(X2 > 0 && Y2 > 0 && Z2 > 0 && M8 % (8l * X2 * Y2 * Z2) == 0);
  return bn_act_fwd_impl<_Float16>(reinterpret_cast<const _Float16*>(Y), C, M8, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1,
                                   reinterpret_cast<_Float16*>(A), lda, nullptr, groups, stream, reinterpret_cast<const _Float16*>(R), ldr,
                                   D2S{X2, Y2, Z2});
}
// f16 activation storage: Z and A are f16 tensors (parameters and statistics fp32, arithmetic fp32)
int arco_bn_act_fwd_h(const void* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                      const float* beta, float slope, int drop_mode, float p, uint64_t seed, long P, void* A, long lda,
                      const uint64_t* seed_dev, int groups, void* stream) {
  return bn_act_fwd_impl<_Float16>(reinterpret_cast<const _Float16*>(Z), ldz, M, C, mean, istd, gamma, beta, slope, drop_mode, p, seed, P,
                                   reinterpret_cast<_Float16*>(A), lda, seed_dev, groups, stream);
}

// ws: groups * (2*C*nblk + 2*C) floats;  nblk = arco_chan_stats_blocks(M / groups)
// cpg = 0: BatchNorm (groups = BN groups).  cpg >= 1: GroupNorm with cpg channels per set / InstanceNorm (cpg = 1),
// groups = samples; gamma / beta may be null there (no affine: gamma = 1, beta = 0 are expected as real tensors by the kernels,
// so the host passes ones / zeros).
extern "C++" {
template <typename T>
static int bn_act_bwd_impl(const T* dA, long ldd, const T* Z, long ldz, long M, int C, const float* mean,
                           const float* istd, const float* gamma, const float* beta, float slope, int drop_mode, float p,
                           uint64_t seed, long P, float* ws, float* dgamma, float* dbeta, int accumulate, T* dZ, long ldo,
                           const uint64_t* seed_dev, int groups, int cpg, void* stream, D2S ds = D2S{0, 0, 0}) {
  if (groups < 1) groups = 1;
  ARCO_CHECK_ARG(C > 0 && (C & 3) == 0 && C <= 1024 && (ldz & 3) == 0 && (ldd & 3) == 0 && (ldo & 3) == 0 && M % groups == 0);
  ARCO_CHECK_ARG(cpg == 0 || (mean && cpg >= 1 && C % cpg == 0));
  const int dm = p > 0.f ? drop_mode : 0;
  hipStream_t st = as_stream(stream);
  const long Mg = M / groups;
  if (mean) {
    const int nblk = arco_chan_stats_blocks(Mg);
    float* s_dy = ws; float* s_dyx = ws + (long)groups * C * nblk; float* sums = ws + 2l * groups * C * nblk;
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<T>, dim3(nblk, groups), dim3(256), 2048 * sizeof(float), st, dA, ldd, Z, ldz, Mg,
                       C, mean, istd, gamma, beta, slope, dm, p, seed, P, s_dy, s_dyx, nblk, seed_dev, ds);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, s_dy, s_dyx, nblk, C, sums, dgamma,
                       dbeta, accumulate, groups);
    if (cpg) {
      const int nset = groups * (C / cpg);
      hipLaunchKernelGGL(gn_merge_bwd_kernel, dim3((nset + 255) / 256), dim3(256), 0, st, sums, gamma, groups, C, cpg);
    }
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(ew_grid(Mg * (C / 4)), groups), dim3(256), 0, st, dA, ldd, Z, ldz, Mg, C,
                       mean, istd, gamma, beta, slope, dm, p, seed, P, sums, sums + C, 1.0f / ((float)Mg * (float)(cpg ? cpg : 1)), dZ, ldo,
                       seed_dev, cpg ? 1 : 0, ds);
  } else {
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(ew_grid(M * (C / 4))), dim3(256), 0, st, dA, ldd, Z, ldz, M, C,
                       nullptr, nullptr, nullptr, nullptr, slope, dm, p, seed, P, nullptr, nullptr, 0.f, dZ, ldo, seed_dev, 0);
  }
  return arco_launch_status();
}
}  // extern "C++"
int arco_bn_act_bwd(const float* dA, long ldd, const float* Z, long ldz, long M, int C, const float* mean,
                    const float* istd, const float* gamma, const float* beta, float slope, int drop_mode, float p,
                    uint64_t seed, long P, float* ws, float* dgamma, float* dbeta, int accumulate, float* dZ, long ldo,
                    const uint64_t* seed_dev, int groups, void* stream) {
  return bn_act_bwd_impl<float>(dA, ldd, Z, ldz, M, C, mean, istd, gamma, beta, slope, drop_mode, p, seed, P, ws, dgamma, dbeta,
                                accumulate, dZ, ldo, seed_dev, groups, 0, stream);
}
// f16 activation storage: dA, Z, dZ are f16 tensors; the channel sums (ws) and dgamma / dbeta fp32
int arco_bn_act_bwd_h(const void* dA, long ldd, const void* Z, long ldz, long M, int C, const float* mean,
                      const float* istd, const float* gamma, const float* beta, float slope, int drop_mode, float p,
                      uint64_t seed, long P, float* ws, float* dgamma, float* dbeta, int accumulate, void* dZ, long ldo,
                      const uint64_t* seed_dev, int groups, void* stream) {
  return bn_act_bwd_impl<_Float16>(reinterpret_cast<const _Float16*>(dA), ldd, reinterpret_cast<const _Float16*>(Z), ldz, M, C, mean, istd,
                                   gamma, beta, slope, drop_mode, p, seed, P, ws, dgamma, dbeta, accumulate,
                                   reinterpret_cast<_Float16*>(dZ), ldo, seed_dev, groups, 0, stream);
}
// backward of arco_bn_act_d2s_fwd: dA in the activation's voxel order, dY in Y's (voxel, tap) row order (what the GEMM's gradients read)
int arco_bn_act_d2s_bwd(const float* dA, long ldd, const float* Y, long M8, int C, const float* mean, const float* istd,
                        const float* gamma, const float* beta, float slope, float* ws, float* dgamma, float* dbeta, int accumulate,
                        float* dY, int X2, int Y2, int Z2, int groups, void* stream) {
  ARCO_CHECK_ARG(X2 > 0 && Y2 > 0 && Z2 > 0 && M8 % (8l * X2 * Y2 * Z2) == 0 && mean);
  return bn_act_bwd_impl<float>(dA, ldd, Y, C, M8, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1, ws, dgamma, dbeta, accumulate, dY, C,
                                nullptr, groups, 0, stream, D2S{X2, Y2, Z2});
}
int arco_bn_act_d2s_bwd_h(const void* dA, long ldd, const void* Y, long M8, int C, const float* mean, const float* istd,
                          const float* gamma, const float* beta, float slope, float* ws, float* dgamma, float* dbeta, int accumulate,
                          void* dY, int X2, int Y2, int Z2, int groups, void* stream) {
  ARCO_CHECK_ARG(X2 > 0 && Y2 > 0 && Z2 > 0 && M8 % (8l * X2 * Y2 * Z2) == 0 && mean);
  return bn_act_bwd_impl<_Float16>(reinterpret_cast<const _Float16*>(dA), ldd, reinterpret_cast<const _Float16*>(Y), C, M8, C, mean, istd,
                                   gamma, beta, slope, 0, 0.f, 0, 1, ws, dgamma, dbeta, accumulate, reinterpret_cast<_Float16*>(dY), C,
                                   nullptr, groups, 0, stream, D2S{X2, Y2, Z2});
}
// GroupNorm / InstanceNorm + activation (vnetWithArgs.py:19-22): statistics of N samples x (C / cpg) channel sets from the
// per-(sample, channel) slabs of arco_chan_stats(groups = N) / the conv epilogue; apply = arco_bn_act_fwd(groups = N)
int arco_gn_finalize(const float* ssum, const float* ssq, int nblk, int C, int cpg, int N, long count_per_sample_channel,
                     float eps, float* mean, float* istd, void* stream) {
  ARCO_CHECK_ARG(ssum && ssq && mean && istd && C > 0 && cpg >= 1 && cpg <= 256 && C % cpg == 0 && N >= 1 && nblk % N == 0 &&
                 count_per_sample_channel > 0);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(C / cpg, N), dim3(256), 0, as_stream(stream), ssum, ssq, nblk, C, cpg,
                     (double)count_per_sample_channel * (double)cpg, eps, mean, istd);
  return arco_launch_status();
}
int arco_gn_act_bwd(const float* dA, long ldd, const float* Z, long ldz, long M, int C, const float* mean,
                    const float* istd, const float* gamma, const float* beta, float slope, float* ws, float* dgamma,
                    float* dbeta, int accumulate, float* dZ, long ldo, int N, int cpg, void* stream) {
  ARCO_CHECK_ARG(mean && istd && gamma && beta && cpg >= 1);
  return bn_act_bwd_impl<float>(dA, ldd, Z, ldz, M, C, mean, istd, gamma, beta, slope, 0, 0.f, 0, 1, ws, dgamma, dbeta, accumulate,
                         dZ, ldo, nullptr, N, cpg, stream);
}

int arco_maxpool2_fwd(const float* X, long ldx, int NB, int H, int W, int C, float* Y, long ldy, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (H & 1) == 0 && (W & 1) == 0);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid((long)NB * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     as_stream(stream), X, ldx, NB, H, W, C, Y, ldy);
  return arco_launch_status();
}
int arco_bn_act_pool_fwd(const float* Z, long ldz, int NB, int H, int W, int C, const float* mean, const float* istd,
                         const float* gamma, const float* beta, float slope, float* A, long lda, float* P, long ldp,
                         int groups, void* stream) {
  if (groups < 1) groups = 1;
  ARCO_CHECK_ARG(Z && mean && istd && gamma && beta && A && P && C > 0 && (C & 3) == 0 && (ldz & 3) == 0 && (lda & 3) == 0 &&
                 (ldp & 3) == 0 && (H & 1) == 0 && (W & 1) == 0 && NB % groups == 0);
  hipLaunchKernelGGL(bn_act_pool_fwd_kernel, dim3(ew_grid((long)NB * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     as_stream(stream), Z, ldz, NB, H, W, C, mean, istd, gamma, beta, slope, A, lda, P, ldp, NB / groups);
  return arco_launch_status();
}
int arco_maxpool2_bwd(const float* X, long ldx, int NB, int H, int W, int C, const float* dY, long ldy, float* dX,
                      long ldo, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (H & 1) == 0 && (W & 1) == 0);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid((long)NB * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     as_stream(stream), X, ldx, NB, H, W, C, dY, ldy, dX, ldo, (const float*)nullptr, 0l);
  return arco_launch_status();
}
// dX = maxpool2_bwd(dY) + add  (add: the gradient x receives from its other consumer, e.g. the U-Net skip connection)
int arco_maxpool2_bwd_add(const float* X, long ldx, int NB, int H, int W, int C, const float* dY, long ldy, const float* add,
                          long ld_add, float* dX, long ldo, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (H & 1) == 0 && (W & 1) == 0 && add && (ld_add & 3) == 0);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid((long)NB * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     as_stream(stream), X, ldx, NB, H, W, C, dY, ldy, dX, ldo, add, ld_add);
  return arco_launch_status();
}

int arco_bilinear_fwd(const float* X, long ldx, int NB, int Hi, int Wi, int C, int Ho, int Wo, float* Y, long ldy,
                      void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(ew_grid((long)NB * Ho * Wo * (C / 4))), dim3(256), 0, as_stream(stream),
                     X, ldx, NB, Hi, Wi, C, Ho, Wo, Y, ldy);
  return arco_launch_status();
}
int arco_bilinear_bwd(const float* dY, long ldy, int NB, int Hi, int Wi, int C, int Ho, int Wo, float* dX, long ldx,
                      int accumulate, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(ew_grid((long)NB * Hi * Wi * (C / 4))), dim3(256), 0, as_stream(stream),
                     dY, ldy, NB, Hi, Wi, C, Ho, Wo, dX, ldx, accumulate);
  return arco_launch_status();
}

int arco_gather_upcat_rows(const float* lo, long ldlo, int Clo, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                           int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldlo & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(gather_upcat_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), lo, ldlo, Clo, Hi, Wi,
                     hi, ldhi, Chi, Ho, Wo, pix, n, X, ldx);
  return arco_launch_status();
}
int arco_scatter_upcat_rows(const float* dX, long ldx, const int64_t* pix, long n, float* dlo, long ldlo, int Clo, int Hi,
                            int Wi, float* dhi, long ldhi, int Chi, int Ho, int Wo, void* stream) {
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(scatter_upcat_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), dX, ldx, pix, n, dlo,
                     ldlo, Clo, Hi, Wi, dhi, ldhi, Chi, Ho, Wo);
  return arco_launch_status();
}

int arco_up_neighbors(const int64_t* pix, long n, int Hi, int Wi, int Ho, int Wo, int64_t* nb4, float* lylx, void* stream) {
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(up_neighbors_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), pix, n, Hi, Wi, Ho, Wo, nb4, lylx);
  return arco_launch_status();
}
int arco_lerp4_cat_rows(const float* V, long ldv, int Clo, const float* lylx, const float* hi, long ldhi, int Chi,
                        const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldv & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(lerp4_cat_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), V, ldv, Clo, lylx, hi, ldhi,
                     Chi, pix, n, X, ldx);
  return arco_launch_status();
}
int arco_lerp4_cat_rows_bwd(const float* dX, long ldx, int Clo, const float* lylx, const int64_t* pix, long n, float* dV,
                            long ldv, float* dhi, long ldhi, int Chi, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (ldv & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(lerp4_cat_rows_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), dX, ldx, Clo, lylx, pix,
                     n, dV, ldv, dhi, ldhi, Chi);
  return arco_launch_status();
}

int arco_s2d3(float* V, long ldv, int NV, int X2, int Y2, int Z2, int C, float* P, long ldp, int dir, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldv & 3) == 0 && (ldp & 3) == 0);
  hipLaunchKernelGGL(s2d3_kernel, dim3(ew_grid((long)NV * X2 * Y2 * Z2 * 8 * (C / 4))), dim3(256), 0, as_stream(stream), V, ldv,
                     NV, X2, Y2, Z2, C, P, ldp, dir);
  return arco_launch_status();
}
// V[(n, 2x+dx, 2y+dy, 2z+dz)][c] = P[(n, x, y, z)][tap * C + c] + ADD[(same voxel)][c]   (tap = dx*4 + dy*2 + dz)
int arco_d2s3_add(const float* P, long ldp, int NV, int X2, int Y2, int Z2, int C, const float* ADD, long lda, float* V, long ldv, void* stream) {
  ARCO_CHECK_ARG(P && ADD && V && (C & 3) == 0 && (ldp & 3) == 0 && (lda & 3) == 0 && (ldv & 3) == 0);
  hipLaunchKernelGGL(d2s3_add_kernel<float>, dim3(ew_grid((long)NV * X2 * Y2 * Z2 * 8 * (C / 4))), dim3(256), 0, as_stream(stream), P, ldp,
                     NV, X2, Y2, Z2, C, ADD, lda, V, ldv);
  return arco_launch_status();
}
int arco_d2s3_add_h(const void* P, long ldp, int NV, int X2, int Y2, int Z2, int C, const void* ADD, long lda, void* V, long ldv, void* stream) {
  ARCO_CHECK_ARG(P && ADD && V && (C & 3) == 0 && (ldp & 3) == 0 && (lda & 3) == 0 && (ldv & 3) == 0);
  hipLaunchKernelGGL(d2s3_add_kernel<_Float16>, dim3(ew_grid((long)NV * X2 * Y2 * Z2 * 8 * (C / 4))), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const _Float16*>(P), ldp, NV, X2, Y2, Z2, C, reinterpret_cast<const _Float16*>(ADD), lda,
                     reinterpret_cast<_Float16*>(V), ldv);
  return arco_launch_status();
}
int arco_trilinear_fwd(const float* X, long ldx, int NV, int Di, int Hi, int Wi, int C, int Do, int Ho, int Wo, float* Y,
                       long ldy, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
  hipLaunchKernelGGL(trilinear_fwd_kernel, dim3(ew_grid((long)NV * Do * Ho * Wo * (C / 4))), dim3(256), 0, as_stream(stream),
                     X, ldx, NV, Di, Hi, Wi, C, Do, Ho, Wo, Y, ldy);
  return arco_launch_status();
}
int arco_trilinear_bwd(const float* dY, long ldy, int NV, int Di, int Hi, int Wi, int C, int Do, int Ho, int Wo, float* dX,
                       long ldx, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
  hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(ew_grid((long)NV * Di * Hi * Wi * (C / 4))), dim3(256), 0, as_stream(stream),
                     dY, ldy, NV, Di, Hi, Wi, C, Do, Ho, Wo, dX, ldx);
  return arco_launch_status();
}

int arco_gather_upcat_rows3d(const float* lo, long ldlo, int Clo, int Di, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                             int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldlo & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(gather_upcat_rows3d_kernel<float>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), lo, ldlo, Clo, Di, Hi, Wi,
                     hi, ldhi, Chi, Do, Ho, Wo, pix, n, X, ldx);
  return arco_launch_status();
}
// ... with the full-resolution map `hi` stored as f16 (f16 activation storage); lo, X fp32
int arco_gather_upcat_rows3d_h(const float* lo, long ldlo, int Clo, int Di, int Hi, int Wi, const void* hi, long ldhi, int Chi,
                               int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldlo & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(gather_upcat_rows3d_kernel<_Float16>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), lo, ldlo, Clo, Di, Hi, Wi,
                     reinterpret_cast<const _Float16*>(hi), ldhi, Chi, Do, Ho, Wo, pix, n, X, ldx);
  return arco_launch_status();
}
// rows of cat(trilinear blend of eight already-evaluated corner rows, hi[pix]) and the blend's adjoint (three-level 3-D head)
int arco_lerp8_cat_rows3d(const float* V, long ldv, int Clo, int Di, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                          int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldv & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(lerp8_cat_rows3d_kernel<float>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), V, ldv, Clo, Di, Hi, Wi,
                     hi, ldhi, Chi, Do, Ho, Wo, pix, n, X, ldx);
  return arco_launch_status();
}
int arco_lerp8_cat_rows3d_h(const float* V, long ldv, int Clo, int Di, int Hi, int Wi, const void* hi, long ldhi, int Chi,
                            int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (Chi & 3) == 0 && (ldv & 3) == 0 && (ldhi & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(lerp8_cat_rows3d_kernel<_Float16>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), V, ldv, Clo, Di, Hi, Wi,
                     reinterpret_cast<const _Float16*>(hi), ldhi, Chi, Do, Ho, Wo, pix, n, X, ldx);
  return arco_launch_status();
}
int arco_lerp8_rows3d_bwd(const float* dX, long ldx, int Clo, const float* w8, long n, float* dV, long ldv, void* stream) {
  ARCO_CHECK_ARG((Clo & 3) == 0 && (ldv & 3) == 0 && (ldx & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(lerp8_rows3d_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), dX, ldx, Clo, w8, n, dV, ldv);
  return arco_launch_status();
}
int arco_scatter_upcat_rows3d(const float* dX, long ldx, const int64_t* pix, long n, float* dlo, long ldlo, int Clo, int Di,
                              int Hi, int Wi, float* dhi, long ldhi, int Chi, int Do, int Ho, int Wo, void* stream) {
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(scatter_upcat_rows3d_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), dX, ldx, pix, n, dlo,
                     ldlo, Clo, Di, Hi, Wi, dhi, ldhi, Chi, Do, Ho, Wo);
  return arco_launch_status();
}

int arco_zero_rows(float* dst, long ld, int C, const int64_t* idx, long n, void* stream) {
  ARCO_CHECK_ARG(dst && idx && (C & 3) == 0 && (ld & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(zero_rows_kernel<float>, dim3(ew_grid(n * (C / 4))), dim3(256), 0, as_stream(stream), dst, ld, C, idx, n);
  return arco_launch_status();
}
int arco_zero_rows_h(void* dst, long ld, int C, const int64_t* idx, long n, void* stream) {
  ARCO_CHECK_ARG(dst && idx && (C & 3) == 0 && (ld & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(zero_rows_kernel<_Float16>, dim3(ew_grid(n * (C / 4))), dim3(256), 0, as_stream(stream), reinterpret_cast<_Float16*>(dst), ld, C, idx, n);
  return arco_launch_status();
}
int arco_cast_rows_f2h(const float* src, long ld_src, int C, const int64_t* idx, long n, float scale, void* dst, long ld_dst, void* stream) {
  ARCO_CHECK_ARG(src && dst && idx && (C & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0);
  if (n == 0) return ARCO_OK;
  hipLaunchKernelGGL(cast_rows_f2h_kernel, dim3(ew_grid(n * (C / 4))), dim3(256), 0, as_stream(stream), src, ld_src, C, idx, n, scale,
                     reinterpret_cast<_Float16*>(dst), ld_dst);
  return arco_launch_status();
}
int arco_fold_residual(const float* W, int n, int c, float* lo, float* hi, void* stream) {
  ARCO_CHECK_ARG(W && lo && hi && n > 0 && c > 0 && c < n);
  hipLaunchKernelGGL(fold_residual_kernel, dim3(ew_grid((long)n * n)), dim3(256), 0, as_stream(stream), W, n, c, lo, hi);
  return arco_launch_status();
}
int arco_unfold_residual(const float* dlo, const float* dhi, int n, int c, float* dW, void* stream) {
  ARCO_CHECK_ARG(dW && n > 0 && c > 0 && c < n);
  hipLaunchKernelGGL(unfold_residual_kernel, dim3(ew_grid((long)n * n)), dim3(256), 0, as_stream(stream), dlo, dhi, n, c, dW);
  return arco_launch_status();
}
// terms: host array of n (<= 8) device pointers to 0-d floats, weights: host array of n floats
int arco_combine_terms(const float* const* terms, const float* weights, int n, float* out, void* stream) {
  ARCO_CHECK_ARG(terms && weights && out && n >= 1 && n <= 8);
  TermTable t{}; t.n = n;
  for (int i = 0; i < n; ++i) { t.p[i] = terms[i]; t.w[i] = weights[i]; }
  hipLaunchKernelGGL(combine_terms_kernel, dim3(1), dim3(64), 0, as_stream(stream), t, out);
  return arco_launch_status();
}
int arco_combine_terms_bwd(const float* weights, int n, const float* g, float* grads, void* stream) {
  ARCO_CHECK_ARG(weights && g && grads && n >= 1 && n <= 8);
  TermTable t{}; t.n = n;
  for (int i = 0; i < n; ++i) t.w[i] = weights[i];
  hipLaunchKernelGGL(combine_terms_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), t, g, grads);
  return arco_launch_status();
}

int arco_copy_rows(const float* X, long ldx, long M, int C, float* Y, long ldy, int accumulate, void* stream) {
  ARCO_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, as_stream(stream), X, ldx, M, C, Y, ldy,
                     accumulate);
  return arco_launch_status();
}

int arco_nchw_to_nhwc(const float* X, int NB, int C, long P, float* Y, long ldy, void* stream) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((P + 31) / 32, (C + 31) / 32, NB), dim3(32, 8), 0, as_stream(stream), X,
                     NB, C, P, Y, ldy);
  return arco_launch_status();
}
int arco_nhwc_to_nchw(const float* X, long ldx, int NB, int C, long P, float* Y, void* stream) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((P + 31) / 32, (C + 31) / 32, NB), dim3(32, 8), 0, as_stream(stream), X,
                     ldx, NB, C, P, Y);
  return arco_launch_status();
}

int arco_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                      int first, void* stream) {
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, momentum,
                     weight_decay, first);
  return arco_launch_status();
}
// torch.optim.SGD(momentum, weight_decay, nesterov=False): the stage-1 trainers' optimizer (pretrain_2D.py:193-195)
int arco_sgd_momentum(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                      int first, void* stream) {
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, momentum,
                     weight_decay, (first ? 1 : 0) | 2);
  return arco_launch_status();
}
int arco_ema(float* k, const float* q, long n, float m, void* stream) {
  hipLaunchKernelGGL(ema_kernel, dim3(ew_grid(n)), dim3(256), 0, as_stream(stream), k, q, n, m);
  return arco_launch_status();
}

}  // extern "C"
