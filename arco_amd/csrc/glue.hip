// Step glue of the trainer (train_arco_2d.py:284-286,342-393,492-498):
// class softmax / max / argmax / entropy, one-hot labels, exact np.percentile (linear) of the
// entropy via radix select, and the low/high entropy masks.  HBM-bound, one pixel per lane.
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
  unsigned long long c = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) c += lab[i] >= 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&st->n_valid, c);
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
__global__ void sel_pick_kernel(int pass, SelState* st) {
  const int r = threadIdx.x;
  if (r < 4) {
    const int shift = 24 - 8 * pass;
    unsigned long long k = st->k[r], run = 0;
    int d = 0;
    for (; d < 256; ++d) { const unsigned long long c = st->hist[r][d]; if (run + c > k) break; run += c; }
    if (d > 255) d = 255;
    st->k[r] = k - run;
    st->prefix[r] |= ((unsigned int)d) << shift;
  }
  __syncthreads();
  if (r < 4) for (int d = 0; d < 256; ++d) st->hist[r][d] = 0;
  if (pass == 3 && r < 2) {      // numpy _lerp in float64
    const double a = (double)fkey_inv(st->prefix[2 * r]), b = (double)fkey_inv(st->prefix[2 * r + 1]);
    const double t = st->gamma[r], diff = b - a;
    st->thr[r] = t >= 0.5 ? b - diff * (1.0 - t) : a + diff * t;
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

static inline int gl_grid(long work) { long g = (work + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1; return (int)g; }

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

long arco_sel_state_bytes() { return (long)sizeof(SelState); }

// masks from the entropy percentiles q_lo / q_hi over pixels with lab_u >= 0; state = arco_sel_state_bytes() scratch
int arco_entropy_masks(const float* ent, const int64_t* lab_l, const int64_t* lab_u, long n_l, long n_u, double q_lo,
                       double q_hi, void* state, float* low, float* high, void* stream) {
  ARCO_CHECK_ARG(n_u > 0 && state);
  hipStream_t st = as_stream(stream);
  SelState* s = reinterpret_cast<SelState*>(state);
  (void)hipMemsetAsync(s, 0, sizeof(SelState), st);
  hipLaunchKernelGGL(sel_count_kernel, dim3(gl_grid(n_u)), dim3(256), 0, st, lab_u, n_u, s);
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, st, s, q_lo, q_hi);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(gl_grid(n_u) > 256 ? 256 : gl_grid(n_u)), dim3(256), 0, st, ent, lab_u, n_u, pass, s);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(64), 0, st, pass, s);
  }
  hipLaunchKernelGGL(entropy_masks_kernel, dim3(gl_grid(n_l + n_u)), dim3(256), 0, st, ent, lab_l, lab_u, n_l, n_u, s,
                     low, high);
  return arco_launch_status();
}

}  // extern "C"
