// batch_transform's per-image photometric augmentation on the GPU (code/augment.py:148-225, 255-281): the reference
// converts every image to an 8-bit PIL image, applies torchvision's ColorJitter (brightness / contrast / saturation /
// hue in a random order) and PIL's GaussianBlur, and converts back to a float tensor.  These kernels reproduce the
// 8-bit INTEGER arithmetic of Pillow (ImageEnhance = Image.blend with a degenerate image, ImagingConvert's L / HSV
// formulas, ImagingGaussianBlur = three box-blur passes per direction with 24-bit fixed-point weights) - pinned bit for
// bit against Pillow itself (tests/golden/g12_jitter.npz, oracle/gen_golden.py g12).
#include "common.h"
// Pillow rounds every float operation separately: no fused multiply-add in this file
#pragma clang fp contract(off)

#define ARCO_MAX_IMG 32
struct JitDesc {
  int order[4];        // torchvision ColorJitter.get_params: permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue}
  float f[4];          // factor of op id 0..3 (hue: the hue_factor in [-0.5, 0.5])
  int jitter, blur;    // flags
  float blur_r;        // fractional box radius of the blur (host: _gaussian_blur_radius(sigma, 3 passes))
};
struct JitTable { JitDesc d[ARCO_MAX_IMG]; };

__device__ __forceinline__ int q8(float x) {          // to_pil_image: pic.mul(255).byte()
  const float t = x * 255.f;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int blend8(int d, int v, float f) {        // ImagingBlend outside / inside [0, 1]: truncation, clipped
  const float t = (float)d + f * (float)(v - d);
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int lum8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // convert("L")
__device__ __forceinline__ void rgb2hsv8(int r, int g, int b, int& uh, int& us, int& uv) {      // ImagingConvert rgb2hsv_row
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
  float h;
  if (r == maxc) h = (float)((double)bc - (double)gc);
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih); us = is < 0 ? 0 : (is > 255 ? 255 : is);
}
__device__ __forceinline__ int rnd8(double t) { const int v = (int)rint(t); return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ void hsv2rgb8(int h, int s, int v, int& r, int& g, int& b) {          // ImagingConvert hsv2rgb_row
  if (s == 0) { r = g = b = v; return; }
  const float fs = (float)s / 255.0f;
  const float hh = (float)h * 6.0f / 255.0f;
  const float fi = floorf(hh), f = hh - fi;
  const int p = rnd8((double)v * (double)(1.0f - fs)), q = rnd8((double)v * (double)(1.0f - fs * f)),
            t = rnd8((double)v * (double)(1.0f - fs * (1.0f - f)));
  switch (((int)fi) % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}
// apply ops order[0 .. n_ops) of image d to one pixel (C = 1 or 3 channels in px[])
__device__ __forceinline__ void jitter_px(const JitDesc& d, int n_ops, int C, int mean, int* px) {
  for (int sidx = 0; sidx < n_ops; ++sidx) {
    const int op = d.order[sidx];
    const float f = d.f[op];
    if (op == 0) {
      for (int c = 0; c < C; ++c) px[c] = blend8(0, px[c], f);
    } else if (op == 1) {
      for (int c = 0; c < C; ++c) px[c] = blend8(mean, px[c], f);
    } else if (op == 2 && C == 3) {
      const int l = lum8(px[0], px[1], px[2]);
      for (int c = 0; c < 3; ++c) px[c] = blend8(l, px[c], f);
    } else if (op == 3 && C == 3) {        // torchvision F_pil.adjust_hue: H channel += uint8(hue_factor * 255) with wrap
      int h, s, v;
      rgb2hsv8(px[0], px[1], px[2], h, s, v);
      h = (h + ((int)(f * 255.f) & 0xff)) & 0xff;
      hsv2rgb8(h, s, v, px[0], px[1], px[2]);
    }                                        // single-channel images: saturation and hue leave the image unchanged
  }
}
// sums[i] = sum over the pixels of image i of the luminance AFTER the ops that precede contrast (ImageStat mean of
// image.convert("L") inside ImageEnhance.Contrast); exact integer atomics
__global__ void jitter_lum_sum_kernel(const float* __restrict__ data, int B, int C, long HW, JitTable t, unsigned long long* __restrict__ sums) {
  const int i = blockIdx.y;
  const JitDesc& d = t.d[i];
  if (!d.jitter) return;
  int n_before = 0;
  while (n_before < 4 && d.order[n_before] != 1) ++n_before;
  unsigned long long acc = 0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    int px[3];
    for (int c = 0; c < C; ++c) px[c] = q8(data[((long)i * C + c) * HW + p]);
    jitter_px(d, n_before, C, 0, px);
    acc += (unsigned)(C == 3 ? lum8(px[0], px[1], px[2]) : px[0]);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sums[i], acc);
}
// q[i][c][p] = the 8-bit image after quantisation (+ jitter), as floats 0..255
__global__ void jitter_apply_kernel(const float* __restrict__ data, int B, int C, long HW, JitTable t, const unsigned long long* __restrict__ sums,
                                    float* __restrict__ q) {
  const int i = blockIdx.y;
  const JitDesc& d = t.d[i];
  const int mean = d.jitter ? (int)((double)sums[i] / (double)HW + 0.5) : 0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    int px[3];
    for (int c = 0; c < C; ++c) px[c] = q8(data[((long)i * C + c) * HW + p]);
    if (d.jitter) jitter_px(d, 4, C, mean, px);
    for (int c = 0; c < C; ++c) q[((long)i * C + c) * HW + p] = (float)px[c];
  }
}
// ImagingGaussianBlur (3 box passes per direction) on 8-bit planes + to_tensor (/255).  One block = one 32 x 32 output tile
// of one (image, channel) plane; every pass clamps its neighbour indices to the IMAGE (edge replication of that pass' input).
template <int R>       // integer part of the box radius
__global__ __launch_bounds__(256) void blur_to_tensor_kernel(const float* __restrict__ q, int B, int C, int H, int W, JitTable t,
                                                            float* __restrict__ out) {
  constexpr int T = 32, HH = 3 * (R + 1), TS = T + 2 * HH;
  __shared__ int a[TS * TS], b[TS * TS];
  const int plane = blockIdx.z, i = plane / C;
  const JitDesc& d = t.d[i];
  const int y0 = blockIdx.y * T, x0 = blockIdx.x * T;
  const float* src = q + (long)plane * H * W;
  float* dst = out + (long)plane * H * W;
  if (d.blur < 0) return;                 // this plane belongs to the launch of another box-radius class
  if (!d.blur) {
    for (int u = threadIdx.x; u < T * T; u += 256) {
      const int y = y0 + u / T, x = x0 + u % T;
      if (y < H && x < W) dst[(long)y * W + x] = src[(long)y * W + x] / 255.0f;
    }
    return;
  }
  const float fr = d.blur_r;
  const unsigned ww = (unsigned)((float)(1 << 24) / (fr * 2.f + 1.f));
  const unsigned fw = ((1u << 24) - (2 * R + 1) * ww) / 2;
  for (int u = threadIdx.x; u < TS * TS; u += 256) {
    const int gy = y0 - HH + u / TS, gx = x0 - HH + u % TS;
    a[u] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (int)src[(long)gy * W + gx] : 0;
  }
  __syncthreads();
  int* in = a; int* ou = b;
  for (int pass = 0; pass < 6; ++pass) {
    const bool horiz = pass < 3;
    for (int u = threadIdx.x; u < TS * TS; u += 256) {
      const int ty = u / TS, tx = u % TS, gy = y0 - HH + ty, gx = x0 - HH + tx;
      int v = 0;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        unsigned long long acc = 0;
        bool ok = true;
        for (int dd = -R - 1; dd <= R + 1; ++dd) {
          int cy = gy, cx = gx;
          if (horiz) cx = min(max(gx + dd, 0), W - 1); else cy = min(max(gy + dd, 0), H - 1);
          const int sy = cy - (y0 - HH), sx = cx - (x0 - HH);
          if (sy < 0 || sy >= TS || sx < 0 || sx >= TS) { ok = false; break; }      // beyond this tile's halo: value unused
          acc += (unsigned long long)in[sy * TS + sx] * ((dd == -R - 1 || dd == R + 1) ? fw : ww);
        }
        v = ok ? (int)((acc + (1u << 23)) >> 24) : 0;
      }
      ou[u] = v;
    }
    __syncthreads();
    int* tmp = in; in = ou; ou = tmp;
  }
  for (int u = threadIdx.x; u < T * T; u += 256) {
    const int y = y0 + u / T, x = x0 + u % T;
    if (y < H && x < W) dst[(long)y * W + x] = (float)in[(u / T + HH) * TS + u % T + HH] / 255.0f;
  }
}
// out = floor(x * 255) / 255 (to_pil_image + to_tensor of the confidence map, augment.py:219-225,196-197)
__global__ void quantize8_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)q8(x[i]) / 255.0f;
}

extern "C" {
long arco_jitter_desc_bytes() { return (long)sizeof(JitDesc); }
/* data / out: [B, C, H, W] fp32 (C = 1 or 3); desc: HOST array of B records {int order[4]; float f[4]; int jitter, blur;
   float blur_r;}; ws: B uint64 + B*C*H*W floats.  out = to_tensor(GaussianBlur(ColorJitter(to_pil_image(data)))) per image. */
int arco_jitter_blur(const float* data, int B, int C, int H, int W, const void* desc_host, void* ws, float* out, void* stream) {
  ARCO_CHECK_ARG(data && desc_host && ws && out && B >= 1 && B <= ARCO_MAX_IMG && (C == 1 || C == 3) && H > 0 && W > 0);
  JitTable t;
  const JitDesc* dh = reinterpret_cast<const JitDesc*>(desc_host);
  int rmax = 0;
  bool any_contrast = false;
  for (int i = 0; i < B; ++i) {
    t.d[i] = dh[i];
    if (dh[i].blur) { ARCO_CHECK_ARG(dh[i].blur_r >= 0.f && dh[i].blur_r < 3.f); if ((int)dh[i].blur_r > rmax) rmax = (int)dh[i].blur_r; }
    if (dh[i].jitter) any_contrast = true;
  }
  unsigned long long* sums = reinterpret_cast<unsigned long long*>(ws);
  float* q = reinterpret_cast<float*>(sums + B);
  const long HW = (long)H * W;
  hipStream_t st = as_stream(stream);
  if (any_contrast) {
    (void)hipMemsetAsync(sums, 0, sizeof(unsigned long long) * B, st);
    hipLaunchKernelGGL(jitter_lum_sum_kernel, dim3(64, B), dim3(256), 0, st, data, B, C, HW, t, sums);
  }
  hipLaunchKernelGGL(jitter_apply_kernel, dim3(128, B), dim3(256), 0, st, data, B, C, HW, t, sums, q);
  // one box radius class per launch: images whose integer radius differs from the launch's are handled by their own launch
  for (int r = 0; r <= rmax; ++r) {
    JitTable tr = t;
    for (int i = 0; i < B; ++i) if (r > 0 ? !(t.d[i].blur && (int)t.d[i].blur_r == r) : (t.d[i].blur && (int)t.d[i].blur_r != 0)) tr.d[i].blur = -1;
    dim3 grid((W + 31) / 32, (H + 31) / 32, B * C);
    // blur == -1: not this launch's image (skipped below by passing a table where it is marked)
    if (r == 0) hipLaunchKernelGGL(blur_to_tensor_kernel<0>, grid, dim3(256), 0, st, q, B, C, H, W, tr, out);
    else if (r == 1) hipLaunchKernelGGL(blur_to_tensor_kernel<1>, grid, dim3(256), 0, st, q, B, C, H, W, tr, out);
    else hipLaunchKernelGGL(blur_to_tensor_kernel<2>, grid, dim3(256), 0, st, q, B, C, H, W, tr, out);
  }
  return arco_launch_status();
}
int arco_quantize8(const float* x, long n, float* out, void* stream) {
  ARCO_CHECK_ARG(x && out && n >= 0);
  if (n == 0) return ARCO_OK;
  long g = (n + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(quantize8_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x, n, out);
  return arco_launch_status();
}
}  // extern "C"
