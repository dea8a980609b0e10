// Torch-free reproducer attempt for the two-queue hazard of round 4 (profiles/r05_notes.md section 1).
// Finding so far: in the failing steps the prologue of lerp4_cat_rows_bwd_kernel computes the two MIXED bilinear weights
//   v_pk_mul_f32 v[16:17], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]      (hy*lx | hx*ly)
//   v_pk_mul_f32 v[18:19], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]      (ly*hx | lx*hy)
// as 0 in lanes 48-63 of some waves while another pass's backward graph runs on a second queue.  This program runs that instruction
// sequence in a checking kernel on one stream while a second stream runs a co-tenant kernel (MFMA chain / LDS traffic / VMEM
// streaming), and counts wrong results by 16-lane group.
//   hipcc -O3 --offload-arch=gfx950 pkmul_repro.hip -o pkmul_repro && ./pkmul_repro [seconds per co-tenant]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <chrono>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the checking kernel: the product kernel's prologue, verbatim instruction forms, then a comparison with plain v_mul_f32 products
__global__ __launch_bounds__(256) void pk_check_kernel(const float* __restrict__ lylx, long n, unsigned* __restrict__ err /* [4 groups][4 weights] */,
                                                      float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  const f32x2* p = reinterpret_cast<const f32x2*>(lylx) + j;
  f32x2 l, h, w3, w0, wa, wb;
  asm volatile(
      "global_load_dwordx2 %0, %6, off\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "v_pk_add_f32 %1, %0, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_mul_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %3, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %4, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %5, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      : "=&v"(l), "=&v"(h), "=&v"(w3), "=&v"(w0), "=&v"(wa), "=&v"(wb)
      : "v"(p)
      : "memory");
  // reference products, one plain multiply each (operands from the registers the sequence left)
  float r3, r0, r1, r2;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r3) : "v"(l[0]), "v"(l[1]));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(h[0]), "v"(h[1]));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(h[0]), "v"(l[1]));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(l[0]), "v"(h[1]));
  const int g = lane >> 4;
  if (w3[0] != r3) atomicAdd(err + g * 4 + 3, 1u);
  if (w0[0] != r0) atomicAdd(err + g * 4 + 0, 1u);
  if (wa[0] != r1) atomicAdd(err + g * 4 + 1, 1u);
  if (wb[0] != r2) atomicAdd(err + g * 4 + 2, 1u);
  if (wa[0] == 0.f && r1 != 0.f) atomicAdd(err + 16 + g, 1u);       // the signature: an exact zero
  if (wb[0] == 0.f && r2 != 0.f) atomicAdd(err + 16 + g, 1u);
  if (lane == 0 && sink) sink[j] = w0[0] + wa[0] + wb[0] + w3[0];
}

// co-tenants
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 3 * i)); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc3, 0, 0, 0);
  }
  const f32x4 s = acc0 + acc1 + acc2 + acc3;
  if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}
__global__ __launch_bounds__(256) void lds_spin_kernel(float* out, int iters) {
  __shared__ float sh[256 * 33];
  for (int i = threadIdx.x; i < 256 * 33; i += 256) sh[i] = i;
  __syncthreads();
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x4 v = *reinterpret_cast<const f32x4*>(sh + ((threadIdx.x * 4 + it * 64) & 8191));
    s += v[0] + v[1] + v[2] + v[3];
    sh[(threadIdx.x + it) & 8191] = s;
  }
  if (s == 12345.f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void vmem_stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = in[i] * 1.0001f;
}
__global__ __launch_bounds__(256) void valu_pk_spin_kernel(float* out, int iters) {      // packed-fp32 VALU traffic from another wave
  f32x2 a = {1.0001f + threadIdx.x * 1e-6f, 0.9999f}, b = {0.99995f, 1.00005f};
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  if (a[0] == 12345.f) out[threadIdx.x] = a[1];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const long n = 16384;                                        // waves of the checking kernel per launch (product: 1024 / 4096)
  std::vector<float> h(2 * n);
  srand(1);
  for (long i = 0; i < 2 * n; ++i) h[i] = (float)(rand() % 127 + 1) / 128.f * 0.999f;
  float *lylx, *sink, *junk; unsigned* err; f32x4 *big_in, *big_out;
  CK(hipMalloc(&lylx, 2 * n * 4)); CK(hipMalloc(&sink, n * 4)); CK(hipMalloc(&junk, 4096)); CK(hipMalloc(&err, 32 * 4));
  const long n4 = 64l << 20;
  CK(hipMalloc(&big_in, n4 * 16)); CK(hipMalloc(&big_out, n4 * 16)); CK(hipMemset(big_in, 0, n4 * 16));
  CK(hipMemcpy(lylx, h.data(), 2 * n * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const char* names[] = {"none", "mfma chain", "lds traffic", "vmem streaming", "packed-fp32 valu", "mfma + vmem"};
  for (int tenant = 0; tenant < 6; ++tenant) {
    CK(hipMemset(err, 0, 32 * 4));
    CK(hipDeviceSynchronize());
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      for (int rep = 0; rep < 20; ++rep) {
        if (tenant == 1 || tenant == 5) hipLaunchKernelGGL(mfma_spin_kernel, dim3(1024), dim3(256), 0, sb, junk, 2000);
        if (tenant == 2) hipLaunchKernelGGL(lds_spin_kernel, dim3(512), dim3(256), 0, sb, junk, 4000);
        if (tenant == 3 || tenant == 5) hipLaunchKernelGGL(vmem_stream_kernel, dim3(2048), dim3(256), 0, sb, big_in, big_out, n4 / 16);
        if (tenant == 4) hipLaunchKernelGGL(valu_pk_spin_kernel, dim3(1024), dim3(256), 0, sb, junk, 20000);
        for (int k = 0; k < 8; ++k) {
          hipLaunchKernelGGL(pk_check_kernel, dim3((n + 3) / 4), dim3(256), 0, sa, lylx, n, err, sink);
          ++launches;
        }
      }
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    }
    unsigned he[32];
    CK(hipMemcpy(he, err, sizeof(he), hipMemcpyDeviceToHost));
    unsigned long tot = 0; for (int i = 0; i < 16; ++i) tot += he[i];
    printf("co-tenant %-18s: %ld checking launches (%ld waves), wrong weights %lu; by 16-lane group x (w0,w1,w2,w3): ", names[tenant], launches, launches * n, tot);
    for (int g = 0; g < 4; ++g) printf("[%u %u %u %u] ", he[g * 4], he[g * 4 + 1], he[g * 4 + 2], he[g * 4 + 3]);
    printf(" exact zeros by group: [%u %u %u %u]\n", he[16], he[17], he[18], he[19]);
    fflush(stdout);
  }
  return 0;
}
