// How many independent VALU instructions of the SAME wave issue in the shadow of one MFMA?  One wave per SIMD (256-thread
// workgroups, one per CU); loop body = 1 MFMA + K x v_fma_f32 (inline asm: no SLP packing, 8 independent chains).
// Prints cycles per iteration (s_memtime) for K = 0..16 and three MFMA shapes.  If the matrix pipe ran beside the VALU the
// curve would stay flat until K x 4 cycles exceeds the MFMA's issue interval; if the MFMA holds the VALU it rises from K = 1.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o egt_amd/lib/var/mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int K>
__global__ void __launch_bounds__(256, 1) k(int iters, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  v4f a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v16f b[2];
  for (int i = 0; i < 16; ++i) { b[0][i] = 0; b[1][i] = 0; }
  float x = lane * 0.001f, y = 1.0f;
  bf8 xb, yb;
  for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(lane * 0.01f); yb[j] = (__bf16)1.0f; }
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = lane + j;
  const float m = 1.0001f, c = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // 4 independent accumulators: the MFMA stream is issue-bound, not latency-bound
      if (SHAPE == 0) a[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[u], 0, 0, 0);
      if (SHAPE == 1) a[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a[u], 0, 0, 0);
      if (SHAPE == 2) b[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, b[u & 1], 0, 0, 0);
      if (SHAPE == 3) b[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, b[u & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j & 7]) : "v"(m), "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int j = 0; j < 8; ++j) r += f[j];
  for (int u = 0; u < 4; ++u) r += a[u][0];
  r += b[0][0] + b[1][3];
  if (r == 123.456f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int SHAPE, int K>
static double run(unsigned long long* d) {
  const int iters = 20000;
  k<SHAPE, K><<<256, 256>>>(iters, d);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  return (double)h / (iters * 4.0);
}
template <int SHAPE>
static void sweep(const char* name, unsigned long long* d) {
  printf("%-24s cycles per (1 MFMA + K v_fma):", name);
  printf(" K=0 %.1f", run<SHAPE, 0>(d));
  printf(" | 1 %.1f", run<SHAPE, 1>(d));
  printf(" | 2 %.1f", run<SHAPE, 2>(d));
  printf(" | 4 %.1f", run<SHAPE, 4>(d));
  printf(" | 6 %.1f", run<SHAPE, 6>(d));
  printf(" | 8 %.1f", run<SHAPE, 8>(d));
  printf(" | 12 %.1f", run<SHAPE, 12>(d));
  printf(" | 16 %.1f\n", run<SHAPE, 16>(d));
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  sweep<0>("f32 16x16x4", d);
  sweep<1>("bf16 16x16x32", d);
  sweep<2>("bf16 32x32x16", d);
  sweep<3>("f32 32x32x2", d);
  return 0;
}
