// Issue rate of a realistic fp32 MFMA stream on one wave per SIMD: NA distinct A registers x NB distinct B registers feeding NACC
// independent accumulators (the S / P.V phases of egt_attn_mfma.hip), bare and with memory operations interleaved.
// Prints s_memtime cycles per MFMA.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_stream.hip -o egt_amd/lib/var/mfma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: 16x16x4, 4 accumulators, 16 A x 16 B registers; 1: same, 2 accumulators; 2: same, 8 accumulators;
// 3: 32x32x2 2 accumulators; 4: MODE 0 + one global_load_dwordx4 per 8 MFMAs; 5: MODE 0 + one ds_read_b128 per 4 MFMAs
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(int iters, const float* src, unsigned long long* out) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63;
  float A[16], B[16];
  for (int i = 0; i < 16; ++i) { A[i] = src[lane + 64 * i]; B[i] = src[1024 + lane + 64 * i]; }
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
  v16f big[2];
  for (int i = 0; i < 16; ++i) { big[0][i] = 0; big[1][i] = 0; }
  float4 ld = make_float4(0, 0, 0, 0), lds4 = ld;
  sm[threadIdx.x * 4] = lane;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      constexpr int NACC = MODE == 1 ? 2 : MODE == 2 ? 8 : 4;
      if (MODE == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i & 15], B[(i >> 2) & 15], big[i & 1], 0, 0, 0);
      else acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(i >> 1) & 15], B[(i * 5) & 15], acc[i % NACC], 0, 0, 0);
      if (MODE == 4 && (i & 7) == 0) { float4 t = *reinterpret_cast<const float4*>(src + ((it * 64 + i) & 1023) * 256 + lane * 4); ld.x += t.x; }
      if (MODE == 5 && (i & 3) == 0) { float4 t = *reinterpret_cast<const float4*>(sm + ((lane * 4 + i * 16) & 1023)); lds4.x += t.x; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = ld.x + lds4.x;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
  r += big[0][0] + big[1][5];
  if (r == 123.456f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// MODE 6 / 7: 512-thread workgroups: waves 0-3 (one per SIMD) run the MODE 5 stream (MFMA + ds_read_b128), waves 4-7 are LOADERS that
// issue LDS-DMA pieces (global_load_lds_dwordx4, 1 KB each; MODE 6) or register loads + ds_write_b128 (MODE 7) back to back:
// does a loader wave on the same SIMD slow the compute wave's MFMA stream?
template <int MODE>
__global__ void __launch_bounds__(512, 1) k2(int iters, const float* src, unsigned long long* out) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  sm[threadIdx.x * 4] = lane;
  __syncthreads();
  if (wave < 4) {
    float A[16], B[16];
    for (int i = 0; i < 16; ++i) { A[i] = src[lane + 64 * i]; B[i] = src[1024 + lane + 64 * i]; }
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float4 lds4 = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(i >> 1) & 15], B[(i * 5) & 15], acc[i & 3], 0, 0, 0);
        if ((i & 3) == 0) { float4 t = *reinterpret_cast<const float4*>(sm + ((lane * 4 + i * 16) & 1023)); lds4.x += t.x; }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = lds4.x;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
    if (r == 123.456f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  } else {
    // loaders: 16 pieces of 1 KB per 64-MFMA trip of the compute waves (= 64 KB per CU and trip)
    const unsigned ldsbase = 16384 + (wave - 4) * 16384;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float* g = src + (size_t)((it * 16 + j + blockIdx.x * 7) & 1023) * 256;
        if (MODE == 6) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "s"(g), "v"(lane * 16), "s"(ldsbase + j * 1024) : "memory");
        } else {
          const float4 t = *reinterpret_cast<const float4*>(g + lane * 4);
          *reinterpret_cast<float4*>(sm + (ldsbase >> 2) + j * 256 + lane * 4) = t;
        }
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 1.5f) out[2] = 1;
  }
}
template <int MODE>
static void run2(const char* name, const float* src, unsigned long long* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k2<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  k2<MODE><<<256, 512, 96 * 1024>>>(10, src, d);
  hipEventRecord(e0);
  k2<MODE><<<256, 512, 96 * 1024>>>(iters, src, d);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double n = iters * 64.0;
  printf("%-58s %.1f cycles / MFMA (compute wave), kernel %.2f ns / MFMA, %.1f TF, loader stream %.2f TB/s\n", name, h / n, ms * 1e6 / n,
         2048.0 * n * 1024 / (ms * 1e-3) / 1e12, iters * 16.0 * 1024 * 4 * 256 / (ms * 1e-3) / 1e12);
}
template <int MODE>
static void run(const char* name, const float* src, unsigned long long* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 256, 16384>>>(10, src, d);
  hipEventRecord(e0);
  k<MODE><<<256, 256, 16384>>>(iters, src, d);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double n = iters * 64.0;
  printf("%-58s %.1f cycles / MFMA, %.2f ns / MFMA, clock %.2f GHz, %.1f TF\n", name, h / n, ms * 1e6 / n, h / (ms * 1e6),
         (MODE == 3 ? 4096.0 : 2048.0) * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
  unsigned long long* d; float* src;
  hipMalloc(&d, 64); hipMalloc(&src, 1024 * 256 * 4 + 4096);
  hipMemset(src, 0, 1024 * 256 * 4 + 4096);
  run<0>("16x16x4 4 acc, 16 A x 16 B registers", src, d);
  run<1>("16x16x4 2 acc", src, d);
  run<2>("16x16x4 8 acc", src, d);
  run<3>("32x32x2 2 acc", src, d);
  run<4>("16x16x4 4 acc + 1 global_load_dwordx4 / 8 MFMA", src, d);
  run<5>("16x16x4 4 acc + 1 ds_read_b128 / 4 MFMA", src, d);
  run2<6>("MODE 5 + loader waves: LDS-DMA 64 KB / trip / CU", src, d);
  run2<7>("MODE 5 + loader waves: register loads + ds_write_b128", src, d);
  return 0;
}
