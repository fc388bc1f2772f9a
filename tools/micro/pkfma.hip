// Micro-benchmark: throughput of v_fma_f32 vs v_pk_fma_f32 on gfx950 (flops per cycle per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/micro/pkfma.hip -o tools/micro/pkfma.bin ; gpurun -- ./tools/micro/pkfma.bin
// MI355X result (round 1): v_fma_f32 99 TFLOP/s, v_pk_fma_f32 115 TFLOP/s with 16 dependent chains per lane: a wave64
// VALU op issues over 2 cycles here, plain FMA already runs at the 157 TFLOP/s vector rate -- packing buys ~15 %.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  float x[16];
  v2f y[8];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) y[i] = (v2f){x[2 * i], x[2 * i + 1]};
  const v2f av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(av), "v"(bv));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 2048;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = 2.0 * 16 * iters * (double)grid * 256;
      if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32  ", ms, fl / ms / 1e9);
    }
  }
  return 0;
}
