// Issue cost of a few VALU instruction classes on gfx950, one wave per SIMD and three waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(unsigned* out, int iters) {
  unsigned a[8]; float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; f[i] = (float)a[i]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 2) asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (KIND == 4) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(a[i]));
      if (KIND == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(*(unsigned long long*)&a[i & 6]) : "v"(a[7]) : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(t1 - t0);
}
template <int KIND> void run(const char* nm) {
  unsigned* d; hipMalloc(&d, 1 << 20);
  for (int threads : {256, 768}) {
    k<KIND><<<1, threads>>>(d, 4000); hipDeviceSynchronize();
    unsigned h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-16s %d waves/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD issue\n", nm, threads / 256, h / (4000.0 * 8), h / (4000.0 * 8) / (threads / 256));
  }
  hipFree(d);
}
int main() { run<0>("v_fma_f32"); run<1>("v_mul_lo_u32"); run<2>("v_mul_u32_u24"); run<3>("v_exp_f32"); run<4>("v_xor_b32"); run<5>("v_mad_u64_u32"); }
