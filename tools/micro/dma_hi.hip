// Does an LDS-DMA (global_load_lds_dwordx4) reach LDS addresses >= 64 KiB on gfx950?  Writes a 1 KB piece at several LDS byte
// addresses through M0 and reads it back with ds_read.  Build: hipcc --offload-arch=gfx950 -O3 -w tools/micro/dma_hi.hip -o egt_amd/lib/var/dma_hi
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(64) k(const float* src, float* out, unsigned lds_byte) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 40960; i += 64) sm[i] = -1.0f;
  __syncthreads();
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "s"(src), "v"(lane * 16), "s"(lds_byte) : "memory");
  __syncthreads();
  // where did it land?  report the first LDS dword that holds src[0] (= 1000) and the value at the requested address
  int found = -1;
  for (int i = 0; i < 40960; ++i) if (sm[i] == 1000.0f) { found = i * 4; break; }
  if (lane == 0) { out[0] = (float)found; out[1] = sm[lds_byte / 4]; out[2] = sm[lds_byte / 4 + 255]; }
}
int main() {
  float *src, *out, h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 1000.0f + i;
  hipMalloc(&src, 4096); hipMalloc(&out, 64);
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (unsigned addr : {0u, 32768u, 65536u - 1024u, 65536u, 98304u, 131072u, 160u * 1024u - 1024u}) {
    k<<<1, 64, 163840>>>(src, out, addr);
    float r[3];
    hipMemcpy(r, out, 12, hipMemcpyDeviceToHost);
    printf("requested LDS byte %6u: first hit at byte %6.0f, value there %.0f, last dword %.0f (expect 1000 / 1255)\n", addr, r[0], r[1], r[2]);
  }
  return 0;
}
