// Semantics probe of ds_read_b64_tr_b16 on gfx950: LDS holds a row-major [16 rows][64 cols] uint16 image with
// value = row * 64 + col.  Every lane of a 16-lane group g passes the address of row (4j + (i >> 2)), cols 4 (i & 3) .. +3
// of block (rows 4j..4j+3, cols 16c..16c+15); prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) img[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // group g reads the block rows 4g..4g+3, cols 0..15: lane i supplies row 4g + (i >> 2), cols 4 (i & 3)
  const unsigned addr = (unsigned)(size_t)(img + (4 * g + (i >> 2)) * 64 + 4 * (i & 3));
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)(v >> (16 * e));
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[4*l]/64, h[4*l]%64, h[4*l+1]/64, h[4*l+1]%64, h[4*l+2]/64, h[4*l+2]%64, h[4*l+3]/64, h[4*l+3]%64);
  return 0;
}
