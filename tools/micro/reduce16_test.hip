// Checks the DPP bank-masked transposed reduction (egt_block_dev.h: reduce16_keep_own) bit for bit against a
// __shfl-based reference on random data.  Build: hipcc --offload-arch=gfx950 -O3 -I egt_amd/csrc -I include ...
#include "egt_common.h"
#include "egt_block_dev.h"
#include <stdlib.h>
#include <vector>

__global__ void k(const float* in, float* out_new, float* out_ref) {
  const int lane = threadIdx.x & 63, p = lane & 15;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[(blockIdx.x * 64 + lane) * 16 + i];
  out_new[blockIdx.x * 64 + lane] = reduce16_keep_own(v, p);
  // reference: same association order as the tree (xor 8, 4, 2, 1 with own-first adds)
  float w8[8], w4[4], w2[2];
  const bool b3 = p & 8, b2 = p & 4, b1 = p & 2, b0 = p & 1;
  for (int i = 0; i < 8; ++i) { const float snd = b3 ? v[i] : v[i + 8], kp = b3 ? v[i + 8] : v[i]; w8[i] = kp + __shfl_xor(snd, 8, 64); }
  for (int i = 0; i < 4; ++i) { const float snd = b2 ? w8[i] : w8[i + 4], kp = b2 ? w8[i + 4] : w8[i]; w4[i] = kp + __shfl_xor(snd, 4, 64); }
  for (int i = 0; i < 2; ++i) { const float snd = b1 ? w4[i] : w4[i + 2], kp = b1 ? w4[i + 2] : w4[i]; w2[i] = kp + __shfl_xor(snd, 2, 64); }
  const float snd = b0 ? w2[0] : w2[1], kp = b0 ? w2[1] : w2[0];
  out_ref[blockIdx.x * 64 + lane] = kp + __shfl_xor(snd, 1, 64);
}

int main() {
  const int nb = 64, n = nb * 64;
  std::vector<float> h(n * 16), a(n), b(n);
  srand(1);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *d, *o1, *o2;
  (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&o1, n * 4); (void)hipMalloc(&o2, n * 4);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  k<<<nb, 64>>>(d, o1, o2);
  (void)hipMemcpy(a.data(), o1, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(b.data(), o2, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (a[i] != b[i]) { if (bad < 5) printf("lane %d: new %.9g ref %.9g\n", i & 63, a[i], b[i]); ++bad; }
  printf("reduce16_keep_own: %d / %d lanes differ from the reference\n", bad, n);
  return bad != 0;
}
