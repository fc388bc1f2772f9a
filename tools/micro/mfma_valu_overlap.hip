// Do fp32-input MFMAs (v_mfma_f32_16x16x4_f32) overlap with another wave's VALU work on the SAME SIMD?
// One 512-thread workgroup per CU (two waves per SIMD); even waves run role X, odd waves role Y; roles: 0 idle, 1 f32 MFMA,
// 2 bf16 MFMA, 3 VALU fma chains, 4 LDS reads.  Prints the time of X alone, Y alone and both.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o egt_amd/lib/var/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512, 2) k(int roleE, int roleO, int iters, float* out) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // waves go to SIMDs round robin: waves w and w+4 share a SIMD; give them different roles
  const int role = (wave < 4) ? roleE : roleO;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  float r = 0.f;
  if (role == 1) {
    v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (role == 2) {
    v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(lane * 0.01f); y[j] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (role == 3) {
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // 32 independent-ish FMAs per iteration
        f0 = fmaf(f0, m, c); f1 = fmaf(f1, m, c); f2 = fmaf(f2, m, c); f3 = fmaf(f3, m, c);
        f4 = fmaf(f4, m, c); f5 = fmaf(f5, m, c); f6 = fmaf(f6, m, c); f7 = fmaf(f7, m, c);
      }
    }
    r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  } else if (role == 5 || role == 6) {   // ONE wave: 4 MFMAs (5: f32, 6: bf16) interleaved with 32 VALU fmas per iteration
    v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 0.001f, y = 1.0f;
    bf8 xb, yb;
    for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(lane * 0.01f); yb[j] = (__bf16)1.0f; }
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
#define MF(acc) acc = (role == 5) ? __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, acc, 0, 0, 0)
#define V8() f0 = fmaf(f0, m, c); f1 = fmaf(f1, m, c); f2 = fmaf(f2, m, c); f3 = fmaf(f3, m, c); f4 = fmaf(f4, m, c); f5 = fmaf(f5, m, c); f6 = fmaf(f6, m, c); f7 = fmaf(f7, m, c)
    if (role == 5) {
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); V8();
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); V8();
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); V8();
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0); V8();
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 8, 0); }
      }
    } else {
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a0, 0, 0, 0); V8();
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a1, 0, 0, 0); V8();
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a2, 0, 0, 0); V8();
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a3, 0, 0, 0); V8();
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 8, 0); }
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  } else if (role == 7) {   // VALU role at raised priority
    __builtin_amdgcn_s_setprio(3);
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { V8(); }
    }
    r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  } else if (role == 4) {
    float s = 0.f;
    int idx = lane * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(&lds[(idx + j * 256) & 4095]);
        s += v.x + v.w;
      }
      idx = (idx + 4) & 4095;
    }
    r = s;
  }
  if (r == 123456.789f) out[threadIdx.x] = r;
}

static float run(int rE, int rO, int iters, float* d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<<<256, 512>>>(rE, rO, iters, d);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<<<256, 512>>>(rE, rO, iters, d);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1000.f;
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  const char* nm[] = {"idle", "f32 MFMA 16x16x4", "bf16 MFMA 16x16x32", "VALU fma", "LDS b128 reads", "f32 MFMA+VALU one wave", "bf16 MFMA+VALU one wave", "VALU fma prio 3"};
  const int iters = 20000;
  const int pairs[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 3}, {2, 3}, {1, 4}, {3, 4}, {1, 1}, {3, 3}, {1, 2}, {5, 0}, {6, 0}, {5, 5}, {6, 6}, {1, 7}, {2, 7}};
  for (auto& pr : pairs)
    printf("%-20s + %-20s : %9.1f us\n", nm[pr[0]], nm[pr[1]], run(pr[0], pr[1], iters, d));
  return 0;
}
