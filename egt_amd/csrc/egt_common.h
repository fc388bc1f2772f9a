// Internal helpers shared by the gfx950 kernels and the C-ABI glue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "egt_amd.h"

#define EGT_NEG 1.0e9f  // the reference's additive mask constant (egt_layers.py:92,99,106)
#define EGT_DROPOUT_STREAM 0x5BD1E995u

void egt_set_error(const char* fmt, ...);

#define EGT_FAIL(code, ...)      \
  do {                           \
    egt_set_error(__VA_ARGS__);  \
    return (code);               \
  } while (0)

#define EGT_HIP_LAUNCH_CHECK(name)                                              \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess)                                                      \
      EGT_FAIL(EGT_E_HIP, "%s launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- counter hash (mirrored bit-exactly by oracle/rng_ref.py) ----------------
__host__ __device__ inline uint32_t egt_fmix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ inline uint32_t egt_hash32(uint32_t idx, uint32_t s0, uint32_t s1) {
  return egt_fmix32(egt_fmix32(idx ^ s0) + s1);
}
// 24-bit threshold: sample u = hash>>8 ; "uniform < p"  <=>  u < floor(p*2^24)
__host__ inline uint32_t egt_threshold24(float p) {
  double t = (double)p * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777216.0) t = 16777216.0;
  return (uint32_t)t;
}

// ---- small device math --------------------------------------------------------
__device__ __forceinline__ float egt_sigmoid(float x) {
  // x = -1e9 -> exp(+1e9) = inf -> rcp(inf) = 0 exactly, as in the reference's fp32 path.
  // v_rcp_f32 (1 ulp) instead of the IEEE division sequence: 1 instruction instead of 10
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float wave_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// ---- cross-lane exchange on the VALU (no LDS-crossbar round trip) -------------------
// DPP: value of lane (i ^ MASK) for MASK in {1,2,4,8} inside each 16-lane row.
template <int CTRL>
__device__ __forceinline__ float egt_dpp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));   // bound_ctrl: every lane is written, no 'old' value to materialise
}
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "row-local xor only");
  if (MASK == 1) return egt_dpp<0xB1>(v);                      // quad_perm [1,0,3,2]
  if (MASK == 2) return egt_dpp<0x4E>(v);                      // quad_perm [2,3,0,1]
  if (MASK == 4) return egt_dpp<0x1B>(egt_dpp<0x141>(v));      // row_half_mirror (i^7) o quad reverse (i^3)
  return egt_dpp<0x141>(egt_dpp<0x140>(v));                    // row_mirror (i^15) o row_half_mirror (i^7)
}
// v_permlane16_swap / v_permlane32_swap with both operands = v leave {own, partner} of the
// lanes i and i^16 (resp. i^32) in the two results: any commutative op of the pair is the
// two-lane reduction.
__device__ __forceinline__ float sum_xor16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// all-reduce over the 16 lanes of a row
__device__ __forceinline__ float row_sum16(float v) {
  v += lane_xor<1>(v); v += lane_xor<2>(v); v += lane_xor<4>(v); v += lane_xor<8>(v);
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, lane_xor<1>(v)); v = fmaxf(v, lane_xor<2>(v));
  v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<8>(v));
  return v;
}

// ---- egt_edge.hip, for the fused pair operator (egt_pair.h): edge-parameter gradients from per-workgroup partials ----
void egt_edge_finish_param_grads(int De, const float* gamma, const float* beta, const float* Wg, const float* We,
                                 const float* part_proj, const float* part_upd, int n_partials, float* red_proj, float* red_upd,
                                 float* d_gamma, float* d_beta, float* d_Wg, float* d_bg, float* d_We, float* d_be,
                                 float* d_Wr, float* d_br, hipStream_t st);

// ---- kernel timing hooks (egt_capi.hip) ---------------------------------------
int egt_prof_is_enabled();
void egt_prof_begin(const char* name, hipStream_t s, void** tok);
void egt_prof_end(void* tok, hipStream_t s);
struct EgtProfScope {
  void* tok = nullptr;
  hipStream_t s;
  EgtProfScope(const char* name, hipStream_t st) : s(st) {
    if (egt_prof_is_enabled()) egt_prof_begin(name, st, &tok);
  }
  ~EgtProfScope() { egt_prof_end(tok, s); }
};

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the driver call costs a few microseconds of host time,
// which is what a launch-bound small batch is made of.  (One process normally drives one GPU; the per-device bit keeps a
// process that touches several correct.)
#define EGT_MAX_LDS_ONCE(...)                                                                              \
  do {                                                                                                     \
    static unsigned long long done__ = 0ull;                                                               \
    int dev__ = 0;                                                                                         \
    (void)hipGetDevice(&dev__);                                                                            \
    if (!((done__ >> (dev__ & 63)) & 1ull)) {                                                              \
      (void)hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      done__ |= 1ull << (dev__ & 63);                                                                      \
    }                                                                                                      \
  } while (0)

// EGT_DEBUG_POISON_LDS=1 (debugging aid, read once): every launch is preceded by a kernel that fills the LDS of every CU with NaN
// bit patterns, so a kernel that consumes LDS it never wrote (stale contents of whatever ran before: results that change with
// the launch history) produces NaNs deterministically.  egt_capi.hip.
int egt_debug_poison_enabled();
void egt_debug_poison_lds(hipStream_t s);

#define EGT_LAUNCH(name, kernel, grid, block, lds, stream, ...)            \
  do {                                                                      \
    if (egt_debug_poison_enabled()) egt_debug_poison_lds(stream);           \
    EgtProfScope ps__(name, stream);                                        \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);      \
  } while (0)

// (graph, 16-row group) of a logical workgroup index.  When N is not a multiple of 16 the last row group of every graph is
// short (its workgroup does a fraction of a full group's work): those workgroups are dealt LAST, so they fill the tail of
// the launch instead of sitting between full-size workgroups while the last round runs half empty.
__device__ __forceinline__ void egt_group_order(int wg, int B, int groups, int N, int& b, int& g, int rows = 16) {   // rows per group
  if ((N % rows) == 0) { b = wg / groups; g = wg % groups; return; }
  const int nfull = groups - 1, split = B * nfull;
  if (wg < split) { b = wg / nfull; g = wg % nfull; }
  else { b = wg - split; g = nfull; }
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Remap the hardware
// index so that CONSECUTIVE logical indices (the workgroups of one graph, which share its
// K/V rows) land on the same XCD: logical = (hw % 8) * (n / 8) + hw / 8.
__device__ __forceinline__ int egt_xcd_remap(int hw, int n) {
  return (n & 7) == 0 ? (hw & 7) * (n >> 3) + (hw >> 3) : hw;
}
