// Fused channel FFN of the reference's ffn_block, for gfx950:
//   y = x + Dense_2( act( Dense_1( LayerNorm(x) ) ) )        width W in {16,32,48,64}, hidden 2W  (W = 8: k_ffn8_*, below)
//   lib/models/graph_xformer_model_base.py:230-258 (ffnlr1 / ffnact / ffnlr2, pre-norm, no
//   cross-talk), applied to the edge channels [B,N,N,De] and the node channels [B,N,Dh] (:309-324).
// At De = 64 this is the largest FLOP consumer of the model (24*W^2 flop per row fwd+bwd) and
// a true dense contraction: MFMA-bound (fp32 v_mfma_f32_16x16x4_f32, 256 flop/clk/CU).
//
// One wave owns 16 rows ("tile": 4 KiB of contiguous HBM).  Rows sit on the MFMA COLUMN axis
// and the weights are the A operand, so (as in egt_block.hip) lane (p = lane&15, q = lane>>4)
// keeps row p's channels 16t + 4q + {0..3}: the accumulator layout of the first GEMM IS the B
// operand layout of the second one (contraction order kappa = 16j + 4q + u on both operands) --
// the 128 hidden activations of a row never leave the registers of its four lanes.  The
// LN-folded weights live in LDS as [tile][k-tile][lane] float4 slabs (conflict-free
// ds_read_b128, four contraction steps per read), built once per call by k_ffn_prep.
//   forward : 256 MFMA / tile, 2 waves per SIMD, tiles streamed through ping-pong LDS buffers.
//   backward: recompute + dHid + dXhat (384 MFMA) and the two weight-gradient contractions over
//             the row axis (256 MFMA) in ONE kernel: 1 wave per SIMD so that a wave owns 512
//             registers and keeps all 64 weight-gradient tiles (256 accumulator registers)
//             for its whole row range; per-workgroup partials are reduced deterministically.
#include <stdlib.h>

#include "egt_tile.h"

// Cache-policy hint (egt_tile.h): the FFN kernels read their input tiles (x; x and dy in the backward) once -- non-temporal loads, so
// that what they WRITE (the next launch's input) stays in the memory-side cache.
#ifndef EGT_NT_FFN
#define EGT_NT_FFN true
#endif

// geometry of width W (multiple of 16, <= 64): TW channel tiles, hidden 2W = TH tiles
#define FFN_GEO(W)                                                                          \
  constexpr int FW = (W), FH = 2 * (W), TW = (W) / 16, TH = 2 * TW, SLABF = FW * FH,        \
                TILEF = 16 * (W), FFN_PART = 2 * SLABF + FH + FW;                           \
  (void)TW; (void)TH; (void)TILEF; (void)FFN_PART   /* slab = W*2W floats; partial = T1 | T2 | s1 | s2 */

struct FfnArgs {
  long rows;
  float ln_eps;
  const float *x, *dy;
  float *y, *dx;
  // parameters (Keras layout: kernel [in,out])
  const float *gamma, *beta, *W1, *b1, *W2, *b2;
  // prepared operands (workspace)
  float *slab1, *slab2, *slab3, *slab4, *b1p;
  uint16_t *sA1, *sA2;   // EGT_MM_BF16X3 / EGT_MM_BF16: bf16 (hi | lo) A-operand slabs of the two forward GEMMs
  uint16_t *sA3, *sA4;   // ... and of the backward's dhid = W2 . dy and dxhat = W1p . dpre
  int mm;                // egt_ffn_desc.matmul
  float *x16;            // workspace: the prepared operands of the width-8 kernels (F8_* offsets, k_ffn8_prep)
  float *part, *red;   // backward: per-workgroup partials, reduced sums
  float *g_gamma, *g_beta, *g_W1, *g_b1, *g_W2, *g_b2;
  int nwg;
  int part_len;   // floats per workgroup partial (k_ffn_sum)
  int row8;       // width 8 on the row-per-lane kernels (k_ffn8_*)
  int W;       // channel width (host copy for the run-time-sized helper kernels)
  int guard;   // always 0: opaque phase guards of the backward (see k_ffn_bwd)
};

// slab1[j][t][lane].u = gamma[c] W1[c][16j+pl],            c = 16t+4q+u     (A operand of  pre  = W1p^T . xhat)
// slab2[i][j][lane].u = W2[16j+4q+u][16i+pl]                                (A operand of  y    = W2^T . hid)
// slab3[j][t][lane].u = W2[16j+pl][16t+4q+u]                                (A operand of  dhid = W2 . dy)
// slab4[i][j][lane].u = gamma[16i+pl] W1[16i+pl][16j+4q+u]                  (A operand of  dxhat= W1p . dpre)
// b1p[h] = b1[h] + sum_c beta[c] W1[c][h]
template <int W>
__global__ void __launch_bounds__(256) k_ffn_prep(FfnArgs a) {
  FFN_GEO(W);
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < SLABF) {
    const int u = idx & 3, lane = (idx >> 2) & 63, pl = lane & 15, q = lane >> 4, blk = idx >> 8;
    {
      const int t = blk % TW, j = blk / TW;
      const int c = 16 * t + 4 * q + u;
      a.slab1[idx] = a.gamma[c] * a.W1[c * FH + 16 * j + pl];
      a.slab3[idx] = a.W2[(16 * j + pl) * FW + c];
    }
    {
      const int j = blk % TH, i = blk / TH;
      const int hc = 16 * j + 4 * q + u;
      a.slab2[idx] = a.W2[hc * FW + 16 * i + pl];
      a.slab4[idx] = a.gamma[16 * i + pl] * a.W1[(16 * i + pl) * FH + hc];
    }
  }
  if (idx < FH) {
    float s = a.b1[idx];
    for (int c = 0; c < FW; ++c) s = fmaf(a.beta[c], a.W1[c * FH + idx], s);
    a.b1p[idx] = s;
  }
}

template <int ACT>   // EGT_ACT_RELU (2), EGT_ACT_ELU (3, Keras default alpha = 1)
__device__ __forceinline__ float ffn_act(float v) {
  if (ACT == EGT_ACT_RELU) return fmaxf(v, 0.f);
  return v > 0.f ? v : __expf(v) - 1.0f;
}
template <int ACT>   // derivative from the activation's OUTPUT (hid > 0 <=> pre > 0 for both)
__device__ __forceinline__ float ffn_dact(float hid) {
  if (ACT == EGT_ACT_RELU) return hid > 0.f ? 1.f : 0.f;
  return hid > 0.f ? 1.f : hid + 1.0f;
}

__device__ __forceinline__ void slab_to_lds(float* dst, const float* src, int nfloats, int nthreads) {
  for (int i = threadIdx.x; i < nfloats / 4; i += nthreads)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}

// N consecutive [lane] float4 slabs (1 KiB apart) of one MFMA group in ONE LDS round trip.  hipcc,
// short of registers, emits ds_read_b128 -> s_waitcnt -> 4 MFMA per slab (one exposed LDS latency
// per four MFMAs at one wave per SIMD); the asm issues the group's reads back to back and
// waits once.
template <int N>
__device__ __forceinline__ void slab_read(v4f (&w)[N], const float* slab_lane) {
  const unsigned addr = (unsigned)(size_t)slab_lane;   // LDS byte offset = low 32 bits of the flat address
  static_assert(N >= 1 && N <= 8, "slab group size");
  if constexpr (N == 1)
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w[0]) : "v"(addr));
  else if constexpr (N == 2)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]) : "v"(addr));
  else if constexpr (N == 3)
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]) : "v"(addr));
  else if constexpr (N == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(addr));
  else if constexpr (N == 6)
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                 "ds_read_b128 %3, %6 offset:3072\n\tds_read_b128 %4, %6 offset:4096\n\tds_read_b128 %5, %6 offset:5120\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]) : "v"(addr));
  else if constexpr (N == 8)
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                 "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                 "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(addr));
  else {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float4 t = *reinterpret_cast<const float4*>(slab_lane + n * 256);
      w[n] = (v4f){t.x, t.y, t.z, t.w};
    }
  }
}

// pre-activation tile j of the lane's row: W1p^T . xhat + b1p   (16 MFMA)
// TWO: two alternating accumulators (k_ffn_bwd: one wave per SIMD, nothing else fills the 8 idle cycles between dependent MFMAs;
// the forward's two waves per SIMD cover each other and keep the single chain -- and its bit pattern)
template <int W, bool TWO = false>
__device__ __forceinline__ v4f ffn_gemm1(const float* s1, const float* b1s, const float4 (&x)[W / 16], int j, int lane, int q) {
  constexpr int TW = W / 16;
  const float4 bj = *reinterpret_cast<const float4*>(b1s + 16 * j + 4 * q);
  v4f acc = {bj.x, bj.y, bj.z, bj.w};
  v4f w[TW];
  slab_read<TW>(w, s1 + (j * TW * 64 + lane) * 4);
  if constexpr (TWO) {
    v4f acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      acc = MFMA(w[t][0], x[t].x, acc);
      acc2 = MFMA(w[t][1], x[t].y, acc2);
      acc = MFMA(w[t][2], x[t].z, acc);
      acc2 = MFMA(w[t][3], x[t].w, acc2);
    }
    return acc + acc2;
  } else {
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    acc = MFMA(w[t][0], x[t].x, acc);
    acc = MFMA(w[t][1], x[t].y, acc);
    acc = MFMA(w[t][2], x[t].z, acc);
    acc = MFMA(w[t][3], x[t].w, acc);
  }
  return acc;
  }
}

// ================================================================== forward =====
template <int W, int ACT>
__global__ void __launch_bounds__(512, 2) k_ffn_fwd(FfnArgs a) {
  FFN_GEO(W);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s1 = sm;
  float* s2 = s1 + SLABF;
  float* b1s = s2 + SLABF;          // [128]
  float* b2s = b1s + FH;            // [64]
  float* tiles = b2s + FW;          // [8 waves][2][16 W]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  slab_to_lds(s1, a.slab1, SLABF, 512);
  slab_to_lds(s2, a.slab2, SLABF, 512);
  if (threadIdx.x < FH) b1s[threadIdx.x] = a.b1p[threadIdx.x];
  if (threadIdx.x < FW) b2s[threadIdx.x] = a.b2[threadIdx.x];
  __syncthreads();
  float* tl0 = tiles + wave * 2 * TILEF;
  const long ntiles = (a.rows + 15) / 16;
  const long stride = (long)gridDim.x * 8;
  long tile = (long)blockIdx.x * 8 + wave;
  TileRegs<FW> tr;
  if (tile < ntiles) tile_gload<FW, EGT_NT_FFN>(tr, a.x + tile * TILEF, lane, (int)min(16L, a.rows - tile * 16));
  long prev = -1;
  for (int it = 0; tile < ntiles; tile += stride, ++it) {
    const int rows_valid = (int)min(16L, a.rows - tile * 16);
    float* tl = tl0 + (it & 1) * TILEF;
    lds_sync();
    if (prev >= 0)   // stream out the previous tile's y from the other buffer
      tile_from_lds<FW>(tl0 + ((it - 1) & 1) * TILEF, a.y + prev * TILEF, lane, (int)min(16L, a.rows - prev * 16));
    tile_lds_put<FW>(tl, tr, lane, rows_valid);
    const long nxt = tile + stride;
    if (nxt < ntiles) tile_gload<FW, EGT_NT_FFN>(tr, a.x + nxt * TILEF, lane, (int)min(16L, a.rows - nxt * 16));
    lds_sync();
    float4 x[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) x[t] = frag_read<FW>(tl, p, q, t);
    ln_frags<W>(x, q, a.ln_eps);                            // norm_fnn (gamma/beta folded into the weights)
    v4f h[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) {                                    // fnn_lr1 + activation
      const v4f pre = ffn_gemm1<W>(s1, b1s, x, j, lane, q);
      h[j] = (v4f){ffn_act<ACT>(pre[0]), ffn_act<ACT>(pre[1]), ffn_act<ACT>(pre[2]), ffn_act<ACT>(pre[3])};
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {                                    // fnn_lr2 + res_fnn
      const float4 xr = frag_read<FW>(tl, p, q, i);
      const float4 b = *reinterpret_cast<const float4*>(b2s + 16 * i + 4 * q);
      v4f acc = {xr.x + b.x, xr.y + b.y, xr.z + b.z, xr.w + b.w};
      v4f w[TH];
      slab_read<TH>(w, s2 + (i * TH * 64 + lane) * 4);
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        acc = MFMA(w[j][0], h[j][0], acc);
        acc = MFMA(w[j][1], h[j][1], acc);
        acc = MFMA(w[j][2], h[j][2], acc);
        acc = MFMA(w[j][3], h[j][3], acc);
      }
      frag_write<FW>(tl, p, q, i, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
    prev = tile;
    if (nxt >= ntiles) {   // last tile of this wave: flush
      lds_sync();
      tile_from_lds<FW>(tl, a.y + tile * TILEF, lane, rows_valid);
    }
  }
}

// ================================================= forward on the bf16 matrix pipe =====
// egt_ffn_desc.matmul = EGT_MM_BF16X3: every fp32 operand v is split into two bfloat16 terms,
// v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits survive), and a product a.b is
// evaluated as a_hi.b_hi + a_lo.b_hi + a_hi.b_lo on v_mfma_f32_16x16x32_bf16 with fp32 accumulation
// (the dropped a_lo.b_lo term is 2^-18 relative; every bf16 x bf16 product is exact in fp32).  Per-product
// error <= 2^-16 instead of fp32's 2^-24 -- the outputs stay inside the fp32 parity tolerances of the
// test-suite (tests/test_ffn_gpu.py runs the SAME tolerances on this mode) -- at 3/16 of the fp32 MFMA
// cost: 96 bf16 MFMAs of 16 cycles per 16-row tile instead of 256 fp32 MFMAs of 32 cycles, which turns
// the FFN from MFMA-bound (0.46 of the fp32 matrix peak) into an HBM-bound streaming kernel.
// EGT_MM_BF16 keeps only the hi terms (plain bf16 products, fp32 accumulate; tolerance rtol 2e-2).
// Same tile flow and fragment layout as k_ffn_fwd: lane (p, q) holds row p's channels 16t + 4q + {0..3};
// one 16x16x32 step contracts the 8 values a lane holds of channel tiles 2s and 2s+1 (slot i <-> tile
// 2s + (i >> 2), offset i & 3) -- both operands use that order, so the accumulators of GEMM 1 are again
// the B operand of GEMM 2 without leaving the lane.
// sA1[j][s][part][lane][i] = bf16_part( gamma[c] W1[c][16j + pl] ),  c = 16 (2s + (i >> 2)) + 4q + (i & 3)   (0 past W)
// sA2[o][s][part][lane][i] = bf16_part( W2[hc][16o + pl] ),         hc = 16 (2s + (i >> 2)) + 4q + (i & 3)
// part 0 = hi, 1 = lo; NS1 = ceil(TW / 2) steps for GEMM 1, TW steps for GEMM 2 (hidden = 2W = TW x 32)
template <int W>
__global__ void __launch_bounds__(256) k_ffn_prep_bf(FfnArgs a) {
  FFN_GEO(W);
  constexpr int NS1 = (TW + 1) / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // one thread per (block, lane, i)
  auto parts = [](float v, uint16_t& hi, uint16_t& lo) {
    const uint32_t h = pk_bf16(v, 0.f) & 0xFFFFu;
    hi = (uint16_t)h;
    lo = (uint16_t)(pk_bf16(v - __uint_as_float(h << 16), 0.f) & 0xFFFFu);
  };
  if (idx < TH * NS1 * 512) {
    const int i = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9, s = blk % NS1, j = blk / NS1;
    const int pl = lane & 15, q = lane >> 4, t = 2 * s + (i >> 2);
    const int c = 16 * t + 4 * q + (i & 3);
    const float v = t < TW ? a.gamma[c] * a.W1[c * FH + 16 * j + pl] : 0.f;
    uint16_t hi, lo;
    parts(v, hi, lo);
    a.sA1[((size_t)(blk * 2 + 0) * 64 + lane) * 8 + i] = hi;
    a.sA1[((size_t)(blk * 2 + 1) * 64 + lane) * 8 + i] = lo;
  }
  if (idx < TW * TW * 512) {
    const int i = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9, s = blk % TW, o = blk / TW;
    const int pl = lane & 15, q = lane >> 4;
    const int hc = 16 * (2 * s + (i >> 2)) + 4 * q + (i & 3);
    uint16_t hi, lo;
    parts(a.W2[hc * FW + 16 * o + pl], hi, lo);
    a.sA2[((size_t)(blk * 2 + 0) * 64 + lane) * 8 + i] = hi;
    a.sA2[((size_t)(blk * 2 + 1) * 64 + lane) * 8 + i] = lo;
    // sA4[o][s][part][lane][i] = gamma[16o + pl] W1[16o + pl][hc]      (A operand of dxhat = W1p . dpre)
    parts(a.gamma[16 * o + pl] * a.W1[(16 * o + pl) * FH + hc], hi, lo);
    a.sA4[((size_t)(blk * 2 + 0) * 64 + lane) * 8 + i] = hi;
    a.sA4[((size_t)(blk * 2 + 1) * 64 + lane) * 8 + i] = lo;
  }
  if (idx < TH * NS1 * 512) {   // sA3[j][s][part][lane][i] = W2[16j + pl][c]   (A operand of dhid = W2 . dy)
    const int i = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9, s = blk % NS1, j = blk / NS1;
    const int pl = lane & 15, q = lane >> 4, t = 2 * s + (i >> 2);
    const int c = 16 * t + 4 * q + (i & 3);
    uint16_t hi, lo;
    parts(t < TW ? a.W2[(16 * j + pl) * FW + c] : 0.f, hi, lo);
    a.sA3[((size_t)(blk * 2 + 0) * 64 + lane) * 8 + i] = hi;
    a.sA3[((size_t)(blk * 2 + 1) * 64 + lane) * 8 + i] = lo;
  }
  if (idx < FH) {
    float s = a.b1[idx];
    for (int c = 0; c < FW; ++c) s = fmaf(a.beta[c], a.W1[c * FH + idx], s);
    a.b1p[idx] = s;
  }
}

template <int W, int ACT, bool SPLIT>
__global__ void __launch_bounds__(512, 2) k_ffn_fwd_bf(FfnArgs a) {
  FFN_GEO(W);
  constexpr int NS1 = (TW + 1) / 2;
  constexpr int A1F = TH * NS1 * 2 * 256, A2F = TW * TW * 2 * 256;   // slab sizes in floats (1 KiB = 256 floats per [part][lane] block)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s1 = sm;
  float* s2 = s1 + A1F;
  float* b1s = s2 + A2F;            // [2W]
  float* b2s = b1s + FH;            // [W]
  float* tiles = b2s + FW;          // [8 waves][2][16 W]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  slab_to_lds(s1, reinterpret_cast<const float*>(a.sA1), A1F, 512);
  slab_to_lds(s2, reinterpret_cast<const float*>(a.sA2), A2F, 512);
  if (threadIdx.x < FH) b1s[threadIdx.x] = a.b1p[threadIdx.x];
  if (threadIdx.x < FW) b2s[threadIdx.x] = a.b2[threadIdx.x];
  __syncthreads();
  float* tl0 = tiles + wave * 2 * TILEF;
  const long ntiles = (a.rows + 15) / 16;
  const long stride = (long)gridDim.x * 8;
  long tile = (long)blockIdx.x * 8 + wave;
  TileRegs<FW> tr;
  if (tile < ntiles) tile_gload<FW, EGT_NT_FFN>(tr, a.x + tile * TILEF, lane, (int)min(16L, a.rows - tile * 16));
  long prev = -1;
  for (int it = 0; tile < ntiles; tile += stride, ++it) {
    const int rows_valid = (int)min(16L, a.rows - tile * 16);
    float* tl = tl0 + (it & 1) * TILEF;
    lds_sync();
    if (prev >= 0)   // stream out the previous tile's y from the other buffer
      tile_from_lds<FW>(tl0 + ((it - 1) & 1) * TILEF, a.y + prev * TILEF, lane, (int)min(16L, a.rows - prev * 16));
    tile_lds_put<FW>(tl, tr, lane, rows_valid);
    const long nxt = tile + stride;
    if (nxt < ntiles) tile_gload<FW, EGT_NT_FFN>(tr, a.x + nxt * TILEF, lane, (int)min(16L, a.rows - nxt * 16));
    lds_sync();
    float4 x[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) x[t] = frag_read<FW>(tl, p, q, t);
    ln_frags<W>(x, q, a.ln_eps);                            // norm_fnn (gamma/beta folded into the weights)
    Bf8 xh[NS1], xl[NS1];
#pragma unroll
    for (int s = 0; s < NS1; ++s) {
      const float4 x0 = x[2 * s];
      const float4 x1 = (2 * s + 1 < TW) ? x[(2 * s + 1 < TW) ? 2 * s + 1 : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      split8<SPLIT>(v, xh[s], xl[s]);
    }
    v4f h[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) {                                    // fnn_lr1 + activation
      const float4 bj = *reinterpret_cast<const float4*>(b1s + 16 * j + 4 * q);
      v4f acc = {bj.x, bj.y, bj.z, bj.w};
#pragma unroll
      for (int s = 0; s < NS1; ++s) {
        Bf8 ah, al;
        ah.q = *reinterpret_cast<const uint4*>(s1 + ((size_t)((j * NS1 + s) * 2 + 0) * 64 + lane) * 4);
        acc = MFMA_BF(ah.v, xh[s].v, acc);
        if (SPLIT) {
          al.q = *reinterpret_cast<const uint4*>(s1 + ((size_t)((j * NS1 + s) * 2 + 1) * 64 + lane) * 4);
          acc = MFMA_BF(al.v, xh[s].v, acc);
          acc = MFMA_BF(ah.v, xl[s].v, acc);
        }
      }
      h[j] = (v4f){ffn_act<ACT>(acc[0]), ffn_act<ACT>(acc[1]), ffn_act<ACT>(acc[2]), ffn_act<ACT>(acc[3])};
    }
    Bf8 hh[TW], hl[TW];
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      const float v[8] = {h[2 * s][0], h[2 * s][1], h[2 * s][2], h[2 * s][3], h[2 * s + 1][0], h[2 * s + 1][1], h[2 * s + 1][2], h[2 * s + 1][3]};
      split8<SPLIT>(v, hh[s], hl[s]);
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) {                                    // fnn_lr2 + res_fnn
      const float4 xr = frag_read<FW>(tl, p, q, i);
      const float4 b = *reinterpret_cast<const float4*>(b2s + 16 * i + 4 * q);
      v4f acc = {xr.x + b.x, xr.y + b.y, xr.z + b.z, xr.w + b.w};
#pragma unroll
      for (int s = 0; s < TW; ++s) {
        Bf8 ah, al;
        ah.q = *reinterpret_cast<const uint4*>(s2 + ((size_t)((i * TW + s) * 2 + 0) * 64 + lane) * 4);
        acc = MFMA_BF(ah.v, hh[s].v, acc);
        if (SPLIT) {
          al.q = *reinterpret_cast<const uint4*>(s2 + ((size_t)((i * TW + s) * 2 + 1) * 64 + lane) * 4);
          acc = MFMA_BF(al.v, hh[s].v, acc);
          acc = MFMA_BF(ah.v, hl[s].v, acc);
        }
      }
      frag_write<FW>(tl, p, q, i, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
    prev = tile;
    if (nxt >= ntiles) {   // last tile of this wave: flush
      lds_sync();
      tile_from_lds<FW>(tl, a.y + tile * TILEF, lane, rows_valid);
    }
  }
}

// ================================================================= backward =====
// MM: egt_ffn_desc.matmul.  EGT_MM_BF16X3 / EGT_MM_BF16 move the three channel contractions (recompute,
// dhid = W2 . dy, dxhat = W1p . dpre: 384 of the 640 fp32 MFMAs of a tile) to the bf16 matrix pipe (144 / 48
// MFMAs of 16 cycles); the weight-gradient contractions over the ROW axis stay exact fp32 (their operands are
// read transposed out of the fp32 LDS tiles).
template <int W, int ACT, int MM>
__global__ void __launch_bounds__(256, 1) k_ffn_bwd(FfnArgs a) {
  FFN_GEO(W);
  constexpr int NS1 = (TW + 1) / 2;
  constexpr bool SPLIT = MM == EGT_MM_BF16X3;
  constexpr int S1F = MM ? TH * NS1 * 2 * 256 : SLABF, S4F = MM ? TW * TW * 2 * 256 : SLABF;   // slab sizes in floats
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s1 = sm;
  float* s3 = s1 + S1F;
  float* s4 = s3 + S1F;
  float* b1s = s4 + S4F;            // [128]
  float* tiles = b1s + FH;          // [4 waves][x | dy | hid/dpre half][TILEF]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p0 = lane & 15, q0 = lane >> 4;
  if constexpr (MM != 0) {
    slab_to_lds(s1, reinterpret_cast<const float*>(a.sA1), S1F, 256);
    slab_to_lds(s3, reinterpret_cast<const float*>(a.sA3), S1F, 256);
    slab_to_lds(s4, reinterpret_cast<const float*>(a.sA4), S4F, 256);
  } else {
    slab_to_lds(s1, a.slab1, SLABF, 256);
    slab_to_lds(s3, a.slab3, SLABF, 256);
    slab_to_lds(s4, a.slab4, SLABF, 256);
  }
  if (threadIdx.x < FH) b1s[threadIdx.x] = a.b1p[threadIdx.x];
  __syncthreads();
  float* et = tiles + wave * 3 * TILEF;
  float* dt = et + TILEF;
  float* hd = dt + TILEF;
  v4f accT1[TW * TH], accT2[TH * TW];   // T1[in 16t+4q+r][hid 16j+pl] = accT1[t*8+j][r] ; T2[hid 16j+4q+r][out 16i+pl] = accT2[j*4+i][r]
#pragma unroll
  for (int k = 0; k < TW * TH; ++k) { accT1[k] = (v4f){0.f, 0.f, 0.f, 0.f}; accT2[k] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  // bias-gradient sums, taken from the ROW-axis operands of the weight-gradient products (the lane already holds column 16j + pl of
  // rows q + 4s there): one register per column tile instead of four per fragment -- 12 registers instead of 48 in a kernel whose
  // 256 weight-gradient accumulators leave no slack (the W = 64 fp32 instance spilled 20-40 registers before)
  float sp[TH];                // sums of dpre over rows q + 4s, column hid 16j + pl
  float sd[TW];                // sums of dy   over rows q + 4s, column out 16i + pl
#pragma unroll
  for (int j = 0; j < TH; ++j) sp[j] = 0.f;
#pragma unroll
  for (int t = 0; t < TW; ++t) sd[t] = 0.f;

  const long ntiles = (a.rows + 15) / 16;
  const long stride = (long)gridDim.x * 4;
  long tile = (long)blockIdx.x * 4 + wave;
  TileRegs<FW> te, td;
  if (tile < ntiles) {
    const int rv = (int)min(16L, a.rows - tile * 16);
    tile_gload<FW, EGT_NT_FFN>(te, a.x + tile * TILEF, lane, rv);
    tile_gload<FW, EGT_NT_FFN>(td, a.dy + tile * TILEF, lane, rv);
  }
  long prev = -1;
  for (; tile < ntiles; tile += stride) {
    const int rows_valid = (int)min(16L, a.rows - tile * 16);
    lds_sync();
    if (prev >= 0) tile_from_lds<FW>(dt, a.dx + prev * TILEF, lane, (int)min(16L, a.rows - prev * 16));   // dx of the previous tile
    lds_sync();
    tile_lds_put<FW>(et, te, lane, rows_valid);     // rows past the end are zero: they add nothing to the sums
    tile_lds_put<FW>(dt, td, lane, rows_valid);
    const long nxt = tile + stride;
    if (nxt < ntiles) {
      const int rv = (int)min(16L, a.rows - nxt * 16);
      tile_gload<FW, EGT_NT_FFN>(te, a.x + nxt * TILEF, lane, rv);
      tile_gload<FW, EGT_NT_FFN>(td, a.dy + nxt * TILEF, lane, rv);
    }
    lds_sync();
    // The phases below sit behind opaque always-true guards (a.guard == 0): the uniform branches
    // split the tile body into basic blocks so that hipcc keeps every phase's operands local
    // (re-read from the LDS tiles) instead of stretching 200+ live registers across the tile.
    // ---- recompute: xhat, hid ----
    float rstd;
    v4f h[TH], dp[TH];
    {
      int p = p0, q = q0;
      asm volatile("" : "+v"(p), "+v"(q));   // keep this phase's LDS address math inside the phase (no hoisting out of the tile loop)
      float4 x[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) x[t] = frag_read<FW>(et, p, q, t);
      rstd = ln_frags<W>(x, q, a.ln_eps);
#pragma unroll
      for (int t = 0; t < TW; ++t) frag_write<FW>(et, p, q, t, x[t]);   // xhat: A operand of T1 (other rows) + LN backward
      if constexpr (MM != 0) {
        v4f xv[TW];
        Bf8 xh[NS1], xl[NS1];
#pragma unroll
        for (int t = 0; t < TW; ++t) xv[t] = (v4f){x[t].x, x[t].y, x[t].z, x[t].w};
        split_tiles<TW, SPLIT>(xv, xh, xl);
#pragma unroll
        for (int j = 0; j < TH; ++j) {
          const float4 bj = *reinterpret_cast<const float4*>(b1s + 16 * j + 4 * q);
          const v4f pre = bf_gemm<NS1, SPLIT>(s1, j * NS1, lane, xh, xl, (v4f){bj.x, bj.y, bj.z, bj.w});
          h[j] = (v4f){ffn_act<ACT>(pre[0]), ffn_act<ACT>(pre[1]), ffn_act<ACT>(pre[2]), ffn_act<ACT>(pre[3])};
        }
      } else {
#pragma unroll
        for (int j = 0; j < TH; ++j) {
          const v4f pre = ffn_gemm1<W, (W < 64)>(s1, b1s, x, j, lane, q);   // (W = 64: the extra accumulator would spill)
          h[j] = (v4f){ffn_act<ACT>(pre[0]), ffn_act<ACT>(pre[1]), ffn_act<ACT>(pre[2]), ffn_act<ACT>(pre[3])};
        }
      }
    }
    lds_sync();
    // ---- dhid = W2 . dy ; dpre = dhid * act'(pre) ----
    if (a.guard == 0) {
      int p = p0, q = q0;
      asm volatile("" : "+v"(p), "+v"(q));   // keep this phase's LDS address math inside the phase (no hoisting out of the tile loop)
      float4 dyf[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        dyf[t] = frag_read<FW>(dt, p, q, t);
      }
      Bf8 dyh[NS1], dyl[NS1];
      if constexpr (MM != 0) {
        v4f dv[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) dv[t] = (v4f){dyf[t].x, dyf[t].y, dyf[t].z, dyf[t].w};
        split_tiles<TW, SPLIT>(dv, dyh, dyl);
      }
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MM != 0) {
          acc = bf_gemm<NS1, SPLIT>(s3, j * NS1, lane, dyh, dyl, acc);
        } else {
          // two alternating accumulators: a chain of MFMAs on ONE accumulator issues every 40 cycles (dependent latency) instead of
          // every 32, and with one wave per SIMD nothing else fills the gap
          v4f w[TW], acc2 = {0.f, 0.f, 0.f, 0.f};
          slab_read<TW>(w, s3 + (j * TW * 64 + lane) * 4);
#pragma unroll
          for (int t = 0; t < TW; ++t) {
            acc = MFMA(w[t][0], dyf[t].x, acc);
            acc2 = MFMA(w[t][1], dyf[t].y, acc2);
            acc = MFMA(w[t][2], dyf[t].z, acc);
            acc2 = MFMA(w[t][3], dyf[t].w, acc2);
          }
          acc += acc2;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= ffn_dact<ACT>(h[j][r]);
        dp[j] = acc;
      }
    }
    // ---- weight gradients: contractions over the 16 rows of the tile (row index rho = q + 4s) ----
    if (a.guard == 0) {                             // T2 += hid^T . dy
      int p = p0, q = q0;
      asm volatile("" : "+v"(p), "+v"(q));   // keep this phase's LDS address math inside the phase (no hoisting out of the tile loop)
      float bdy[TW][4];   // dy[rho][16i+pl]
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) bdy[i][s] = elem_read<FW>(dt, q + 4 * s, 16 * i + p);
#pragma unroll
      for (int i = 0; i < TW; ++i) sd[i] += (bdy[i][0] + bdy[i][1]) + (bdy[i][2] + bdy[i][3]);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        lds_sync();
#pragma unroll
        for (int jj = 0; jj < TW; ++jj)
          frag_write<FW>(hd, p, q, jj, make_float4(h[TW * half + jj][0], h[TW * half + jj][1], h[TW * half + jj][2], h[TW * half + jj][3]));
        lds_sync();
#pragma unroll
        for (int jj = 0; jj < TW; ++jj)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float ah = elem_read<FW>(hd, q + 4 * s, 16 * jj + p);
#pragma unroll
            for (int i = 0; i < TW; ++i) accT2[(TW * half + jj) * TW + i] = MFMA(ah, bdy[i][s], accT2[(TW * half + jj) * TW + i]);
          }
      }
    }
    if (a.guard == 0) {                             // T1 += xhat^T . dpre
      int p = p0, q = q0;
      asm volatile("" : "+v"(p), "+v"(q));   // keep this phase's LDS address math inside the phase (no hoisting out of the tile loop)
      float axh[TW][4];   // xhat[rho][16t+pl]
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) axh[t][s] = elem_read<FW>(et, q + 4 * s, 16 * t + p);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        lds_sync();
#pragma unroll
        for (int jj = 0; jj < TW; ++jj)
          frag_write<FW>(hd, p, q, jj, make_float4(dp[TW * half + jj][0], dp[TW * half + jj][1], dp[TW * half + jj][2], dp[TW * half + jj][3]));
        lds_sync();
#pragma unroll
        for (int jj = 0; jj < TW; ++jj)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float bd = elem_read<FW>(hd, q + 4 * s, 16 * jj + p);
            sp[TW * half + jj] += bd;
#pragma unroll
            for (int t = 0; t < TW; ++t) accT1[t * TH + TW * half + jj] = MFMA(axh[t][s], bd, accT1[t * TH + TW * half + jj]);
          }
      }
    }
    // ---- dxhat = W1p . dpre ; LayerNorm backward ; dx = dy + ... (in place over the dy tile) ----
    if (a.guard == 0) {
      int p = p0, q = q0;
      asm volatile("" : "+v"(p), "+v"(q));   // keep this phase's LDS address math inside the phase (no hoisting out of the tile loop)
      float4 dxh[TW], x[TW];
      float m1 = 0.f, m2 = 0.f;
      Bf8 dph[TW], dpl[TW];
      if constexpr (MM != 0) split_tiles<TH, SPLIT>(dp, dph, dpl);
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        x[i] = frag_read<FW>(et, p, q, i);          // xhat
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MM != 0) {
          acc = bf_gemm<TW, SPLIT>(s4, i * TW, lane, dph, dpl, acc);
        } else {
          v4f w[TH], acc2 = {0.f, 0.f, 0.f, 0.f};   // (two alternating accumulators, as in the dhid product)
          slab_read<TH>(w, s4 + (i * TH * 64 + lane) * 4);
#pragma unroll
          for (int j = 0; j < TH; ++j) {
            acc = MFMA(w[j][0], dp[j][0], acc);
            acc2 = MFMA(w[j][1], dp[j][1], acc2);
            acc = MFMA(w[j][2], dp[j][2], acc);
            acc2 = MFMA(w[j][3], dp[j][3], acc2);
          }
          acc += acc2;
        }
        dxh[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        m1 += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        m2 = fmaf(acc[0], x[i].x, m2); m2 = fmaf(acc[1], x[i].y, m2);
        m2 = fmaf(acc[2], x[i].z, m2); m2 = fmaf(acc[3], x[i].w, m2);
      }
      m1 = sum_over_q(m1) * (1.0f / FW); m2 = sum_over_q(m2) * (1.0f / FW);
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const float4 dyv = frag_read<FW>(dt, p, q, i);
        float4 o;
        o.x = dyv.x + rstd * (dxh[i].x - m1 - x[i].x * m2);
        o.y = dyv.y + rstd * (dxh[i].y - m1 - x[i].y * m2);
        o.z = dyv.z + rstd * (dxh[i].z - m1 - x[i].z * m2);
        o.w = dyv.w + rstd * (dxh[i].w - m1 - x[i].w * m2);
        frag_write<FW>(dt, p, q, i, o);
      }
    }
    prev = tile;
  }
  lds_sync();
  if (prev >= 0) tile_from_lds<FW>(dt, a.dx + prev * TILEF, lane, (int)min(16L, a.rows - prev * 16));

  // ---- per-workgroup partial: (wave 0 + wave 2) + (wave 1 + wave 3), a pairwise tree through two LANE-LINEAR LDS images
  //      (accumulator tile k of lane l at float4 slot k*64 + l: conflict-free b128 moves, no read-modify-write).  The
  //      canonical [row][col] layout is only formed by the final gather, which all four waves share.  (The first version
  //      added the waves one after the other into a canonical image: 4-way bank conflicts on ~520 b32 read-modify-writes
  //      per wave, 7 us per wave = 28 of the 73 us the kernel takes on the node channels.) ----
  const int p = p0, q = q0;
  __syncthreads();
  constexpr int NT = TW * TH;
  float* img0 = sm;
  float* img1 = img0 + 2 * NT * 256;
  float* ssum = img1 + 2 * NT * 256;   // [4 waves][FH + FW]: bias-gradient sums in canonical order
  auto img_put = [&](float* img) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      *reinterpret_cast<v4f*>(img + (k * 64 + lane) * 4) = accT1[k];
      *reinterpret_cast<v4f*>(img + ((NT + k) * 64 + lane) * 4) = accT2[k];
    }
  };
  auto img_add = [&](const float* img) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      accT1[k] += *reinterpret_cast<const v4f*>(img + (k * 64 + lane) * 4);
      accT2[k] += *reinterpret_cast<const v4f*>(img + ((NT + k) * 64 + lane) * 4);
    }
  };
  {
    float* sw = ssum + wave * (FH + FW);
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const float v = sum_over_q(sp[j]);       // the four lane rows hold rows q, q + 4, q + 8, q + 12 of every tile
      if (q == 0) sw[16 * j + p] = v;
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const float v = sum_over_q(sd[t]);
      if (q == 0) sw[FH + 16 * t + p] = v;
    }
  }
  if (wave == 2) img_put(img0);
  if (wave == 3) img_put(img1);
  __syncthreads();
  if (wave == 0) img_add(img0);
  if (wave == 1) { img_add(img1); img_put(img1); }   // (a lane rewrites the slots it has just read: in order within the wave)
  __syncthreads();
  if (wave == 0) { img_add(img1); img_put(img0); }
  __syncthreads();
  float* out = a.part + (size_t)blockIdx.x * FFN_PART;
  for (int i = threadIdx.x; i < FFN_PART; i += 256) {
    float v;
    if (i < SLABF) {                 // T1[row = 16t + 4q + r][col = 16j + p]
      const int row = i / FH, col = i % FH;
      v = img0[(((row >> 4) * TH + (col >> 4)) * 64 + 4 * (row & 12) + (col & 15)) * 4 + (row & 3)];
    } else if (i < 2 * SLABF) {      // T2[row = 16j + 4q + r][col = 16i + p]
      const int e = i - SLABF, row = e / FW, col = e % FW;
      v = img0[((NT + (row >> 4) * TW + (col >> 4)) * 64 + 4 * (row & 12) + (col & 15)) * 4 + (row & 3)];
    } else {
      const int e = i - 2 * SLABF;
      v = ((ssum[e] + ssum[(FH + FW) + e]) + ssum[2 * (FH + FW) + e]) + ssum[3 * (FH + FW) + e];
    }
    out[i] = v;
  }
}

// T1 = sum xhat^T.dpre, s1 = sum dpre, T2 = sum hid^T.dy, s2 = sum dy (a.red)  ->  parameter gradients
//   dW1[c][h] = gamma_c T1[c][h] + beta_c s1[h] ; dgamma_c = sum_h W1[c][h] T1[c][h] ; dbeta_c = sum_h W1[c][h] s1[h]
//   db1 = s1 ; dW2 = T2 ; db2 = s2          (width 8 keeps T2 transposed: T2^T[o][h])
// One workgroup of 1024 threads, 16 lanes per channel for the gamma / beta contractions.  (Folding this into k_ffn_sum behind
// a last-workgroup ticket was tried: the device-scope fences cost more than the launch, 20 us against 7 + 10 as two launches.)
__global__ void __launch_bounds__(1024) k_ffn_param_grads(FfnArgs a) {
  const int FW = a.W, FH = 2 * a.W, SLABF = FW * FH, t = threadIdx.x;
  const float* T1 = a.red;
  const float* T2 = a.red + SLABF;
  const float* s1 = a.red + 2 * SLABF;
  const float* s2 = s1 + FH;
  // every load before the first store (the gradient pointers may alias anything as far as the compiler knows: a store
  // inside the loops would make each iteration its own memory round trip)
  float w1v[8], w2v[8];   // SLABF <= 64 * 128 = 8 x 1024
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = t + 1024 * u;
    if (i < SLABF) {
      w1v[u] = fmaf(a.gamma[i / FH], T1[i], a.beta[i / FH] * s1[i % FH]);
      w2v[u] = a.row8 ? T2[(i % FW) * FH + i / FW] : T2[i];
    }
  }
  const float b1v = t < FH ? s1[t] : 0.f, b2v = t < FW ? s2[t] : 0.f;
  float dg = 0.f, db = 0.f;
  const int c = t >> 4, part = t & 15;   // 16 lanes per channel
  if (t < FW * 16) {
    for (int h = part; h < FH; h += 16) {
      const float w = a.W1[c * FH + h];
      dg = fmaf(w, T1[c * FH + h], dg);
      db = fmaf(w, s1[h], db);
    }
    dg = row_sum16(dg); db = row_sum16(db);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = t + 1024 * u;
    if (i < SLABF) { a.g_W1[i] = w1v[u]; a.g_W2[i] = w2v[u]; }
  }
  if (t < FH) a.g_b1[t] = b1v;
  if (t < FW) a.g_b2[t] = b2v;
  if (t < FW * 16 && part == 0) { a.g_gamma[c] = dg; a.g_beta[c] = db; }
}

// deterministic sum over the workgroup partials: 64 outputs per workgroup, partial axis over 16 waves with 8 loads in
// flight each (the loop is a chain of HBM round trips: 1024 partials are 8 trips per thread)
__global__ void __launch_bounds__(1024) k_ffn_sum(FfnArgs a) {
  const int FFN_PART = a.part_len;
  __shared__ float red[16][64];
  const int l = threadIdx.x & 63, o = blockIdx.x * 64 + l, pg = threadIdx.x >> 6;
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = 0.f;
  if (o < FFN_PART) {
    int pi = pg;
    for (; pi + 112 < a.nwg; pi += 128)
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += a.part[(size_t)(pi + 16 * u) * FFN_PART + o];
    for (; pi < a.nwg; pi += 16) v[0] += a.part[(size_t)pi * FFN_PART + o];
  }
  red[pg][l] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  __syncthreads();
  if (pg == 0 && o < FFN_PART) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][l];
    a.red[o] = s;
  }
}

// ============================================== width 8: one row per lane =====
// De = 8 (configs 3/4: the CIFAR10 / PATTERN edge channels).  A 16x16 MFMA tile is the wrong shape for an 8 -> 16 -> 8
// FFN (two 8-wide rows riding in one 16-wide row behind block-diagonal weights: half of every MFMA multiplies zeros, and the
// tile round trips through LDS made the W = 16 backward VALU/latency-bound at 0.18 of the HBM roof).  Here a lane owns one
// whole row: LayerNorm and the three channel contractions (128 FMAs each) are in-lane VALU work on v_pk_fma_f32 with the
// weights as wave-uniform SGPR pairs (s_load from the constant address space, re-fetched per chunk), x/dy/dx move as two
// 16-byte accesses per lane (a wave covers 2 KiB of contiguous HBM), and only the two weight-gradient contractions over the
// ROW axis go to the matrix pipe: the wave transposes [xhat | dy] and dpre / hid through a padded LDS image ([16][68] floats:
// conflict-free b32 column writes and b128 row reads) and contracts its 64 rows with 16 + 16 v_mfma_f32_16x16x4_f32.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* cfp;   // constant address space: uniform reads become s_load
#define F8_W1P 0      // gamma[c] W1[c][h]            [8][16]   (pairs over h: pre)
#define F8_W1PT 128   // gamma[c] W1[c][h] as [h][c]  [16][8]   (pairs over c: dxhat)
#define F8_W2T 256    // W2[h][o] as [o][h]           [8][16]   (pairs over h: dhid)
#define F8_W2 384     // W2[h][o]                     [16][8]   (pairs over o: y)
#define F8_B1P 512    // b1[h] + sum_c beta[c] W1[c][h]
#define F8_B2 528
#define F8_PREP_FLOATS 536
#define F8_PART 280   // per-workgroup partial: T1[c][h] 128 | T2^T[o][h] 128 | s1[h] 16 | s2[o] 8
#define F8_LD 68      // row pitch of the transposed LDS images (floats)

__global__ void __launch_bounds__(128) k_ffn8_prep(FfnArgs a) {
  const int t = threadIdx.x;
  float* w = a.x16;
  const int c = t >> 4, h = t & 15;
  const float v = a.gamma[c] * a.W1[c * 16 + h];
  w[F8_W1P + c * 16 + h] = v;
  w[F8_W1PT + h * 8 + c] = v;
  const float u = a.W2[h * 8 + c];   // (h, o = c)
  w[F8_W2 + h * 8 + c] = u;
  w[F8_W2T + c * 16 + h] = u;
  if (t < 16) {
    float s = a.b1[t];
    for (int k = 0; k < 8; ++k) s = fmaf(a.beta[k], a.W1[k * 16 + t], s);
    w[F8_B1P + t] = s;
  }
  if (t < 8) w[F8_B2 + t] = a.b2[t];
}

__device__ __forceinline__ v2f splat2(float v) { return (v2f){v, v}; }
__device__ __forceinline__ v2f ldw2(cfp w, int i) { return (v2f){w[i], w[i + 1]}; }

// the lane's row: two-pass moments, x -> xhat in place, returns rstd
__device__ __forceinline__ float ffn8_ln(float (&x)[8], float eps) {
  const float mu = (((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * 0.125f;
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) { x[c] -= mu; v = fmaf(x[c], x[c], v); }
  const float rstd = rsqrtf(fmaf(v, 0.125f, eps));
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] *= rstd;
  return rstd;
}
// hid = act(W1p^T . xhat + b1p)
template <int ACT>
__device__ __forceinline__ void ffn8_hidden(cfp w, const float (&xh)[8], float (&hid)[16]) {
  v2f pre[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) pre[k] = ldw2(w, F8_B1P + 2 * k);
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) pre[k] = ldw2(w, F8_W1P + c * 16 + 2 * k) * splat2(xh[c]) + pre[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) { hid[2 * k] = ffn_act<ACT>(pre[k][0]); hid[2 * k + 1] = ffn_act<ACT>(pre[k][1]); }
}
__device__ __forceinline__ void ffn8_load(const float* src, long row, bool ok, float (&v)[8]) {
  float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
  if (ok) {
    lo = *reinterpret_cast<const float4*>(src + row * 8);
    hi = *reinterpret_cast<const float4*>(src + row * 8 + 4);
  }
  v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

template <int ACT>
__global__ void __launch_bounds__(256, 4) k_ffn8_fwd(FfnArgs a) {
  const int lane = threadIdx.x & 63;
  const long nchunk = (a.rows + 63) / 64, stride = (long)gridDim.x * 4;
  for (long ch = (long)blockIdx.x * 4 + (threadIdx.x >> 6); ch < nchunk; ch += stride) {
    const long row = ch * 64 + lane;
    const bool ok = row < a.rows;
    float x[8], xh[8], hid[16];
    ffn8_load(a.x, row, ok, x);
    cfp w = (cfp)a.x16;
    asm volatile("" : "+s"(w));   // per-chunk copy of the base: the weight fetches stay inside the loop (272 SGPRs do not exist)
#pragma unroll
    for (int c = 0; c < 8; ++c) xh[c] = x[c];
    ffn8_ln(xh, a.ln_eps);
    ffn8_hidden<ACT>(w, xh, hid);
    v2f y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = (v2f){x[2 * k], x[2 * k + 1]} + ldw2(w, F8_B2 + 2 * k);
#pragma unroll
    for (int h = 0; h < 16; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = ldw2(w, F8_W2 + h * 8 + 2 * k) * splat2(hid[h]) + y[k];
    if (ok) {
      *reinterpret_cast<float4*>(a.y + row * 8) = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]);
      *reinterpret_cast<float4*>(a.y + row * 8 + 4) = make_float4(y[2][0], y[2][1], y[3][0], y[3][1]);
    }
  }
}

#ifndef F8_BWD_OCC
#define F8_BWD_OCC 3   // waves per SIMD: 4 (128 registers) spills ~20 values per chunk
#endif
template <int ACT>
__global__ void __launch_bounds__(256, F8_BWD_OCC) k_ffn8_bwd(FfnArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[4 * 2 * 16 * F8_LD];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kq = lane >> 4;
  float* tA = sm + wave * 2 * 16 * F8_LD;   // [xhat 0-7 | dy 8-15][row]
  float* tB = tA + 16 * F8_LD;              // dpre, then hid: [h][row]
  v4f accT1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // [xhat | dy]^T . dpre : rows 0-7 = T1[c][h]
  v4f accT2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // [xhat | dy]^T . hid  : rows 8-15 = T2^T[o][h]
  // bias gradients from the transposed images too: lane (m, kq) adds up the 16 rows it reads of dpre[h = m] and of
  // [xhat | dy][m] (one register each instead of 24 per-row accumulators)
  v4f sp4 = {0.f, 0.f, 0.f, 0.f}, sd4 = {0.f, 0.f, 0.f, 0.f};
  const long nchunk = (a.rows + 63) / 64, stride = (long)gridDim.x * 4;
  for (long ch = (long)blockIdx.x * 4 + wave; ch < nchunk; ch += stride) {
    const long row = ch * 64 + lane;
    const bool ok = row < a.rows;
    float xh[8], dy[8], hid[16], dp[16];
    ffn8_load(a.x, row, ok, xh);     // rows past the end are zero: xhat = dy = dpre = 0 add nothing to the sums
    ffn8_load(a.dy, row, ok, dy);
    cfp w = (cfp)a.x16;
    asm volatile("" : "+s"(w));
    const float rstd = ffn8_ln(xh, a.ln_eps);
    ffn8_hidden<ACT>(w, xh, hid);
    {   // dhid = W2 . dy ; dpre = dhid * act'(pre)
      v2f dh[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) dh[k] = splat2(0.f);
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) dh[k] = ldw2(w, F8_W2T + o * 16 + 2 * k) * splat2(dy[o]) + dh[k];
#pragma unroll
      for (int h = 0; h < 16; ++h) dp[h] = dh[h >> 1][h & 1] * ffn_dact<ACT>(hid[h]);
    }
    // ---- weight gradients: transpose through LDS, contract the wave's 64 rows on the matrix pipe ----
    lds_sync();
#pragma unroll
    for (int c = 0; c < 8; ++c) { tA[c * F8_LD + lane] = xh[c]; tA[(8 + c) * F8_LD + lane] = dy[c]; }
#pragma unroll
    for (int h = 0; h < 16; ++h) tB[h * F8_LD + lane] = dp[h];
    lds_sync();
    v4f av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[j] = *reinterpret_cast<const v4f*>(tA + m * F8_LD + 16 * kq + 4 * j);
      bv[j] = *reinterpret_cast<const v4f*>(tB + m * F8_LD + 16 * kq + 4 * j);
    }
    lds_sync();
    sp4 += (bv[0] + bv[1]) + (bv[2] + bv[3]);
    sd4 += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
    for (int h = 0; h < 16; ++h) tB[h * F8_LD + lane] = hid[h];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) accT1[u & 1] = MFMA(av[j][u], bv[j][u], accT1[u & 1]);
    lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const v4f*>(tB + m * F8_LD + 16 * kq + 4 * j);
    // ---- dxhat = W1p . dpre ; LayerNorm backward ; dx = dy + ... ----
    v2f dxh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) dxh[k] = splat2(0.f);
#pragma unroll
    for (int h = 0; h < 16; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) dxh[k] = ldw2(w, F8_W1PT + h * 8 + 2 * k) * splat2(dp[h]) + dxh[k];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float d = dxh[c >> 1][c & 1]; m1 += d; m2 = fmaf(d, xh[c], m2); }
    m1 *= 0.125f; m2 *= 0.125f;
    float o8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o8[c] = dy[c] + rstd * (dxh[c >> 1][c & 1] - m1 - xh[c] * m2);
    if (ok) {
      *reinterpret_cast<float4*>(a.dx + row * 8) = make_float4(o8[0], o8[1], o8[2], o8[3]);
      *reinterpret_cast<float4*>(a.dx + row * 8 + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
    lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) accT2[u & 1] = MFMA(av[j][u], bv[j][u], accT2[u & 1]);
  }
  // ---- per-workgroup partial: the four waves add into one LDS image, one after the other (fixed order) ----
  __syncthreads();
  float* red = sm;   // [F8_PART]
  const v4f t1 = accT1[0] + accT1[1], t2 = accT2[0] + accT2[1];
  const v4f tt = kq < 2 ? t1 : t2;   // D[mrow = 4 kq + r][n = m]: rows 0-7 of T1, rows 8-15 of T2^T  ->  image index 16 mrow + n
  const float s1m = sum_over_q((sp4[0] + sp4[1]) + (sp4[2] + sp4[3]));   // s1[h = m]
  const float s2m = sum_over_q((sd4[0] + sd4[1]) + (sd4[2] + sd4[3]));   // m >= 8: s2[o = m - 8]
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
      const bool first = wv == 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* d = red + 16 * (4 * kq + r) + m;
        *d = first ? tt[r] : *d + tt[r];
      }
      if (kq == 0) {
        red[256 + m] = first ? s1m : red[256 + m] + s1m;
        if (m >= 8) red[264 + m] = first ? s2m : red[264 + m] + s2m;
      }
    }
    __syncthreads();
  }
  float* out = a.part + (size_t)blockIdx.x * F8_PART;
  for (int i = threadIdx.x; i < F8_PART; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ host glue --
#define FFN_NWG 256   // W >= 48: at most one backward workgroup per CU (4 waves, 1 per SIMD: the 512-register kernel)
// narrower rows need < 128 registers and ~20 KB of LDS per workgroup: several workgroups per CU hide each other's latency
static size_t ffn_nwg_cap(size_t W) { return W <= 16 ? 1024 : (W <= 32 ? 512 : FFN_NWG); }   // (measured: 256 -> 1024 workgroups: 128 -> 96 us at De = 8; 1280/2048 no better)

static size_t ffn_al(size_t x) { return (x + 63) & ~(size_t)63; }

extern "C" int egt_ffn_supported(const egt_ffn_desc* d) {
  if (!d || d->dtype != EGT_F32 || d->rows <= 0) return 0;
  if (d->matmul != EGT_MM_F32 && d->matmul != EGT_MM_BF16X3 && d->matmul != EGT_MM_BF16) return 0;
  if (d->width == 8) return d->matmul == EGT_MM_F32 && (d->activation == EGT_ACT_RELU || d->activation == EGT_ACT_ELU);   // VALU kernels: exact fp32 only
  if (d->width != 16 && d->width != 32 && d->width != 48 && d->width != 64) return 0;
  return d->activation == EGT_ACT_RELU || d->activation == EGT_ACT_ELU;
}

// [slab1 slab2 slab3 slab4 b1p | red | part x FFN_NWG | sA1 sA2 (bf16 modes)]
static size_t ffn_bf_slab_floats(size_t W) {   // sA1, sA3: [TH][NS1][2][256] floats each; sA2, sA4: [TW][TW][2][256]
  const size_t TW = W / 16, NS1 = (TW + 1) / 2;
  return 2 * (2 * TW * NS1 * 2 * 256) + 2 * (TW * TW * 2 * 256);
}
static int ffn_bf_prep_blocks(int W) {
  const int TW = W / 16, NS1 = (TW + 1) / 2;
  const int n1 = 2 * TW * NS1 * 512, n2 = TW * TW * 512;
  return ((n1 > n2 ? n1 : n2) + 255) / 256;
}
extern "C" size_t egt_ffn_workspace_bytes(const egt_ffn_desc* d) {
  if (!egt_ffn_supported(d)) return 0;
  const size_t W = d->width == 8 ? 16 : d->width, slab = 2 * W * W, part = 2 * slab + 3 * W;
  return (4 * slab + ffn_al(2 * W) + ffn_al(part) + ffn_nwg_cap(W) * part + ffn_bf_slab_floats(W) + 1280) * sizeof(float);
}

static int ffn_fill(const egt_ffn_desc* d, const egt_ffn_params* p, void* ws, FfnArgs& a) {
  if (!d || !p || !ws) EGT_FAIL(EGT_E_NULL, "desc/params/workspace is NULL");
  if (!egt_ffn_supported(d))
    EGT_FAIL(EGT_E_SHAPE, "fused FFN covers widths 16/32/48/64 (and 8 with exact fp32 products), fp32, relu/elu "
                          "(got width %d, rows %lld, act %d, matmul %d)", d->width, (long long)d->rows, d->activation, d->matmul);
  if (!p->norm_gamma || !p->norm_beta || !p->lr1_kernel || !p->lr1_bias || !p->lr2_kernel || !p->lr2_bias)
    EGT_FAIL(EGT_E_NULL, "an FFN parameter pointer is NULL");
  a = FfnArgs{};
  a.rows = d->rows; a.ln_eps = d->ln_eps; a.W = d->width;
  a.gamma = (const float*)p->norm_gamma; a.beta = (const float*)p->norm_beta;
  a.W1 = (const float*)p->lr1_kernel; a.b1 = (const float*)p->lr1_bias;
  a.W2 = (const float*)p->lr2_kernel; a.b2 = (const float*)p->lr2_bias;
  const size_t W = d->width == 8 ? 16 : d->width, slab = 2 * W * W, part = 2 * slab + 3 * W;
  float* w = (float*)ws;
  a.slab1 = w; a.slab2 = w + slab; a.slab3 = w + 2 * slab; a.slab4 = w + 3 * slab;
  a.b1p = w + 4 * slab;
  a.red = a.b1p + ffn_al(2 * W);
  a.part = a.red + ffn_al(part);
  a.mm = d->matmul;
  {
    const size_t TW = W / 16, NS1 = (TW + 1) / 2;
    float* bf = a.part + ffn_nwg_cap(W) * part;
    const size_t a1 = 2 * TW * NS1 * 2 * 256, a2 = TW * TW * 2 * 256;
    a.sA1 = reinterpret_cast<uint16_t*>(bf);
    a.sA2 = reinterpret_cast<uint16_t*>(bf + a1);
    a.sA3 = reinterpret_cast<uint16_t*>(bf + a1 + a2);
    a.sA4 = reinterpret_cast<uint16_t*>(bf + 2 * a1 + a2);
    a.x16 = bf + 2 * a1 + 2 * a2;
  }
  a.part_len = (int)part;
  if (d->width == 8) {   // one row per lane (k_ffn8_*)
    a.row8 = 1; a.part_len = F8_PART;
    const long want = (((a.rows + 63) / 64) + 3) / 4;
    a.nwg = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
    return EGT_OK;
  }
  {   // backward workgroups: one per CU for large inputs; small inputs (node channels) one tile
      // per wave (a tile is ~16 us of dependent work: spreading beats amortising the slab staging)
    const long ntiles = (a.rows + 15) / 16;
    const long want = (ntiles + 3) / 4;
    const long cap = (long)ffn_nwg_cap(W);
    a.nwg = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  }
  return EGT_OK;
}

template <int W>
static void ffn_launch_fwd(const FfnArgs& a, int act, hipStream_t st) {
  const size_t lds = (2 * (size_t)(2 * W * W) + 3 * W + 8 * 2 * 16 * W) * 4;
  const long ntiles = (a.rows + 15) / 16;
  const long wantf = (ntiles + 7) / 8;
  const int grid = (int)(wantf < 1 ? 1 : (wantf > 256 ? 256 : wantf));
  if (act == EGT_ACT_RELU) {
    EGT_MAX_LDS_ONCE(k_ffn_fwd<W, EGT_ACT_RELU>);
    EGT_LAUNCH("k_ffn_fwd", (k_ffn_fwd<W, EGT_ACT_RELU>), dim3(grid), dim3(512), lds, st, a);
  } else {
    EGT_MAX_LDS_ONCE(k_ffn_fwd<W, EGT_ACT_ELU>);
    EGT_LAUNCH("k_ffn_fwd", (k_ffn_fwd<W, EGT_ACT_ELU>), dim3(grid), dim3(512), lds, st, a);
  }
}

template <int W>
static void ffn_launch_fwd_bf(const FfnArgs& a, int act, hipStream_t st) {
  constexpr int TW = W / 16, NS1 = (TW + 1) / 2;
  const size_t lds = ((size_t)(2 * TW * NS1 * 2 * 256 + TW * TW * 2 * 256) + 3 * W + 8 * 2 * 16 * W) * 4;
  const long ntiles = (a.rows + 15) / 16;
  const long wantf = (ntiles + 7) / 8;
  const int grid = (int)(wantf < 1 ? 1 : (wantf > 512 ? 512 : wantf));
  EGT_LAUNCH("k_ffn_prep", k_ffn_prep_bf<W>, dim3(ffn_bf_prep_blocks(W)), dim3(256), 0, st, a);
#define FBF(ACT_, SPLIT_)                                                                                             \
  do {                                                                                                                \
    EGT_MAX_LDS_ONCE(k_ffn_fwd_bf<W, ACT_, SPLIT_>); \
    EGT_LAUNCH("k_ffn_fwd", (k_ffn_fwd_bf<W, ACT_, SPLIT_>), dim3(grid), dim3(512), lds, st, a);                       \
  } while (0)
  const bool split = a.mm == EGT_MM_BF16X3;
  if (act == EGT_ACT_RELU) { if (split) FBF(EGT_ACT_RELU, true); else FBF(EGT_ACT_RELU, false); }
  else { if (split) FBF(EGT_ACT_ELU, true); else FBF(EGT_ACT_ELU, false); }
#undef FBF
}

template <int W, int MM>
static void ffn_launch_bwd_mm(const FfnArgs& a, int act, hipStream_t st) {
  constexpr int TW = W / 16, NS1 = (TW + 1) / 2;
  constexpr size_t slabs = MM ? (size_t)(2 * (2 * TW * NS1 * 2 * 256) + TW * TW * 2 * 256) : 3 * (size_t)(2 * W * W);
  // the per-workgroup partial (2 * 2W*W + 3W floats) is staged over the slab area at the end
  constexpr size_t part = 2 * (size_t)(2 * W * W) + 3 * W;
  const size_t lds = ((slabs > part ? slabs : part) + 2 * W + 4 * 3 * 16 * W) * 4;
  if (act == EGT_ACT_RELU) {
    EGT_MAX_LDS_ONCE(k_ffn_bwd<W, EGT_ACT_RELU, MM>);
    EGT_LAUNCH("k_ffn_bwd", (k_ffn_bwd<W, EGT_ACT_RELU, MM>), dim3(a.nwg), dim3(256), lds, st, a);
  } else {
    EGT_MAX_LDS_ONCE(k_ffn_bwd<W, EGT_ACT_ELU, MM>);
    EGT_LAUNCH("k_ffn_bwd", (k_ffn_bwd<W, EGT_ACT_ELU, MM>), dim3(a.nwg), dim3(256), lds, st, a);
  }
}
template <int W>
static void ffn_launch_bwd(const FfnArgs& a, int act, hipStream_t st) {
  if (a.mm == EGT_MM_BF16X3) ffn_launch_bwd_mm<W, EGT_MM_BF16X3>(a, act, st);
  else if (a.mm == EGT_MM_BF16) ffn_launch_bwd_mm<W, EGT_MM_BF16>(a, act, st);
  else ffn_launch_bwd_mm<W, EGT_MM_F32>(a, act, st);
}

#define FFN_DISPATCH_W(width, CALL)               \
  switch (width) {                                \
    case 16: { constexpr int W = 16; CALL; } break; \
    case 32: { constexpr int W = 32; CALL; } break; \
    case 48: { constexpr int W = 48; CALL; } break; \
    default: { constexpr int W = 64; CALL; } break; \
  }

extern "C" int egt_ffn_fwd(const egt_ffn_desc* desc, const egt_ffn_params* params, const void* x, void* y,
                           void* workspace, void* stream) {
  FfnArgs a;
  int rc = ffn_fill(desc, params, workspace, a);
  if (rc) return rc;
  if (!x || !y) EGT_FAIL(EGT_E_NULL, "x/y is NULL");
  a.x = (const float*)x; a.y = (float*)y;
  hipStream_t st = (hipStream_t)stream;
  if (a.row8) {
    EGT_LAUNCH("k_ffn_prep", k_ffn8_prep, dim3(1), dim3(128), 0, st, a);
    const long want = (((a.rows + 63) / 64) + 3) / 4;
    const int grid = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    if (desc->activation == EGT_ACT_RELU) EGT_LAUNCH("k_ffn_fwd", k_ffn8_fwd<EGT_ACT_RELU>, dim3(grid), dim3(256), 0, st, a);
    else EGT_LAUNCH("k_ffn_fwd", k_ffn8_fwd<EGT_ACT_ELU>, dim3(grid), dim3(256), 0, st, a);
    EGT_HIP_LAUNCH_CHECK("egt_ffn_fwd");
    return EGT_OK;
  }
  if (desc->matmul != EGT_MM_F32) {
    FFN_DISPATCH_W(desc->width, ffn_launch_fwd_bf<W>(a, desc->activation, st));
  } else {
    FFN_DISPATCH_W(desc->width, EGT_LAUNCH("k_ffn_prep", k_ffn_prep<W>, dim3((2 * W * W + 255) / 256), dim3(256), 0, st, a));
    FFN_DISPATCH_W(desc->width, ffn_launch_fwd<W>(a, desc->activation, st));
  }
  EGT_HIP_LAUNCH_CHECK("egt_ffn_fwd");
  return EGT_OK;
}

extern "C" int egt_ffn_bwd(const egt_ffn_desc* desc, const egt_ffn_params* params, const void* x, const void* dy,
                           void* dx, const egt_ffn_params* grads, void* workspace, void* stream) {
  FfnArgs a;
  int rc = ffn_fill(desc, params, workspace, a);
  if (rc) return rc;
  if (!x || !dy || !dx || !grads) EGT_FAIL(EGT_E_NULL, "x/dy/dx/grads is NULL");
  if (!grads->norm_gamma || !grads->norm_beta || !grads->lr1_kernel || !grads->lr1_bias || !grads->lr2_kernel || !grads->lr2_bias)
    EGT_FAIL(EGT_E_NULL, "an FFN gradient pointer is NULL");
  a.x = (const float*)x; a.dy = (const float*)dy; a.dx = (float*)dx;
  a.g_gamma = (float*)grads->norm_gamma; a.g_beta = (float*)grads->norm_beta;
  a.g_W1 = (float*)grads->lr1_kernel; a.g_b1 = (float*)grads->lr1_bias;
  a.g_W2 = (float*)grads->lr2_kernel; a.g_b2 = (float*)grads->lr2_bias;
  hipStream_t st = (hipStream_t)stream;
  const bool prepared = (desc->flags & EGT_FFN_WS_PREPARED) != 0;   // the forward's workspace: its prepared operands are still there
  if (a.row8) {
    if (!prepared) EGT_LAUNCH("k_ffn_prep", k_ffn8_prep, dim3(1), dim3(128), 0, st, a);
    if (desc->activation == EGT_ACT_RELU) EGT_LAUNCH("k_ffn_bwd", k_ffn8_bwd<EGT_ACT_RELU>, dim3(a.nwg), dim3(256), 0, st, a);
    else EGT_LAUNCH("k_ffn_bwd", k_ffn8_bwd<EGT_ACT_ELU>, dim3(a.nwg), dim3(256), 0, st, a);
    EGT_LAUNCH("k_ffn_sum", k_ffn_sum, dim3((F8_PART + 63) / 64), dim3(1024), 0, st, a);
    EGT_LAUNCH("k_ffn_param_grads", k_ffn_param_grads, dim3(1), dim3(1024), 0, st, a);
    EGT_HIP_LAUNCH_CHECK("egt_ffn_bwd");
    return EGT_OK;
  }
  if (!prepared) {   // (the bf16 modes read their own slabs and b1p only: k_ffn_prep_bf writes all of that)
    if (desc->matmul == EGT_MM_F32) {
      FFN_DISPATCH_W(desc->width, EGT_LAUNCH("k_ffn_prep", k_ffn_prep<W>, dim3((2 * W * W + 255) / 256), dim3(256), 0, st, a));
    } else {
      FFN_DISPATCH_W(desc->width, EGT_LAUNCH("k_ffn_prep", k_ffn_prep_bf<W>, dim3(ffn_bf_prep_blocks(W)), dim3(256), 0, st, a));
    }
  }
  FFN_DISPATCH_W(desc->width, ffn_launch_bwd<W>(a, desc->activation, st));
  const int part = a.part_len;
  EGT_LAUNCH("k_ffn_sum", k_ffn_sum, dim3((part + 63) / 64), dim3(1024), 0, st, a);
  EGT_LAUNCH("k_ffn_param_grads", k_ffn_param_grads, dim3(1), dim3(1024), 0, st, a);
  EGT_HIP_LAUNCH_CHECK("egt_ffn_bwd");
  return EGT_OK;
}
