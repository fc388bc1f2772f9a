// Narrow edge channels (De = 8, H = 8, d <= 8: BASELINE configs 3 and 4) -- pair kernels on the VALU.
//
// A De = 8 pair row is 32 bytes: the 16-pair MFMA tile of the wide kernels is 512 B, its per-tile
// skeleton and dependent phases (LDS round trip -> LayerNorm over lanes -> MFMA chain -> ...) cost more
// than the arithmetic, and at ~240 VGPRs only two waves per SIMD hide the HBM latency of tiles that
// small (measured: 1.1 TB/s, 0.08 of the roof).  These kernels turn the mapping around:
//   lane = 4 p + q,  p = query row (forward) / key (backward) of the wave's 16,  q = head pair AND channel pair:
//   lane (p, q) owns heads 2q, 2q+1 and edge channels 2q, 2q+1 of ONE pair per step, all in registers.
//   * e arrives by plain 8-byte global loads straight into the lane (no LDS tile, no transposition);
//   * LayerNorm statistics, the projection sums and dense_edge_r are 2-channel / 2-head partial sums in
//     the lane, combined across the QUAD with DPP quad_perm adds.  Each lane evaluates the partial sums in
//     quad-RELATIVE order (slot g of lane q holds the partial for quarter g ^ q), so the combine is three
//     DPP adds per value and needs no selects;
//   * the forward keeps a whole query row's softmax state in the lane (online softmax over the key loop:
//     no cross-lane reduction at all); the four waves of a workgroup split the key range and merge their
//     (max, sum, A.V) triples once at the end;
//   * ~100-130 VGPRs: 3-4 waves per SIMD, 20 KB of LDS per workgroup (K / V travel through a 2 KB
//     wave-private LDS chunk four keys at a time).
// Same BlockArgs, saved tensors, partial-buffer layouts, mask order / RNG stream and node-side epilogue /
// prologue as the wide kernels: the dispatch in launch_fwd / launch_bwd is the only difference.
// Own translation unit: built with -fno-slp-vectorize (build.py) -- hipcc otherwise pairs the scalar adds into
// v_pk_add_f32, which cannot take a DPP operand, and every quad exchange becomes v_mov_dpp + add.
#include "egt_common.h"

#include "egt_block.h"
#include "egt_block_dev.h"

#define NRW_DE 8
#ifndef NRW_ABL          // timing ablations (measurement builds only, tools/build_variant.sh -DNRW_ABL=<bits>):
#define NRW_ABL 0        // 1 no e' stores, 2 no e loads, 4 no K/V chunk traffic, 8 no mask hash, 16 no exp/sigmoid, 32 no epilogue
#endif
#define NRW_KB 4                      // keys per K/V chunk and per e prefetch block (one 128-byte line of a pair row)
#define NRW_KV_CHUNK (NRW_KB * 128)   // floats: [key][K 64 | V 64]


// quad exchange: value of lane (q ^ X).  bound_ctrl + full masks: `old` is dead, so the DPP move folds into the add that uses it
template <int CTRL>
__device__ __forceinline__ float nrw_quad(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// sum over the quad of slot g of lane (q ^ g): own + xor1 + xor2 + xor3 partner slots
__device__ __forceinline__ float nrw_combine(float s0, float s1, float s2, float s3) {
  float t = s0 + nrw_quad<0xB1>(s1);
  t += nrw_quad<0x4E>(s2);
  t += nrw_quad<0x1B>(s3);
  return t;
}
__device__ __forceinline__ float nrw_quad_sum(float v) {
  v += nrw_quad<0xB1>(v);
  v += nrw_quad<0x4E>(v);
  return v;
}
// plain v_max_f32 / v_min_f32 (fmaxf's llvm.maxnum canonicalises both operands first: three instructions per max)
__device__ __forceinline__ float nrw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nrw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <bool BF> struct NrwLd;
template <> struct NrwLd<false> {
  typedef float2 raw;
  static __device__ __forceinline__ raw load(const void* base, size_t pair, int q) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(base) + pair * NRW_DE + 2 * q);
  }
  static __device__ __forceinline__ float2 cvt(raw r) { return r; }
  static __device__ __forceinline__ void store(void* base, size_t pair, int q, float2 v) {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + pair * NRW_DE + 2 * q) = v;
  }
};
template <> struct NrwLd<true> {
  typedef uint32_t raw;
  static __device__ __forceinline__ raw load(const void* base, size_t pair, int q) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(base) + pair * NRW_DE + 2 * q);
  }
  static __device__ __forceinline__ float2 cvt(raw r) { return make_float2(__uint_as_float(r << 16), __uint_as_float(r & 0xFFFF0000u)); }
  static __device__ __forceinline__ void store(void* base, size_t pair, int q, float2 v) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(base) + pair * NRW_DE + 2 * q) = f2_to_bf2(v.x, v.y);
  }
};

// feature sets with compile-time flags (the run-time flag tests cost scalar branches / selects in every step)
#define NRW_F_GATED 1
#define NRW_F_CLIP 2
#define NRW_F_RUNTIME (-1)

// ------------------------------------------------------------------------------------ forward ---
// Workgroup = (graph b, 16 query rows), 4 waves = the four quarters of the key range.  A step = one block of
// NRW_KB keys: logits of the four keys first, then ONE softmax rescale for the block (flash-style blocking).
// LDS: [NRW_FWD_AREA] = [4 waves][NRW_KV_CHUNK] K/V chunks + [4][NRW_KB] key-mask adds during the key loop, the
//      merge buffer [3][20][64] after it, then the epilogue's staging rows; [16][QS_LD] V_att rows for the epilogue.
#define NRW_FWD_AREA (3 * 20 * 64)
template <bool BF, int FEAT>
__global__ void __launch_bounds__(256, 3) k_narrow_fwd(BlockArgs a) {
  typedef NrwLd<BF> LD;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane >> 2, q = lane & 3;
  const int N = a.N;
  const int lgroups = (N + 15) / 16;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / lgroups, lg = wg % lgroups;
  float* kvw = sm + wave * NRW_KV_CHUNK;
  float* kmw = sm + 4 * NRW_KV_CHUNK + wave * NRW_KB;
  float* qs = sm + NRW_FWD_AREA;
  const bool gated = FEAT >= 0 ? (FEAT & NRW_F_GATED) != 0 : (a.flags & EGT_BF_GATE) != 0;
  const bool clip = FEAT >= 0 ? (FEAT & NRW_F_CLIP) != 0 : (a.flags & EGT_BF_CLIP) != 0;
  const bool ln_on = (a.flags & EGT_BF_NO_EDGE_LN) == 0;
  const int l = lg * 16 + p;
  const bool row_ok = l < N;
  const size_t rowl = (size_t)b * N + min(l, N - 1);

  // ---- lane constants: Q of the row, LN-folded projection weights and dense_edge_r in quad-relative order ----
  float Qf[16];
  {
    const float4* qp = reinterpret_cast<const float4*>(a.qkvp + rowl * QKVP + q * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float4 v = qp[u]; Qf[4*u] = v.x; Qf[4*u+1] = v.y; Qf[4*u+2] = v.z; Qf[4*u+3] = v.w; }
  }
  float wp[2][4][4], wr[2][4][2], c2r[4], brr[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) wp[c][g][r] = a.pw[(2 * q + c) * 16 + 4 * (g ^ q) + r];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 2; ++c) wr[j][g][c] = a.Wr[(2 * q + j) * NRW_DE + 2 * (g ^ q) + c];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[16 * 16 + 4 * q + r];
  brr[0] = a.br[2 * q]; brr[1] = a.br[2 * q + 1];

  // ---- the wave's key blocks ----
  const int nblk = (N + NRW_KB - 1) / NRW_KB;
  const int blk0 = (wave * nblk) >> 2, blk1 = ((wave + 1) * nblk) >> 2;
  float mx[2] = {-3.0e38f, -3.0e38f}, sum[2] = {0.f, 0.f}, O[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) O[k] = 0.f;

  const size_t pair_row = rowl * N;
  typename LD::raw eb[NRW_KB];
  float4 kvr[2];
  float kmr = 0.f;
  auto fetch = [&](int blk) {
    const int m0 = blk * NRW_KB;
#pragma unroll
    for (int kk = 0; kk < NRW_KB; ++kk) eb[kk] = LD::load(a.e, (NRW_ABL & 2) ? (size_t)(q + kk) : pair_row + min(m0 + kk, N - 1), q);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = lane + 64 * u, key = f >> 5, piece = f & 31;
      if (!(NRW_ABL & 4) || blk == blk0) kvr[u] = *reinterpret_cast<const float4*>(a.qkvp + ((size_t)b * N + min(m0 + key, N - 1)) * QKVP + 64 + piece * 4);
    }
    if (a.km) kmr = (a.km[(size_t)b * N + min(m0 + (lane & 3), N - 1)] == 0) ? -EGT_NEG : 0.0f;
  };
  if (blk0 < blk1) fetch(blk0);

  // one block of keys; nv = number of real keys in it (NRW_KB except in the graph's last block)
  auto block = [&](int blk, int nv) {
    const int m0 = blk * NRW_KB;
    // commit the chunk to the wave's LDS slot (DS operations of a wave retire in order: the previous
    // chunk's reads are behind these writes), take the e registers, then put the next block in flight
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = lane + 64 * u;
      *reinterpret_cast<float4*>(kvw + (f >> 5) * 128 + (f & 31) * 4) = kvr[u];
    }
    if (lane < NRW_KB) kmw[lane] = kmr;
    float2 ev[NRW_KB];
#pragma unroll
    for (int kk = 0; kk < NRW_KB; ++kk) ev[kk] = LD::cvt(eb[kk]);
    if (blk + 1 < blk1) fetch(blk + 1);
    lds_sync();
    float xl[NRW_KB][2], gl[NRW_KB][2];
#pragma unroll
    for (int kk = 0; kk < NRW_KB; ++kk) {
      const int m = m0 + kk;
      // ---- norm_edge: the pair's 8 channels sit in the quad, two per lane (two-pass moments) ----
      float x0 = ev[kk].x, x1 = ev[kk].y;
      const float mu = ln_on ? nrw_quad_sum(x0 + x1) * 0.125f : 0.0f;
      x0 -= mu; x1 -= mu;
      const float var = nrw_quad_sum(fmaf(x0, x0, x1 * x1)) * 0.125f;
      const float rstd = ln_on ? __builtin_amdgcn_rsqf(var + a.ln_eps) : 1.0f;
      x0 *= rstd; x1 *= rstd;
      // ---- [attention_gates | dense_edge_b]: acc[r] = column 4q + r of the pair ----
      float acc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s0 = fmaf(x1, wp[1][0][r], x0 * wp[0][0][r]);
        const float s1 = fmaf(x1, wp[1][1][r], x0 * wp[0][1][r]);
        const float s2 = fmaf(x1, wp[1][2][r], x0 * wp[0][2][r]);
        const float s3 = fmaf(x1, wp[1][3][r], x0 * wp[0][3][r]);
        acc[r] = nrw_combine(s0 + c2r[r], s1, s2, s3);
      }
      // ---- scaled QK^T, clip, + E (egt_layers.py:79-86) ----
      float Kf[16];
      {
        const float4* kp = reinterpret_cast<const float4*>(kvw + kk * 128 + q * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float4 v = kp[u]; Kf[4*u] = v.x; Kf[4*u+1] = v.y; Kf[4*u+2] = v.z; Kf[4*u+3] = v.w; }
      }
      float hh[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float dot = Qf[j] * Kf[j];
#pragma unroll
        for (int k = 1; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
        float ah = dot * a.scale;
        if (clip) ah = nrw_min(nrw_max(ah, a.clip_lo), a.clip_hi);
        hh[j] = ah + acc[2 * j + 1];
        xl[kk][j] = hh[j];
        gl[kk][j] = acc[2 * j];
      }
      const MaskRegs mr{make_float2(1.f, 1.f), 0};
      if (!(NRW_ABL & 8)) apply_masks<false>(a, kmw[kk], mr, (pair_row + m) * BH, q, xl[kk], gl[kk]);
      if (kk >= nv) { xl[kk][0] = -3.0e38f; xl[kk][1] = -3.0e38f; }   // past the graph's last key: probability exactly 0
      // ---- dense_edge_r + res_edge: e' = e + H_hat.Wr + br, the lane's two channels ----
      float2 eo;
      {
        const float t0 = nrw_combine(fmaf(hh[1], wr[1][0][0], fmaf(hh[0], wr[0][0][0], brr[0])), fmaf(hh[1], wr[1][1][0], hh[0] * wr[0][1][0]),
                                     fmaf(hh[1], wr[1][2][0], hh[0] * wr[0][2][0]), fmaf(hh[1], wr[1][3][0], hh[0] * wr[0][3][0]));
        const float t1 = nrw_combine(fmaf(hh[1], wr[1][0][1], fmaf(hh[0], wr[0][0][1], brr[1])), fmaf(hh[1], wr[1][1][1], hh[0] * wr[0][1][1]),
                                     fmaf(hh[1], wr[1][2][1], hh[0] * wr[0][2][1]), fmaf(hh[1], wr[1][3][1], hh[0] * wr[0][3][1]));
        eo.x = ev[kk].x + t0;
        eo.y = ev[kk].y + t1;
      }
      if (row_ok && kk < nv && !(NRW_ABL & 1)) LD::store(a.e_out, pair_row + m, q, eo);
      if ((NRW_ABL & 1) && eo.x == 123.456f) LD::store(a.e_out, pair_row + m, q, eo);
    }
    // ---- one online-softmax step for the block, x gate, A.V: the row's state never leaves the lane ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float mn = nrw_max(nrw_max(mx[j], nrw_max(xl[0][j], xl[1][j])), nrw_max(xl[2][j], xl[3][j]));
      const float alpha = (NRW_ABL & 16) ? (mx[j] - mn) : __expf(mx[j] - mn);
      mx[j] = mn;
      sum[j] *= alpha;
#pragma unroll
      for (int k = 0; k < 8; ++k) O[2 * k + j] *= alpha;
    }
#pragma unroll
    for (int kk = 0; kk < NRW_KB; ++kk) {
      float Vf[16];
      {
        const float4* vp = reinterpret_cast<const float4*>(kvw + kk * 128 + 64 + q * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float4 v = vp[u]; Vf[4*u] = v.x; Vf[4*u+1] = v.y; Vf[4*u+2] = v.z; Vf[4*u+3] = v.w; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float pe = (NRW_ABL & 16) ? (xl[kk][j] - mx[j]) : __expf(xl[kk][j] - mx[j]);
        sum[j] += pe;
        const float av = gated ? pe * ((NRW_ABL & 16) ? gl[kk][j] : egt_sigmoid(gl[kk][j])) : pe;
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(av, Vf[2 * k + j], O[2 * k + j]);
      }
    }
  };
  {
    const int nfull = N / NRW_KB;                 // blocks below nfull hold NRW_KB real keys
    const int bfull = min(blk1, nfull);
    for (int blk = blk0; blk < bfull; ++blk) block(blk, NRW_KB);
    if (blk1 > nfull && blk0 <= nfull) block(nfull, N - nfull * NRW_KB);   // the graph's ragged last block
  }
  // ---- merge the four key quarters (waves 1..3 -> LDS -> wave 0), write V_att / statistics ----
  __syncthreads();   // every wave is done with its K/V chunk: the area becomes the merge buffer
  float* mg = sm;    // [3][20][64]
  if (wave > 0) {
    float* o = mg + (wave - 1) * 20 * 64 + lane;
    o[0] = mx[0]; o[64] = mx[1]; o[128] = sum[0]; o[192] = sum[1];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[(4 + k) * 64] = O[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float* o = mg + w * 20 * 64 + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mo = o[j * 64], so = o[(2 + j) * 64];
        const float mn = fmaxf(mx[j], mo);
        const float f0 = __expf(mx[j] - mn), f1 = __expf(mo - mn);
        mx[j] = mn;
        sum[j] = fmaf(sum[j], f0, so * f1);
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(O[2 * k + j], f0, o[(4 + 2 * k + j) * 64] * f1);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 vo = make_float2(O[2 * k] / sum[0], O[2 * k + 1] / sum[1]);
      if (row_ok && k < a.DK) *reinterpret_cast<float2*>(a.v_att + rowl * a.Dh + k * BH + 2 * q) = vo;
      if (a.epi) *reinterpret_cast<float2*>(qs + p * QS_LD + k * BH + 2 * q) = vo;
    }
    if (row_ok) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* st = a.stats + (rowl * BH + 2 * q + j) * 4;
        st[0] = mx[j];
        st[1] = sum[j];
      }
    }
  }
  // node-side epilogue on the 16 rows (its own lane roles: MFMA layout); its staging rows reuse the merge area
  if (a.epi && !(NRW_ABL & 32)) fwd_node_epilogue(a, sm, qs, b, lg, N, wave, lane & 15, lane >> 4);
}
static_assert(4 * NRW_KV_CHUNK + 4 * NRW_KB <= NRW_FWD_AREA, "key-loop buffers fit the merge area");

// a.epi must already hold the epilogue the geometry allows (launch_fwd decides)
void egt_narrow_launch_fwd(BlockArgs& a, hipStream_t st) {
  const dim3 grid(a.B * ((a.N + 15) / 16)), block(256);
  const size_t lds = ((size_t)NRW_FWD_AREA + 16 * QS_LD) * 4;
  const int full = NRW_F_GATED | NRW_F_CLIP;
  const int feat = ((a.flags & EGT_BF_GATE) ? NRW_F_GATED : 0) | ((a.flags & EGT_BF_CLIP) ? NRW_F_CLIP : 0);
#define NRW_FWD(BF_, FEAT_) EGT_LAUNCH("k_block_fwd", (k_narrow_fwd<BF_, FEAT_>), grid, block, lds, st, a)
  if (a.bf16) { if (feat == full) NRW_FWD(true, NRW_F_GATED | NRW_F_CLIP); else NRW_FWD(true, NRW_F_RUNTIME); }
  else { if (feat == full) NRW_FWD(false, NRW_F_GATED | NRW_F_CLIP); else NRW_FWD(false, NRW_F_RUNTIME); }
#undef NRW_FWD
}
