// Edge-channel projections around the inner op (composed block path):
//   egt_edge_proj_*   : [norm_edge] -> attention_gates (De->H) , dense_edge_b (De->H)
//                       graph_xformer_model_base.py:195,201-204,149-162
//   egt_edge_update_* : e' = e + H_hat·Wr + br   (dense_edge_r + res_edge, :214-218)
// The projections and their backward contractions run on the matrix cores (see "MFMA tiles"
// below); edge_update forward is a pure streaming kernel.  Weight gradients are per-workgroup
// partials reduced by a deterministic second kernel (no float atomics).
#include "egt_common.h"
#include "egt_tile.h"
#ifndef EDGE_NT_IN
#define EDGE_NT_IN true   // cache-policy hint (egt_tile.h): k_edge_update_fwd reads its e rows once (115 -> 106 us at config 5; the same hint
                          // on the MFMA-fragment loads of k_edge_proj_* measured as a loss, 180 -> 196 us: not applied there)
#endif

#define EDGE_H 8

struct EdgeArgs {
  long rows;
  int De;
  uint32_t flags;
  int act;
  float act_alpha, ln_eps;
  const float *e, *gamma, *beta, *Wg, *bg, *We, *be, *Wr, *br, *h_hat, *E_out, *dG, *dE, *de_out;
  float *G_out, *E_o, *e_o, *d_e, *d_h_hat, *ws;
  float *d_gamma, *d_beta, *d_Wg, *d_bg, *d_We, *d_be, *d_Wr, *d_br;
  int n_partials;
};

__device__ __forceinline__ float act_fwd(int act, float alpha, float x) {
  switch (act) {
    case EGT_ACT_LRELU: return x >= 0.f ? x : alpha * x;
    case EGT_ACT_RELU: return fmaxf(x, 0.f);
    case EGT_ACT_ELU: return x > 0.f ? x : (__expf(x) - 1.0f);
    default: return x;
  }
}
// derivative from the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(int act, float alpha, float y) {
  switch (act) {
    case EGT_ACT_LRELU: return y >= 0.f ? 1.0f : alpha;
    case EGT_ACT_RELU: return y > 0.f ? 1.0f : 0.0f;
    case EGT_ACT_ELU: return y > 0.f ? 1.0f : (y + 1.0f);
    default: return 1.0f;
  }
}

// ---------------------------------------------------------------- MFMA tiles ---
// The three heavy kernels (proj forward, proj backward, update backward) run on
// v_mfma_f32_16x16x4_f32 in the transposed formulation of the fused block kernels: the 16 edge
// rows of a tile sit on the MFMA column axis, lane (p = lane&15, q = lane>>4) owns row p and
// channels 16t + 4q + {0..3} -- one aligned 16-byte global load per fragment, straight into the
// register that feeds the matrix core.  LayerNorm statistics are two cross-lane sums over q.
// The weight-gradient contractions run over the ROW axis, so that one operand needs the rows on
// the contraction (q) axis: the tile takes one round trip through a wave-private LDS tile
// (row stride = 16 mod 32 floats: the transposed ds_read_b32 are conflict-free).
#define EDGE_WAVES 8   // waves per workgroup in the backward kernels

template <int DE> struct EdgeGeo {
  static constexpr int T = (DE + 15) / 16;
  static constexpr int LD = (T * 16) % 32 == 16 ? T * 16 : T * 16 + 16;   // 16 mod 32
};

// fragments of row `rc` (already clamped); channels >= DE (only DE = 8) read as zero
template <int DE>
__device__ __forceinline__ void frag_gload(float4 (&x)[EdgeGeo<DE>::T], const float* src, long rc, int q) {
#pragma unroll
  for (int t = 0; t < EdgeGeo<DE>::T; ++t) {
    const int c = 16 * t + 4 * q;
    const bool ok = c < DE;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)rc * DE + (ok ? c : 0));
    x[t] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
}
__device__ __forceinline__ float f4c(const float4& v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; }

// column i of the concatenated projection [attention_gates | dense_edge_b] for channel c
__device__ __forceinline__ float wcat(const EdgeArgs& a, bool gates, int c, int i) {
  return i < EDGE_H ? (gates ? a.Wg[c * EDGE_H + i] : 0.f) : a.We[c * EDGE_H + (i - EDGE_H)];
}

// ------------------------------------------------------------- proj forward ---
// out[i][row] = sum_c (gamma_c Wcat[c][i]) xhat[row][c] + (b_i + sum_c beta_c Wcat[c][i]):
// the folded weights are the A operand (4*T registers per lane, loaded once per wave).
template <int DE>
__global__ void __launch_bounds__(256) k_edge_proj_fwd(EdgeArgs a) {
  constexpr int T = EdgeGeo<DE>::T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const bool ln = (a.flags & EGT_EP_LAYERNORM) != 0, gates = (a.flags & EGT_EP_GATES) != 0;
  float wA[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * q + r;
      wA[t][r] = c < DE ? wcat(a, gates, c, p) * (ln ? a.gamma[c] : 1.0f) : 0.f;
    }
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * q + r;
    float b = i < EDGE_H ? (gates ? a.bg[i] : 0.f) : a.be[i - EDGE_H];
    if (ln)
      for (int c = 0; c < DE; ++c) b = fmaf(a.beta[c], wcat(a, gates, c, i), b);
    bias[r] = b;
  }
  const long ntiles = (a.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 16 + p;
    float4 x[T];
    frag_gload<DE>(x, a.e, row < a.rows ? row : a.rows - 1, q);
    ln_frags<DE>(x, q, a.ln_eps, ln);
    v4f acc = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc = MFMA(wA[t][0], x[t].x, acc);
      acc = MFMA(wA[t][1], x[t].y, acc);
      acc = MFMA(wA[t][2], x[t].z, acc);
      acc = MFMA(wA[t][3], x[t].w, acc);
    }
    if (row < a.rows) {   // lane holds outputs 4q..4q+3 of its row: q < 2 gates, q >= 2 edge bias
      if (q < 2) {
        if (gates) *reinterpret_cast<float4*>(a.G_out + (size_t)row * EDGE_H + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
        *reinterpret_cast<float4*>(a.E_o + (size_t)row * EDGE_H + 4 * (q - 2)) =
            make_float4(act_fwd(a.act, a.act_alpha, acc[0]), act_fwd(a.act, a.act_alpha, acc[1]),
                        act_fwd(a.act, a.act_alpha, acc[2]), act_fwd(a.act, a.act_alpha, acc[3]));
      }
    }
  }
}

// ------------------------------------------------------------ proj backward ---
// partial layout per workgroup: That[DE][16] then s[16]
//   dxh[c][row]  = gamma_c sum_i Wcat[c][i] dpre[row][i]      (A = folded weights, B = dpre in place)
//   de           = LayerNorm backward of dxh, in the lane's fragments
//   That[c][i]  += sum_row xhat[row][c] dpre[row][i]          (both operands transposed through LDS)
template <int DE>
__global__ void __launch_bounds__(64 * EDGE_WAVES) k_edge_proj_bwd(EdgeArgs a) {
  constexpr int T = EdgeGeo<DE>::T, LD = EdgeGeo<DE>::LD, PSZ = DE * 16 + 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  float* xs = smem + wave * (16 * LD + 256);   // xhat tile [16][LD]
  float* ds = xs + 16 * LD;                    // dpre tile [16][16]
  const bool ln = (a.flags & EGT_EP_LAYERNORM) != 0, gates = (a.flags & EGT_EP_GATES) != 0;
  float wB[T][4];   // A operand of the dxh GEMM: row = channel 16t + p, contraction index i = 4q + s
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 16 * t + p;
      wB[t][s] = c < DE ? wcat(a, gates, c, 4 * q + s) * (ln ? a.gamma[c] : 1.0f) : 0.f;
    }
  v4f accT[T];
#pragma unroll
  for (int t = 0; t < T; ++t) accT[t] = (v4f){0.f, 0.f, 0.f, 0.f};
  float4 accS = make_float4(0.f, 0.f, 0.f, 0.f);
  const long ntiles = (a.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * EDGE_WAVES + wave; tile < ntiles; tile += (long)gridDim.x * EDGE_WAVES) {
    const long row = tile * 16 + p;
    const bool valid = row < a.rows;
    const long rc = valid ? row : a.rows - 1;
    float4 x[T], base[T];
    frag_gload<DE>(x, a.e, rc, q);
    if (a.de_out) frag_gload<DE>(base, a.de_out, rc, q);   // gradient e already carries (residual branch)
    // dpre[row][4q..4q+3]: q < 2 from dG, q >= 2 from dE (through the activation derivative)
    float4 dp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < 2) {
      if (gates) dp = *reinterpret_cast<const float4*>(a.dG + (size_t)rc * EDGE_H + 4 * q);
    } else {
      dp = *reinterpret_cast<const float4*>(a.dE + (size_t)rc * EDGE_H + 4 * (q - 2));
      if (a.act != EGT_ACT_NONE) {
        const float4 y = *reinterpret_cast<const float4*>(a.E_out + (size_t)rc * EDGE_H + 4 * (q - 2));
        dp.x *= act_grad_from_out(a.act, a.act_alpha, y.x); dp.y *= act_grad_from_out(a.act, a.act_alpha, y.y);
        dp.z *= act_grad_from_out(a.act, a.act_alpha, y.z); dp.w *= act_grad_from_out(a.act, a.act_alpha, y.w);
      }
    }
    if (!valid) dp = make_float4(0.f, 0.f, 0.f, 0.f);
    const float rstd = ln_frags<DE>(x, q, a.ln_eps, ln);
    float4 dx[T];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = MFMA(wB[t][0], dp.x, acc);
      acc = MFMA(wB[t][1], dp.y, acc);
      acc = MFMA(wB[t][2], dp.z, acc);
      acc = MFMA(wB[t][3], dp.w, acc);
      dx[t] = make_float4(acc[0], acc[1], acc[2], acc[3]);   // channels 16t + 4q + r of row p (zero past DE)
      m1 += (acc[0] + acc[1]) + (acc[2] + acc[3]);
      m2 = fmaf(acc[0], x[t].x, m2); m2 = fmaf(acc[1], x[t].y, m2);
      m2 = fmaf(acc[2], x[t].z, m2); m2 = fmaf(acc[3], x[t].w, m2);
    }
    if (ln) {
      m1 = sum_over_q(m1) * (1.0f / DE);
      m2 = sum_over_q(m2) * (1.0f / DE);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        dx[t].x = rstd * (dx[t].x - m1 - x[t].x * m2); dx[t].y = rstd * (dx[t].y - m1 - x[t].y * m2);
        dx[t].z = rstd * (dx[t].z - m1 - x[t].z * m2); dx[t].w = rstd * (dx[t].w - m1 - x[t].w * m2);
      }
    }
    if (a.de_out) {
#pragma unroll
      for (int t = 0; t < T; ++t) { dx[t].x += base[t].x; dx[t].y += base[t].y; dx[t].z += base[t].z; dx[t].w += base[t].w; }
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (16 * t + 4 * q < DE) *reinterpret_cast<float4*>(a.d_e + (size_t)row * DE + 16 * t + 4 * q) = dx[t];
    }
    // transposed operands of the weight-gradient contraction
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<float4*>(xs + p * LD + 16 * t + 4 * q) = x[t];
    *reinterpret_cast<float4*>(ds + p * 16 + 4 * q) = dp;
    lds_sync();
    float aD[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aD[s] = ds[(4 * s + q) * 16 + p];        // A[i = p][row = 4s + q]
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        accT[t] = MFMA(aD[s], xs[(4 * s + q) * LD + 16 * t + p], accT[t]);   // B[row = 4s + q][c = 16t + p]
    accS.x += dp.x; accS.y += dp.y; accS.z += dp.z; accS.w += dp.w;
    lds_sync();
  }
  // accT[t][r] = That[c = 16t + p][i = 4q + r]; s[4q + r] = sum over the 16 lanes p
  __syncthreads();
  float* red = smem + wave * PSZ;   // the tiles are dead: reuse the space, one image per wave
#pragma unroll
  for (int t = 0; t < T; ++t)
    if (16 * t + p < DE)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(16 * t + p) * 16 + 4 * q + r] = accT[t][r];
  const float s0 = row_sum16(accS.x), s1 = row_sum16(accS.y), s2 = row_sum16(accS.z), s3 = row_sum16(accS.w);
  if (p == 0) { red[DE * 16 + 4 * q] = s0; red[DE * 16 + 4 * q + 1] = s1; red[DE * 16 + 4 * q + 2] = s2; red[DE * 16 + 4 * q + 3] = s3; }
  __syncthreads();
  float* part = a.ws + (size_t)blockIdx.x * PSZ;
  for (int i = threadIdx.x; i < PSZ; i += 64 * EDGE_WAVES) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < EDGE_WAVES; ++w) v += smem[w * PSZ + i];
    part[i] = v;
  }
}

template <int DE>
__global__ void __launch_bounds__(256) k_edge_proj_bwd_final(EdgeArgs a) {
  __shared__ float T[DE * 16 + 16];
  const bool ln = (a.flags & EGT_EP_LAYERNORM) != 0;
  const bool gates = (a.flags & EGT_EP_GATES) != 0;
  constexpr int PSZ = DE * 16 + 16;
  for (int i = threadIdx.x; i < PSZ; i += 256) {
    float s = 0.f;
    for (int p = 0; p < a.n_partials; ++p) s += a.ws[(size_t)p * PSZ + i];
    T[i] = s;
  }
  __syncthreads();
  const float* s = T + DE * 16;
  for (int i = threadIdx.x; i < DE * 16; i += 256) {
    const int k = i >> 4, j = i & 15;
    float v = T[i];
    if (ln) v = a.gamma[k] * v + a.beta[k] * s[j];
    if (j < 8) { if (gates) a.d_Wg[k * EDGE_H + j] = v; }
    else a.d_We[k * EDGE_H + (j - 8)] = v;
  }
  if (threadIdx.x < 16) {
    const int j = threadIdx.x;
    if (j < 8) { if (gates) a.d_bg[j] = s[j]; }
    else a.d_be[j - 8] = s[j];
  }
  if (ln && threadIdx.x < DE) {
    const int k = threadIdx.x;
    float dg = 0.f, db = 0.f;
    for (int j = 0; j < 16; ++j) {
      if (j < 8 && !gates) continue;
      const float w = (j < 8) ? a.Wg[k * EDGE_H + j] : a.We[k * EDGE_H + (j - 8)];
      dg = fmaf(w, T[k * 16 + j], dg);
      db = fmaf(w, s[j], db);
    }
    a.d_gamma[k] = dg;
    a.d_beta[k] = db;
  }
}

// ----------------------------------------------------------- update forward ---
template <int DE>
__global__ void __launch_bounds__(256) k_edge_update_fwd(EdgeArgs a) {
  __shared__ __attribute__((aligned(16))) float w[EDGE_H * DE + DE];
  for (int i = threadIdx.x; i < EDGE_H * DE; i += 256) w[i] = a.Wr[i];
  for (int i = threadIdx.x; i < DE; i += 256) w[EDGE_H * DE + i] = a.br[i];
  __syncthreads();
  constexpr int F4 = DE / 4;
  const size_t total = (size_t)a.rows * F4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / F4;
    const int c = (int)(i % F4) * 4;
    const float4 ev = EDGE_NT_IN ? egt_ld4_nt(a.e + row * DE + c) : *reinterpret_cast<const float4*>(a.e + row * DE + c);
    const float4* hp = reinterpret_cast<const float4*>(a.h_hat + row * EDGE_H);
    const float4 h0 = hp[0], h1 = hp[1];
    const float hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    float4 acc = *reinterpret_cast<const float4*>(w + EDGE_H * DE + c);
#pragma unroll
    for (int h = 0; h < EDGE_H; ++h) {
      const float4 wv = *reinterpret_cast<const float4*>(w + h * DE + c);
      acc.x = fmaf(hh[h], wv.x, acc.x);
      acc.y = fmaf(hh[h], wv.y, acc.y);
      acc.z = fmaf(hh[h], wv.z, acc.z);
      acc.w = fmaf(hh[h], wv.w, acc.w);
    }
    // dense_edge_r output, then res_edge: y + e  (graph_xformer_model_base.py:214,218)
    *reinterpret_cast<float4*>(a.e_o + row * DE + c) =
        make_float4(acc.x + ev.x, acc.y + ev.y, acc.z + ev.z, acc.w + ev.w);
  }
}

// ---------------------------------------------------------- update backward ---
// partial layout per workgroup: dWr[8][DE] then dbr[DE]
//   d_h_hat[h][row] = sum_c Wr[h][c] de'[row][c]             (A = Wr rows, B = de' fragments in place)
//   dWr[h][c]      += sum_row h_hat[row][h] de'[row][c]      (A = h_hat straight from global, rows on
//                                                            the contraction axis; B = de' through LDS)
template <int DE>
__global__ void __launch_bounds__(64 * EDGE_WAVES) k_edge_update_bwd(EdgeArgs a) {
  constexpr int T = EdgeGeo<DE>::T, LD = EdgeGeo<DE>::LD, PSZ = EDGE_H * DE + DE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  float* xs = smem + wave * (16 * LD);
  float wR[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * q + r;
      wR[t][r] = (p < EDGE_H && c < DE) ? a.Wr[p * DE + c] : 0.f;
    }
  v4f accW[T];
  float4 accB[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { accW[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accB[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
  const long ntiles = (a.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * EDGE_WAVES + wave; tile < ntiles; tile += (long)gridDim.x * EDGE_WAVES) {
    const long row0 = tile * 16, row = row0 + p;
    const bool valid = row < a.rows;
    float4 df[T];
    frag_gload<DE>(df, a.de_out, valid ? row : a.rows - 1, q);
    float hA[4];   // A[h = p][row = 4s + q]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long rr = row0 + 4 * s + q;
      const bool ok = p < EDGE_H && rr < a.rows;
      const float v = a.h_hat[(size_t)(rr < a.rows ? rr : a.rows - 1) * EDGE_H + (p & 7)];
      hA[s] = ok ? v : 0.f;
    }
    if (!valid) {
#pragma unroll
      for (int t = 0; t < T; ++t) df[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc = MFMA(wR[t][0], df[t].x, acc);
      acc = MFMA(wR[t][1], df[t].y, acc);
      acc = MFMA(wR[t][2], df[t].z, acc);
      acc = MFMA(wR[t][3], df[t].w, acc);
    }
    if (valid && q < 2)
      *reinterpret_cast<float4*>(a.d_h_hat + (size_t)row * EDGE_H + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<float4*>(xs + p * LD + 16 * t + 4 * q) = df[t];
    lds_sync();
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        accW[t] = MFMA(hA[s], xs[(4 * s + q) * LD + 16 * t + p], accW[t]);   // B[row = 4s + q][c = 16t + p]
      accB[t].x += df[t].x; accB[t].y += df[t].y; accB[t].z += df[t].z; accB[t].w += df[t].w;
    }
    lds_sync();
  }
  // accW[t][r] = dWr[h = 4q + r][c = 16t + p] (q < 2); dbr[16t + 4q + r] = sum over the 16 lanes p
  __syncthreads();
  float* red = smem + wave * PSZ;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (q < 2 && 16 * t + p < DE)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(4 * q + r) * DE + 16 * t + p] = accW[t][r];
    const float b0 = row_sum16(accB[t].x), b1 = row_sum16(accB[t].y), b2 = row_sum16(accB[t].z), b3 = row_sum16(accB[t].w);
    if (p == 0 && 16 * t + 4 * q < DE) {
      float* o = red + EDGE_H * DE + 16 * t + 4 * q;
      o[0] = b0; o[1] = b1; o[2] = b2; o[3] = b3;
    }
  }
  __syncthreads();
  float* part = a.ws + (size_t)blockIdx.x * PSZ;
  for (int i = threadIdx.x; i < PSZ; i += 64 * EDGE_WAVES) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < EDGE_WAVES; ++w) v += smem[w * PSZ + i];
    part[i] = v;
  }
}

template <int DE>
__global__ void __launch_bounds__(256) k_edge_update_bwd_final(EdgeArgs a) {
  constexpr int PSZ = EDGE_H * DE + DE;
  for (int i = threadIdx.x; i < PSZ; i += 256) {
    float s = 0.f;
    for (int p = 0; p < a.n_partials; ++p) s += a.ws[(size_t)p * PSZ + i];
    if (i < EDGE_H * DE) a.d_Wr[i] = s; else a.d_br[i - EDGE_H * DE] = s;
  }
}

// out[i] = sum_p part[p][i]: 16 outputs per workgroup (64-byte runs), the partial axis split over
// 16 thread groups with four independent accumulators each; fixed order -> bit-reproducible
__global__ void __launch_bounds__(256) k_edge_reduce_partials(const float* part, int n, int np, float* out) {
  __shared__ float red[16][17];
  const int oi = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + oi;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (o < n) {
    int pi = pg;
    for (; pi + 48 < np; pi += 64) {
      v0 += part[(size_t)pi * n + o];
      v1 += part[(size_t)(pi + 16) * n + o];
      v2 += part[(size_t)(pi + 32) * n + o];
      v3 += part[(size_t)(pi + 48) * n + o];
    }
    for (; pi < np; pi += 16) v0 += part[(size_t)pi * n + o];
  }
  red[pg][oi] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (pg == 0 && o < n) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][oi];
    out[o] = s;
  }
}

// ------------------------------------------------------------------ host glue --
#define EDGE_MAX_PARTIALS 512

static int check_edge(const egt_edge_desc* d) {
  if (!d) EGT_FAIL(EGT_E_NULL, "desc is NULL");
  if (d->dtype != EGT_F32) EGT_FAIL(EGT_E_DTYPE, "only EGT_F32 is supported (got %d)", d->dtype);
  if (d->H != EDGE_H) EGT_FAIL(EGT_E_SHAPE, "edge kernels are built for num_heads=8 (got %d)", d->H);
  if (d->rows <= 0) EGT_FAIL(EGT_E_SHAPE, "rows must be positive");
  switch (d->De) {
    case 8: case 16: case 32: case 48: case 64: break;
    default: EGT_FAIL(EGT_E_SHAPE, "edge_width must be one of 8,16,32,48,64 (got %d)", d->De);
  }
  return EGT_OK;
}

// workgroups for a kernel whose workgroup walks `waves` 16-row tiles at a time
static int edge_grid(const egt_edge_desc* d, int waves, int cap) {
  const long ntiles = (d->rows + 15) / 16;
  const long wgs = (ntiles + waves - 1) / waves;
  return (int)(wgs < cap ? wgs : cap);
}
template <int DE> static size_t edge_bwd_lds(int tile_floats, int psz) {
  const size_t tiles = (size_t)EDGE_WAVES * tile_floats, red = (size_t)EDGE_WAVES * psz;
  return (tiles > red ? tiles : red) * sizeof(float);
}

#define DISPATCH_DE(De, CALL)                 \
  switch (De) {                               \
    case 8: { constexpr int DE = 8; CALL; } break;   \
    case 16: { constexpr int DE = 16; CALL; } break; \
    case 32: { constexpr int DE = 32; CALL; } break; \
    case 48: { constexpr int DE = 48; CALL; } break; \
    default: { constexpr int DE = 64; CALL; } break; \
  }

static void fill_edge(const egt_edge_desc* d, EdgeArgs& a) {
  a = EdgeArgs{};
  a.rows = d->rows; a.De = d->De; a.flags = d->flags; a.act = d->act;
  a.act_alpha = d->act_alpha; a.ln_eps = d->ln_eps;
}

extern "C" int egt_edge_proj_fwd(const egt_edge_desc* desc, const void* e, const void* ln_gamma,
                                 const void* ln_beta, const void* Wg, const void* bg,
                                 const void* We, const void* be, void* G_out, void* E_out,
                                 void* stream) {
  int rc = check_edge(desc);
  if (rc) return rc;
  if (!e || !We || !be || !E_out) EGT_FAIL(EGT_E_NULL, "e/We/be/E_out is NULL");
  if ((desc->flags & EGT_EP_LAYERNORM) && (!ln_gamma || !ln_beta)) EGT_FAIL(EGT_E_NULL, "LN params NULL");
  if ((desc->flags & EGT_EP_GATES) && (!Wg || !bg || !G_out)) EGT_FAIL(EGT_E_NULL, "gate params NULL");
  EdgeArgs a; fill_edge(desc, a);
  a.e = (const float*)e; a.gamma = (const float*)ln_gamma; a.beta = (const float*)ln_beta;
  a.Wg = (const float*)Wg; a.bg = (const float*)bg; a.We = (const float*)We; a.be = (const float*)be;
  a.G_out = (float*)G_out; a.E_o = (float*)E_out;
  const int grid = edge_grid(desc, 4, 4096);
  DISPATCH_DE(desc->De, {
    EGT_LAUNCH("k_edge_proj_fwd", k_edge_proj_fwd<DE>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_proj_fwd");
  return EGT_OK;
}

extern "C" size_t egt_edge_proj_bwd_workspace_bytes(const egt_edge_desc* d) {
  if (!d) return 0;
  return (size_t)(EDGE_MAX_PARTIALS + 1) * (d->De * 16 + 16) * sizeof(float);
}

static int edge_proj_bwd_impl(const egt_edge_desc* desc, const void* e, const void* ln_gamma,
                              const void* ln_beta, const void* Wg, const void* We,
                              const void* E_out, const void* d_G, const void* d_E, const void* d_e_base,
                              void* d_e, void* d_ln_gamma, void* d_ln_beta, void* d_Wg, void* d_bg,
                              void* d_We, void* d_be, void* workspace, void* stream) {
  int rc = check_edge(desc);
  if (rc) return rc;
  if (!e || !We || !d_E || !d_e || !d_We || !d_be || !workspace)
    EGT_FAIL(EGT_E_NULL, "e/We/d_E/d_e/d_We/d_be/workspace is NULL");
  if ((desc->flags & EGT_EP_LAYERNORM) && (!ln_gamma || !ln_beta || !d_ln_gamma || !d_ln_beta))
    EGT_FAIL(EGT_E_NULL, "LN params/grads NULL");
  if ((desc->flags & EGT_EP_GATES) && (!Wg || !d_G || !d_Wg || !d_bg)) EGT_FAIL(EGT_E_NULL, "gate params/grads NULL");
  if (desc->act != EGT_ACT_NONE && !E_out) EGT_FAIL(EGT_E_NULL, "E_out needed for the activation grad");
  EdgeArgs a; fill_edge(desc, a);
  a.e = (const float*)e; a.gamma = (const float*)ln_gamma; a.beta = (const float*)ln_beta;
  a.Wg = (const float*)Wg; a.We = (const float*)We; a.E_out = (const float*)E_out;
  a.dG = (const float*)d_G; a.dE = (const float*)d_E; a.d_e = (float*)d_e; a.de_out = (const float*)d_e_base;
  a.d_gamma = (float*)d_ln_gamma; a.d_beta = (float*)d_ln_beta; a.d_Wg = (float*)d_Wg;
  a.d_bg = (float*)d_bg; a.d_We = (float*)d_We; a.d_be = (float*)d_be; a.ws = (float*)workspace;
  const int grid = edge_grid(desc, EDGE_WAVES, EDGE_MAX_PARTIALS);
  a.n_partials = grid;
  DISPATCH_DE(desc->De, {
    constexpr int PSZ = DE * 16 + 16;
    const size_t lds = edge_bwd_lds<DE>(16 * EdgeGeo<DE>::LD + 256, PSZ);
    EGT_MAX_LDS_ONCE(k_edge_proj_bwd<DE>);
    EGT_LAUNCH("k_edge_proj_bwd", k_edge_proj_bwd<DE>, dim3(grid), dim3(64 * EDGE_WAVES), lds, (hipStream_t)stream, a);
    float* red = a.ws + (size_t)EDGE_MAX_PARTIALS * PSZ;
    EGT_LAUNCH("k_edge_reduce_partials", k_edge_reduce_partials, dim3((PSZ + 15) / 16), dim3(256), 0,
               (hipStream_t)stream, (const float*)a.ws, PSZ, grid, red);
    a.ws = red; a.n_partials = 1;
    EGT_LAUNCH("k_edge_proj_bwd_final", k_edge_proj_bwd_final<DE>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_proj_bwd");
  return EGT_OK;
}

extern "C" int egt_edge_proj_bwd(const egt_edge_desc* desc, const void* e, const void* ln_gamma,
                                 const void* ln_beta, const void* Wg, const void* We,
                                 const void* E_out, const void* d_G, const void* d_E, void* d_e,
                                 void* d_ln_gamma, void* d_ln_beta, void* d_Wg, void* d_bg,
                                 void* d_We, void* d_be, void* workspace, void* stream) {
  return edge_proj_bwd_impl(desc, e, ln_gamma, ln_beta, Wg, We, E_out, d_G, d_E, nullptr, d_e, d_ln_gamma,
                            d_ln_beta, d_Wg, d_bg, d_We, d_be, workspace, stream);
}

extern "C" int egt_edge_proj_bwd_acc(const egt_edge_desc* desc, const void* e, const void* ln_gamma,
                                     const void* ln_beta, const void* Wg, const void* We,
                                     const void* E_out, const void* d_G, const void* d_E,
                                     const void* d_e_base, void* d_e, void* d_ln_gamma, void* d_ln_beta,
                                     void* d_Wg, void* d_bg, void* d_We, void* d_be, void* workspace,
                                     void* stream) {
  return edge_proj_bwd_impl(desc, e, ln_gamma, ln_beta, Wg, We, E_out, d_G, d_E, d_e_base, d_e, d_ln_gamma,
                            d_ln_beta, d_Wg, d_bg, d_We, d_be, workspace, stream);
}

extern "C" int egt_edge_update_fwd(const egt_edge_desc* desc, const void* e, const void* h_hat,
                                   const void* Wr, const void* br, void* e_out, void* stream) {
  int rc = check_edge(desc);
  if (rc) return rc;
  if (!e || !h_hat || !Wr || !br || !e_out) EGT_FAIL(EGT_E_NULL, "e/h_hat/Wr/br/e_out is NULL");
  EdgeArgs a; fill_edge(desc, a);
  a.e = (const float*)e; a.h_hat = (const float*)h_hat; a.Wr = (const float*)Wr;
  a.br = (const float*)br; a.e_o = (float*)e_out;
  const size_t total = (size_t)desc->rows * (desc->De / 4);
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  DISPATCH_DE(desc->De, {
    EGT_LAUNCH("k_edge_update_fwd", k_edge_update_fwd<DE>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_update_fwd");
  return EGT_OK;
}

// ---- internal (egt_common.h): finish the edge-parameter gradients from per-workgroup partials in the layouts of k_edge_proj_bwd
// ([That[De][16] | s[16]]) and k_edge_update_bwd ([dWr[8][De] | dbr[De]]).  Used by the fused pair operator (egt_pair.h), whose
// backward accumulates the same partials inside its pair kernel.  Two launches: both partial arrays reduced by one grid (same
// fixed summation order as k_edge_reduce_partials), then both final stages by one grid of two workgroups.
__global__ void __launch_bounds__(256) k_pair_reduce_partials(const float* part1, int n1, const float* part2, int n2, int np, float* out1, float* out2) {
  __shared__ float red[16][17];
  const int nb1 = (n1 + 15) / 16;
  const bool first = (int)blockIdx.x < nb1;
  const float* part = first ? part1 : part2;
  float* out = first ? out1 : out2;
  const int n = first ? n1 : n2, blk = first ? blockIdx.x : blockIdx.x - nb1;
  const int oi = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int o = blk * 16 + oi;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (o < n) {
    int pi = pg;
    for (; pi + 48 < np; pi += 64) {
      v0 += part[(size_t)pi * n + o];
      v1 += part[(size_t)(pi + 16) * n + o];
      v2 += part[(size_t)(pi + 32) * n + o];
      v3 += part[(size_t)(pi + 48) * n + o];
    }
    for (; pi < np; pi += 16) v0 += part[(size_t)pi * n + o];
  }
  red[pg][oi] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (pg == 0 && o < n) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][oi];
    out[o] = s;
  }
}

// workgroup 0: the projection side (k_edge_proj_bwd_final with LayerNorm and gates), workgroup 1: dense_edge_r (k_edge_update_bwd_final)
template <int DE>
__global__ void __launch_bounds__(256) k_pair_param_final(EdgeArgs a, const float* red_proj, const float* red_upd) {
  if (blockIdx.x == 1) {
    constexpr int PSZ = EDGE_H * DE + DE;
    for (int i = threadIdx.x; i < PSZ; i += 256) {
      const float s = red_upd[i];
      if (i < EDGE_H * DE) a.d_Wr[i] = s; else a.d_br[i - EDGE_H * DE] = s;
    }
    return;
  }
  const float* T = red_proj;
  const float* s = T + DE * 16;
  for (int i = threadIdx.x; i < DE * 16; i += 256) {
    const int k = i >> 4, j = i & 15;
    const float v = a.gamma[k] * T[i] + a.beta[k] * s[j];
    if (j < 8) a.d_Wg[k * EDGE_H + j] = v; else a.d_We[k * EDGE_H + (j - 8)] = v;
  }
  if (threadIdx.x < 16) {
    const int j = threadIdx.x;
    if (j < 8) a.d_bg[j] = s[j]; else a.d_be[j - 8] = s[j];
  }
  if (threadIdx.x < DE) {
    const int k = threadIdx.x;
    float dg = 0.f, db = 0.f;
    for (int j = 0; j < 16; ++j) {
      const float w = (j < 8) ? a.Wg[k * EDGE_H + j] : a.We[k * EDGE_H + (j - 8)];
      dg = fmaf(w, T[k * 16 + j], dg);
      db = fmaf(w, s[j], db);
    }
    a.d_gamma[k] = dg;
    a.d_beta[k] = db;
  }
}

void egt_edge_finish_param_grads(int De, const float* gamma, const float* beta, const float* Wg, const float* We,
                                 const float* part_proj, const float* part_upd, int n_partials, float* red_proj, float* red_upd,
                                 float* d_gamma, float* d_beta, float* d_Wg, float* d_bg, float* d_We, float* d_be,
                                 float* d_Wr, float* d_br, hipStream_t st) {
  EdgeArgs a{};
  a.De = De; a.flags = EGT_EP_LAYERNORM | EGT_EP_GATES;
  a.gamma = gamma; a.beta = beta; a.Wg = Wg; a.We = We;
  a.d_gamma = d_gamma; a.d_beta = d_beta; a.d_Wg = d_Wg; a.d_bg = d_bg; a.d_We = d_We; a.d_be = d_be; a.d_Wr = d_Wr; a.d_br = d_br;
  DISPATCH_DE(De, {
    constexpr int PSZ1 = DE * 16 + 16;
    constexpr int PSZ2 = EDGE_H * DE + DE;
    EGT_LAUNCH("k_pair_reduce_partials", k_pair_reduce_partials, dim3((PSZ1 + 15) / 16 + (PSZ2 + 15) / 16), dim3(256), 0, st,
               part_proj, PSZ1, part_upd, PSZ2, n_partials, red_proj, red_upd);
    EGT_LAUNCH("k_pair_param_final", k_pair_param_final<DE>, dim3(2), dim3(256), 0, st, a, (const float*)red_proj, (const float*)red_upd);
  });
}

extern "C" size_t egt_edge_update_bwd_workspace_bytes(const egt_edge_desc* d) {
  if (!d) return 0;
  return (size_t)(EDGE_MAX_PARTIALS + 1) * (EDGE_H * d->De + d->De) * sizeof(float);
}

extern "C" int egt_edge_update_bwd(const egt_edge_desc* desc, const void* d_e_out,
                                   const void* h_hat, const void* Wr, void* d_h_hat, void* d_Wr,
                                   void* d_br, void* workspace, void* stream) {
  int rc = check_edge(desc);
  if (rc) return rc;
  if (!d_e_out || !h_hat || !Wr || !d_h_hat || !d_Wr || !d_br || !workspace)
    EGT_FAIL(EGT_E_NULL, "d_e_out/h_hat/Wr/d_h_hat/d_Wr/d_br/workspace is NULL");
  EdgeArgs a; fill_edge(desc, a);
  a.de_out = (const float*)d_e_out; a.h_hat = (const float*)h_hat; a.Wr = (const float*)Wr;
  a.d_h_hat = (float*)d_h_hat; a.d_Wr = (float*)d_Wr; a.d_br = (float*)d_br; a.ws = (float*)workspace;
  const int grid = edge_grid(desc, EDGE_WAVES, EDGE_MAX_PARTIALS);
  a.n_partials = grid;
  DISPATCH_DE(desc->De, {
    constexpr int PSZ = EDGE_H * DE + DE;
    const size_t lds = edge_bwd_lds<DE>(16 * EdgeGeo<DE>::LD, PSZ);
    EGT_MAX_LDS_ONCE(k_edge_update_bwd<DE>);
    EGT_LAUNCH("k_edge_update_bwd", k_edge_update_bwd<DE>, dim3(grid), dim3(64 * EDGE_WAVES), lds, (hipStream_t)stream, a);
    float* red = a.ws + (size_t)EDGE_MAX_PARTIALS * PSZ;
    EGT_LAUNCH("k_edge_reduce_partials", k_edge_reduce_partials, dim3((PSZ + 15) / 16), dim3(256), 0,
               (hipStream_t)stream, (const float*)a.ws, PSZ, grid, red);
    a.ws = red; a.n_partials = 1;
    EGT_LAUNCH("k_edge_update_bwd_final", k_edge_update_bwd_final<DE>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_update_bwd");
  return EGT_OK;
}
