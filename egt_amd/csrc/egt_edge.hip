// Edge-channel projections around the inner op (composed block path):
//   egt_edge_proj_*   : [norm_edge] -> attention_gates (De->H) , dense_edge_b (De->H)
//                       graph_xformer_model_base.py:195,201-204,149-162
//   egt_edge_update_* : e' = e + H_hat·Wr + br   (dense_edge_r + res_edge, :214-218)
// Tile machinery: a 256-thread workgroup stages 256 edge rows (256 x De fp32,
// coalesced 16-byte loads) in LDS with a +1 padded stride, then one thread owns
// one row (bank-conflict-free column walk) with the weights broadcast from
// scalar loads.  Weight gradients are per-workgroup register partials reduced by
// a deterministic second kernel (no float atomics).
#include "egt_common.h"

#define EDGE_H 8
#define TILE_ROWS 256

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

template <int DE>
__device__ __forceinline__ void load_tile(float* tile, const float* src, long row0, long rows) {
  // 256 rows x DE floats, coalesced float4 loads -> LDS stride DE+1
  constexpr int F4_PER_ROW = DE / 4;
  constexpr int ITERS = F4_PER_ROW;  // 256*DE/4 float4s over 256 threads
#pragma unroll 4
  for (int i = 0; i < ITERS; ++i) {
    const int f = i * TILE_ROWS + threadIdx.x;
    const int r = f / F4_PER_ROW, c = (f % F4_PER_ROW) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < rows) v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * DE + c);
    float* t = tile + r * (DE + 1) + c;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
}

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

// ------------------------------------------------------------- proj forward ---
template <int DE>
__global__ void __launch_bounds__(256) k_edge_proj_fwd(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const long ntiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * TILE_ROWS;
    __syncthreads();
    load_tile<DE>(tile, a.e, row0, a.rows);
    __syncthreads();
    const long row = row0 + threadIdx.x;
    float x[DE];
    const float* tr = tile + threadIdx.x * (DE + 1);
#pragma unroll
    for (int k = 0; k < DE; ++k) x[k] = tr[k];
    if (a.flags & EGT_EP_LAYERNORM) {
      float mu = 0.f;
#pragma unroll
      for (int k = 0; k < DE; ++k) mu += x[k];
      mu *= (1.0f / DE);
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < DE; ++k) { const float c = x[k] - mu; var = fmaf(c, c, var); }
      var *= (1.0f / DE);
      const float rstd = rsqrtf(var + a.ln_eps);
#pragma unroll
      for (int k = 0; k < DE; ++k) x[k] = fmaf((x[k] - mu) * rstd, a.gamma[k], a.beta[k]);
    }
    float g[EDGE_H], eb[EDGE_H];
#pragma unroll
    for (int j = 0; j < EDGE_H; ++j) { g[j] = (a.flags & EGT_EP_GATES) ? a.bg[j] : 0.f; eb[j] = a.be[j]; }
    if (a.flags & EGT_EP_GATES) {
#pragma unroll
      for (int k = 0; k < DE; ++k)
#pragma unroll
        for (int j = 0; j < EDGE_H; ++j) g[j] = fmaf(x[k], a.Wg[k * EDGE_H + j], g[j]);
    }
#pragma unroll
    for (int k = 0; k < DE; ++k)
#pragma unroll
      for (int j = 0; j < EDGE_H; ++j) eb[j] = fmaf(x[k], a.We[k * EDGE_H + j], eb[j]);
    if (row < a.rows) {
      if (a.flags & EGT_EP_GATES) {
        float4* go = reinterpret_cast<float4*>(a.G_out + (size_t)row * EDGE_H);
        go[0] = make_float4(g[0], g[1], g[2], g[3]);
        go[1] = make_float4(g[4], g[5], g[6], g[7]);
      }
#pragma unroll
      for (int j = 0; j < EDGE_H; ++j) eb[j] = act_fwd(a.act, a.act_alpha, eb[j]);
      float4* eo = reinterpret_cast<float4*>(a.E_o + (size_t)row * EDGE_H);
      eo[0] = make_float4(eb[0], eb[1], eb[2], eb[3]);
      eo[1] = make_float4(eb[4], eb[5], eb[6], eb[7]);
    }
  }
}

// ------------------------------------------------------------ proj backward ---
// partial layout per workgroup: That[DE][16] then s[16]
template <int DE>
__global__ void __launch_bounds__(256) k_edge_proj_bwd(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                              // xhat [256][DE+1]
  float* dt = smem + TILE_ROWS * (DE + 1);         // dpre [256][17]
  const bool ln = (a.flags & EGT_EP_LAYERNORM) != 0;
  const bool gates = (a.flags & EGT_EP_GATES) != 0;
  const int kq = threadIdx.x >> 2, jq = threadIdx.x & 3;
  float accT[4] = {0.f, 0.f, 0.f, 0.f}, accS[4] = {0.f, 0.f, 0.f, 0.f};
  const long ntiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * TILE_ROWS;
    __syncthreads();
    load_tile<DE>(tile, a.e, row0, a.rows);
    __syncthreads();
    const long row = row0 + threadIdx.x;
    const bool valid = row < a.rows;
    float x[DE];
    float* tr = tile + threadIdx.x * (DE + 1);
#pragma unroll
    for (int k = 0; k < DE; ++k) x[k] = tr[k];
    float rstd = 1.0f;
    if (ln) {
      float mu = 0.f;
#pragma unroll
      for (int k = 0; k < DE; ++k) mu += x[k];
      mu *= (1.0f / DE);
      float var = 0.f;
#pragma unroll
      for (int k = 0; k < DE; ++k) { const float c = x[k] - mu; var = fmaf(c, c, var); }
      var *= (1.0f / DE);
      rstd = rsqrtf(var + a.ln_eps);
#pragma unroll
      for (int k = 0; k < DE; ++k) { x[k] = (x[k] - mu) * rstd; tr[k] = x[k]; }
    }
    float dp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) dp[j] = 0.f;
    if (valid) {
      if (gates) {
        const float4* gp = reinterpret_cast<const float4*>(a.dG + (size_t)row * EDGE_H);
        float4 u = gp[0], v = gp[1];
        dp[0] = u.x; dp[1] = u.y; dp[2] = u.z; dp[3] = u.w;
        dp[4] = v.x; dp[5] = v.y; dp[6] = v.z; dp[7] = v.w;
      }
      const float4* ep = reinterpret_cast<const float4*>(a.dE + (size_t)row * EDGE_H);
      float4 u = ep[0], v = ep[1];
      dp[8] = u.x; dp[9] = u.y; dp[10] = u.z; dp[11] = u.w;
      dp[12] = v.x; dp[13] = v.y; dp[14] = v.z; dp[15] = v.w;
      if (a.act != EGT_ACT_NONE) {
        const float* yo = a.E_out + (size_t)row * EDGE_H;
#pragma unroll
        for (int j = 0; j < EDGE_H; ++j) dp[8 + j] *= act_grad_from_out(a.act, a.act_alpha, yo[j]);
      }
    }
    float* dr = dt + threadIdx.x * 17;
#pragma unroll
    for (int j = 0; j < 16; ++j) dr[j] = dp[j];
    // d(e_ln)_k = sum_j dpre_j * Wcat[k][j]
    float dx[DE];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      float v = 0.f;
      if (gates) {
#pragma unroll
        for (int j = 0; j < EDGE_H; ++j) v = fmaf(dp[j], a.Wg[k * EDGE_H + j], v);
      }
#pragma unroll
      for (int j = 0; j < EDGE_H; ++j) v = fmaf(dp[8 + j], a.We[k * EDGE_H + j], v);
      if (ln) {
        v *= a.gamma[k];
        m1 += v;
        m2 = fmaf(v, x[k], m2);
      }
      dx[k] = v;
    }
    if (ln) {
      m1 *= (1.0f / DE);
      m2 *= (1.0f / DE);
#pragma unroll
      for (int k = 0; k < DE; ++k) dx[k] = rstd * (dx[k] - m1 - x[k] * m2);
    }
    if (valid) {
      float4* o = reinterpret_cast<float4*>(a.d_e + (size_t)row * DE);
#pragma unroll
      for (int k = 0; k < DE; k += 4) o[k / 4] = make_float4(dx[k], dx[k + 1], dx[k + 2], dx[k + 3]);
    }
    __syncthreads();
    // That[k][j] += sum_r xhat[r][k] * dpre[r][j]
    if (kq < DE) {
      for (int r = 0; r < TILE_ROWS; ++r) {
        const float xv = tile[r * (DE + 1) + kq];
        const float* d4 = dt + r * 17 + jq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) accT[i] = fmaf(xv, d4[i], accT[i]);
        if (kq == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) accS[i] += d4[i];
        }
      }
    }
  }
  float* part = a.ws + (size_t)blockIdx.x * (DE * 16 + 16);
  if (kq < DE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) part[kq * 16 + jq * 4 + i] = accT[i];
    if (kq == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[DE * 16 + jq * 4 + i] = accS[i];
    }
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
    const float4 ev = *reinterpret_cast<const float4*>(a.e + row * DE + c);
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
template <int DE>
__global__ void __launch_bounds__(256) k_edge_update_bwd(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                          // d_e_out [256][DE+1]
  float* ht = smem + TILE_ROWS * (DE + 1);     // h_hat   [256][9]
  const int c = threadIdx.x & 63, hq = threadIdx.x >> 6;
  float accW0 = 0.f, accW1 = 0.f, accB = 0.f;
  const long ntiles = (a.rows + TILE_ROWS - 1) / TILE_ROWS;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = t * TILE_ROWS;
    __syncthreads();
    load_tile<DE>(tile, a.de_out, row0, a.rows);
    const long row = row0 + threadIdx.x;
    {
      float* hr = ht + threadIdx.x * 9;
      if (row < a.rows) {
        const float4* hp = reinterpret_cast<const float4*>(a.h_hat + (size_t)row * EDGE_H);
        const float4 u = hp[0], v = hp[1];
        hr[0] = u.x; hr[1] = u.y; hr[2] = u.z; hr[3] = u.w;
        hr[4] = v.x; hr[5] = v.y; hr[6] = v.z; hr[7] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) hr[j] = 0.f;
      }
    }
    __syncthreads();
    // d_h_hat[row][h] = sum_c de'[row][c] * Wr[h][c]
    float dh[EDGE_H];
#pragma unroll
    for (int h = 0; h < EDGE_H; ++h) dh[h] = 0.f;
    const float* tr = tile + threadIdx.x * (DE + 1);
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      const float v = tr[k];
#pragma unroll
      for (int h = 0; h < EDGE_H; ++h) dh[h] = fmaf(v, a.Wr[h * DE + k], dh[h]);
    }
    if (row < a.rows) {
      float4* o = reinterpret_cast<float4*>(a.d_h_hat + (size_t)row * EDGE_H);
      o[0] = make_float4(dh[0], dh[1], dh[2], dh[3]);
      o[1] = make_float4(dh[4], dh[5], dh[6], dh[7]);
    }
    // dWr[h][c] += sum_r h_hat[r][h] * de'[r][c]
    if (c < DE) {
      for (int r = 0; r < TILE_ROWS; ++r) {
        const float dv = tile[r * (DE + 1) + c];
        accW0 = fmaf(ht[r * 9 + 2 * hq], dv, accW0);
        accW1 = fmaf(ht[r * 9 + 2 * hq + 1], dv, accW1);
        if (hq == 0) accB += dv;
      }
    }
  }
  float* part = a.ws + (size_t)blockIdx.x * (EDGE_H * DE + DE);
  if (c < DE) {
    part[(2 * hq) * DE + c] = accW0;
    part[(2 * hq + 1) * DE + c] = accW1;
    if (hq == 0) part[EDGE_H * DE + c] = accB;
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

// out[i] = sum_p part[p][i]: 64 outputs per workgroup, the partial axis split over 4 wavefronts
__global__ void __launch_bounds__(256) k_edge_reduce_partials(const float* part, int n, int np, float* out) {
  __shared__ float red[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  float v = 0.f;
  if (o < n)
    for (int pi = pg; pi < np; pi += 4) v += part[(size_t)pi * n + o];
  red[pg][threadIdx.x & 63] = v;
  __syncthreads();
  if (pg == 0 && o < n) out[o] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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

static int edge_grid(const egt_edge_desc* d, int cap) {
  long ntiles = (d->rows + TILE_ROWS - 1) / TILE_ROWS;
  return (int)(ntiles < cap ? ntiles : cap);
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
  const int grid = edge_grid(desc, 4096);
  DISPATCH_DE(desc->De, {
    const size_t lds = (size_t)TILE_ROWS * (DE + 1) * 4;
    (void)hipFuncSetAttribute((const void*)k_edge_proj_fwd<DE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    EGT_LAUNCH("k_edge_proj_fwd", k_edge_proj_fwd<DE>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_proj_fwd");
  return EGT_OK;
}

extern "C" size_t egt_edge_proj_bwd_workspace_bytes(const egt_edge_desc* d) {
  if (!d) return 0;
  return (size_t)(EDGE_MAX_PARTIALS + 1) * (d->De * 16 + 16) * sizeof(float);
}

extern "C" int egt_edge_proj_bwd(const egt_edge_desc* desc, const void* e, const void* ln_gamma,
                                 const void* ln_beta, const void* Wg, const void* We,
                                 const void* E_out, const void* d_G, const void* d_E, void* d_e,
                                 void* d_ln_gamma, void* d_ln_beta, void* d_Wg, void* d_bg,
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
  a.dG = (const float*)d_G; a.dE = (const float*)d_E; a.d_e = (float*)d_e;
  a.d_gamma = (float*)d_ln_gamma; a.d_beta = (float*)d_ln_beta; a.d_Wg = (float*)d_Wg;
  a.d_bg = (float*)d_bg; a.d_We = (float*)d_We; a.d_be = (float*)d_be; a.ws = (float*)workspace;
  const int grid = edge_grid(desc, EDGE_MAX_PARTIALS);
  a.n_partials = grid;
  DISPATCH_DE(desc->De, {
    const size_t lds = (size_t)TILE_ROWS * (DE + 1 + 17) * 4;
    (void)hipFuncSetAttribute((const void*)k_edge_proj_bwd<DE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    EGT_LAUNCH("k_edge_proj_bwd", k_edge_proj_bwd<DE>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    constexpr int PSZ = DE * 16 + 16;
    float* red = a.ws + (size_t)EDGE_MAX_PARTIALS * PSZ;
    EGT_LAUNCH("k_edge_reduce_partials", k_edge_reduce_partials, dim3((PSZ + 63) / 64), dim3(256), 0,
               (hipStream_t)stream, (const float*)a.ws, PSZ, grid, red);
    a.ws = red; a.n_partials = 1;
    EGT_LAUNCH("k_edge_proj_bwd_final", k_edge_proj_bwd_final<DE>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_proj_bwd");
  return EGT_OK;
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
  const int grid = edge_grid(desc, EDGE_MAX_PARTIALS);
  a.n_partials = grid;
  DISPATCH_DE(desc->De, {
    const size_t lds = (size_t)TILE_ROWS * (DE + 1 + 9) * 4;
    (void)hipFuncSetAttribute((const void*)k_edge_update_bwd<DE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    EGT_LAUNCH("k_edge_update_bwd", k_edge_update_bwd<DE>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
    constexpr int PSZ = EDGE_H * DE + DE;
    float* red = a.ws + (size_t)EDGE_MAX_PARTIALS * PSZ;
    EGT_LAUNCH("k_edge_reduce_partials", k_edge_reduce_partials, dim3((PSZ + 63) / 64), dim3(256), 0,
               (hipStream_t)stream, (const float*)a.ws, PSZ, grid, red);
    a.ws = red; a.n_partials = 1;
    EGT_LAUNCH("k_edge_update_bwd_final", k_edge_update_bwd_final<DE>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  });
  EGT_HIP_LAUNCH_CHECK("egt_edge_update_bwd");
  return EGT_OK;
}
