// Node-side kernels of the fused attention block (O(N*Dh) work, once per layer):
//   k_node_pre      : norm_mha -> dense_qkv, written packed per head-pair
//                     (graph_xformer_model_base.py:109,113)         [+ edge-weight prep]
//   k_node_post     : dense_mha + res_mha (:136,140)
//   k_node_post_bwd : dV_att = dh'.Wo^T (packed), delta = sum_k dV_att*V_att,
//                     dWo/dbo partials                              [+ edge-weight prep]
//   k_node_pre_bwd  : dQKV -> d(h_ln) -> LN backward -> dh ; dWqkv/dbqkv/dgamma/dbeta partials
//   k_sum_segments  : deterministic reduction of all per-workgroup partials
//   k_edge_param_grads : T,s,R -> grads of norm_edge / attention_gates / dense_edge_b / dense_edge_r
// One workgroup per graph; every contraction is a 16x16 tile on
// v_mfma_f32_16x16x4_f32 with the activation rows staged in LDS.
#include "egt_block.h"

#define LDP 4  // LDS row padding (floats)

__device__ __forceinline__ float sum16(float v) { return row_sum16(v); }

// acc += X[16 rows][K] . W[K][16 cols]; X and W staged in LDS (strides ld / ldw, W pointer
// already at column 0 of the tile); colok: this lane's output column exists
__device__ __forceinline__ v4f mm_xw(const float* xs, int ld, const float* W, int ldw,
                                     int K, int p, int q, bool colok, v4f acc) {
#pragma unroll 4
  for (int t = 0; t < K; t += 4) {
    const float bv = colok ? W[(size_t)(t + q) * ldw + p] : 0.f;
    acc = MFMA(xs[p * ld + t + q], bv, acc);
  }
  return acc;
}
// acc += X[16 rows][K] . W^T, W = [16 out-rows][K] in LDS (row stride ldw, pointer at out-row 0)
__device__ __forceinline__ v4f mm_xwt(const float* xs, int ld, const float* W, int ldw,
                                      int K, int p, int q, bool colok, v4f acc) {
#pragma unroll 4
  for (int t = 0; t < K; t += 4) {
    const float bv = colok ? W[(size_t)p * ldw + t + q] : 0.f;
    acc = MFMA(xs[p * ld + t + q], bv, acc);
  }
  return acc;
}

// ---- edge-weight preparation (runs as one extra workgroup of the first node kernel) ----
// Wp[c][i] = gamma_c * Wsel[c][head(i)], c2[i] = sum_c beta_c * Wsel[c][head(i)] + bias
struct PrepLayer { const float *ne_g, *ne_b, *Wg, *bg, *We, *be; float* pw; };

__device__ void prep_layer(const PrepLayer& a, int De, bool gated, float* red) {
  const int DEP = ((De + 15) / 16) * 16, t = threadIdx.x;
  for (int idx = t; idx < DEP * 16; idx += blockDim.x) {
    const int c = idx >> 4, i = idx & 15;
    float v = 0.f;
    if (c < De) {
      const int hd = col_head(i);
      if (col_is_gate(i)) v = gated ? a.ne_g[c] * a.Wg[c * BH + hd] : 0.f;
      else v = a.ne_g[c] * a.We[c * BH + hd];
    }
    a.pw[idx] = v;
  }
  if (t < 256) {
    const int i = t & 15, part = t >> 4, hd = col_head(i);
    const bool isg = col_is_gate(i);
    float v = 0.f;
    if (!isg || gated) {
      const float* W = isg ? a.Wg : a.We;
      for (int c = part; c < De; c += 16) v = fmaf(a.ne_b[c], W[c * BH + hd], v);
    }
    red[part * 16 + i] = v;
  }
  __syncthreads();
  if (t < 16) {
    const int hd = col_head(t);
    const bool isg = col_is_gate(t);
    float v = 0.f;
    if (!isg || gated) {
      v = isg ? a.bg[hd] : a.be[hd];
      for (int part = 0; part < 16; ++part) v += red[part * 16 + t];
    }
    a.pw[DEP * 16 + t] = v;
  }
}
__device__ void prep_device(const BlockArgs& a, float* red) {
  const PrepLayer L{a.ne_g, a.ne_b, a.Wg, a.bg, a.We, a.be, a.pw};
  prep_layer(L, a.De, (a.flags & EGT_BF_GATE) != 0, red);
}

// the same preparation for every layer of a stack in one launch (one workgroup per layer)
#define PREP_MAX_LAYERS 64
struct PrepArgs { PrepLayer L[PREP_MAX_LAYERS]; int De; uint32_t flags; };
__global__ void __launch_bounds__(256) k_edge_prep(PrepArgs pa) {
  __shared__ float red[256];
  prep_layer(pa.L[blockIdx.x], pa.De, (pa.flags & EGT_BF_GATE) != 0, red);
}

// stage a dense [rows][width] weight matrix into LDS (row stride ldw): 16-byte loads, four in
// flight per thread before the LDS writes
__device__ __forceinline__ void stage_weight(float* ws, int ldw, const float* W, int rows, int width) {
  const int n4 = rows * width / 4;
  const int NT = blockDim.x;
  if ((reinterpret_cast<uintptr_t>(W) & 15) != 0) {   // parameter views need not be 16-byte aligned
    for (int i = threadIdx.x; i < rows * width; i += NT) ws[(i / width) * ldw + (i % width)] = W[i];
    return;
  }
  for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * NT) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT;
      v[u] = *reinterpret_cast<const float4*>(W + (size_t)(i < n4 ? i : 0) * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT;
      if (i < n4) {
        const int r = (i * 4) / width, c = (i * 4) % width;
        *reinterpret_cast<float4*>(ws + r * ldw + c) = v[u];
      }
    }
  }
}

// stage `nr` rows of a [.., width] tensor into LDS (stride ld), zero-padding up to nrp rows
__device__ __forceinline__ void stage_rows(float* xs, int ld, const float* src, int width, int nr, int nrp) {
  const int w4 = width >> 2, n4 = nrp * w4, NT = blockDim.x;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * NT) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT, r = i / w4, c4 = i % w4;
      const bool ok = i < n4 && r < nr;
      v[u] = *reinterpret_cast<const float4*>(src + (ok ? (size_t)r * width + c4 * 4 : 0));
      if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT, r = i / w4, c4 = i % w4;
      if (i < n4) *reinterpret_cast<float4*>(xs + r * ld + c4 * 4) = v[u];
    }
  }
}

// --------------------------------------------------------------- node: pre -----
__global__ void __launch_bounds__(512) k_node_pre(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  if ((int)blockIdx.x == a.B * NCH) { prep_device(a, sm); return; }
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, ld = Dh + LDP, D3 = 3 * Dh;
  const int lane = t & 63, wave = t >> 6, p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;                 // [NODE_RC][ld]
  float* ws = xs + NODE_RC * ld;  // Wqkv [Dh][ldw]
  const int ldw = D3 + LDP;
  stage_weight(ws, ldw, a.Wqkv, Dh, D3);
  const int nct = (D3 + 15) / 16;
  for (int r0 = chunk * NODE_RC; r0 < min(N, (chunk + 1) * NODE_RC); r0 += NODE_RC) {  // one chunk
    const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
    const size_t row0 = (size_t)b * N + r0;
    __syncthreads();
    stage_rows(xs, ld, a.h + row0 * Dh, Dh, nr, nrp);
    __syncthreads();
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // LayerNorm: 16 lanes per row
      float* x = xs + (rb + q) * ld;
      float v[4], s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; v[i] = c < Dh ? x[c] : 0.f; s += v[i]; }
      const float mu = sum16(s) / Dh;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) { v[i] -= mu; ss = fmaf(v[i], v[i], ss); } }
      const float rstd = rsqrtf(sum16(ss) / Dh + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) x[c] = fmaf(v[i] * rstd, a.nm_g[c], a.nm_b[c]); }
    }
    __syncthreads();
    const int ntr = nrp / 16;
    for (int tile = wave; tile < ntr * nct; tile += NW) {
      const int rt = tile / nct, ct = tile % nct, c = ct * 16 + p;
      const bool colok = c < D3;
      const float bias = colok ? a.bqkv[c] : 0.f;
      v4f acc = {bias, bias, bias, bias};
      acc = mm_xw(xs + rt * 16 * ld, ld, ws + ct * 16, ldw, Dh, p, q, colok, acc);
      if (colok) {
        const int s = c / Dh, cc = c % Dh, k = cc >> 3, hh = cc & 7;
        const int pos = s * 64 + (hh >> 1) * 16 + k * 2 + (hh & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * q + r;
          if (row < nr) a.qkvp[(row0 + row) * QKVP + pos] = acc[r];
        }
      }
    }
    if (a.DK < 8) {
      for (int i = t; i < nr * QKVP; i += blockDim.x) {
        const int pos = i % QKVP;
        if (((pos >> 1) & 7) >= a.DK) a.qkvp[row0 * QKVP + i] = 0.f;
      }
    }
  }
}

// -------------------------------------------------------------- node: post -----
__global__ void __launch_bounds__(512) k_node_post(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, ld = Dh + LDP;
  const int lane = t & 63, wave = t >> 6, p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;
  float* ws = xs + NODE_RC * ld;  // Wo [Dh][ld]
  stage_weight(ws, ld, a.Wo, Dh, Dh);
  const int nct = (Dh + 15) / 16;
  for (int r0 = chunk * NODE_RC; r0 < min(N, (chunk + 1) * NODE_RC); r0 += NODE_RC) {  // one chunk
    const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
    const size_t row0 = (size_t)b * N + r0;
    __syncthreads();
    stage_rows(xs, ld, a.v_att + row0 * Dh, Dh, nr, nrp);
    __syncthreads();
    const int ntr = nrp / 16;
    for (int tile = wave; tile < ntr * nct; tile += NW) {
      const int rt = tile / nct, ct = tile % nct, c = ct * 16 + p;
      const bool colok = c < Dh;
      const float bias = colok ? a.bo[c] : 0.f;
      v4f acc = {bias, bias, bias, bias};
      acc = mm_xw(xs + rt * 16 * ld, ld, ws + ct * 16, ld, Dh, p, q, colok, acc);
      if (colok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * q + r;
          if (row < nr) {
            const size_t o = (row0 + row) * Dh + c;
            a.h_out[o] = acc[r] + a.h[o];   // dense_mha output, then res_mha: y + h
          }
        }
      }
    }
  }
}

// ------------------------------------------------------ node: post backward -----
// node partial layout per graph: [dWqkv Dh*3Dh | dbqkv 3Dh | dgamma Dh | dbeta Dh | dWo Dh*Dh | dbo Dh]
__global__ void __launch_bounds__(512) k_node_post_bwd(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  if ((int)blockIdx.x == a.B * NCH) { prep_device(a, sm); return; }
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, ld = Dh + LDP;
  const int lane = t & 63, wave = t >> 6, p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* ds = sm;                    // dh'   [NODE_RC][ld]
  float* vs = ds + NODE_RC * ld;     // v_att [NODE_RC][ld]
  float* ws = vs + NODE_RC * ld;     // Wo    [Dh][ld]
  float* dlp = ws + Dh * ld;         // delta partials [nit][NODE_RC][8]
  stage_weight(ws, ld, a.Wo, Dh, Dh);
  const int nit = (Dh + 15) / 16;    // <= 4
  v4f accW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) accW[j] = (v4f){0.f, 0.f, 0.f, 0.f};
  float accB = 0.f;
  for (int r0 = chunk * NODE_RC; r0 < min(N, (chunk + 1) * NODE_RC); r0 += NODE_RC) {  // one chunk
    const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
    const size_t row0 = (size_t)b * N + r0;
    __syncthreads();
    stage_rows(ds, ld, a.dh_out + row0 * Dh, Dh, nr, nrp);
    stage_rows(vs, ld, a.v_att + row0 * Dh, Dh, nr, nrp);
    __syncthreads();
    // (1) dV_att = dh'.Wo^T: (row tile, i tile) pairs spread over the waves; the per-head
    //     delta contributions of each i tile go through LDS and are summed in fixed order
    for (int tile = wave; tile < (nrp / 16) * nit; tile += NW) {
      const int rt = tile / nit, it = tile % nit;
      const int i = it * 16 + p;
      const bool colok = i < Dh;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = mm_xwt(ds + rt * 16 * ld, ld, ws + it * 16 * ld, ld, Dh, p, q, colok, acc);
      const int k = i >> 3, hh = i & 7;
      const int pos = (hh >> 1) * 16 + k * 2 + (hh & 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        if (colok && row < nr) a.dvp[(row0 + row) * 64 + pos] = acc[r];
        float pr = colok ? acc[r] * vs[row * ld + i] : 0.f;
        pr += lane_xor<8>(pr);   // the tile's two k values of head p&7
        if (p < 8) dlp[(it * NODE_RC + row) * 8 + p] = pr;
      }
    }
    __syncthreads();
    for (int i = t; i < nr * 8; i += blockDim.x) {
      float dl = 0.f;
      for (int it = 0; it < nit; ++it) dl += dlp[it * NODE_RC * 8 + i];
      a.stats[(row0 * BH + i) * 4 + 2] = dl;
    }
    if (a.DK < 8) {
      for (int i = t; i < nr * 64; i += blockDim.x)
        if ((((i & 63) >> 1) & 7) >= a.DK) a.dvp[row0 * 64 + i] = 0.f;
    }
    // (2) dWo[i][c] += sum_rows v_att[row][i] * dh'[row][c]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = wave + NW * j;
      if (idx < nit * nit) {
        const int it = idx / nit, ct = idx % nit;
#pragma unroll 4
        for (int rr = 0; rr < nrp; rr += 4)
          accW[j] = MFMA(vs[(rr + q) * ld + it * 16 + p], ds[(rr + q) * ld + ct * 16 + p], accW[j]);
      }
    }
    if (t < Dh) {
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;   // independent partial sums: the LDS reads pipeline
      for (int r = 0; r < nrp; r += 4) {
        b0 += ds[r * ld + t]; b1 += ds[(r + 1) * ld + t]; b2 += ds[(r + 2) * ld + t]; b3 += ds[(r + 3) * ld + t];
      }
      accB += (b0 + b1) + (b2 + b3);
    }
  }
  float* part = a.npart + (size_t)blockIdx.x * (Dh * 3 * Dh + 3 * Dh + 2 * Dh + Dh * Dh + Dh) + Dh * 3 * Dh + 3 * Dh + 2 * Dh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = wave + NW * j;
    if (idx < nit * nit) {
      const int it = idx / nit, ct = idx % nit, c = ct * 16 + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + 4 * q + r;
        if (i < Dh && c < Dh) part[i * Dh + c] = accW[j][r];
      }
    }
  }
  if (t < Dh) part[Dh * Dh + t] = accB;
}

// ------------------------------------------------------- node: pre backward -----
__global__ void __launch_bounds__(512) k_node_pre_bwd(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, D3 = 3 * Dh;
  const int ld = Dh + LDP, ld3 = D3 + LDP;
  const int lane = t & 63, wave = t >> 6, p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;                        // xhat  [NODE_RC][ld]
  float* dqs = xs + NODE_RC * ld;        // dQKV  [NODE_RC][ld3]
  float* dls = dqs + NODE_RC * ld3;      // d h_ln [NODE_RC][ld]
  float* rs = dls + NODE_RC * ld;        // rstd  [NODE_RC]
  float* ws = rs + NODE_RC;              // Wqkv  [Dh][ld3]
  stage_weight(ws, ld3, a.Wqkv, Dh, D3);
  const int nkt = (Dh + 15) / 16, nct = (D3 + 15) / 16, ntl = nkt * nct;  // <= 4 x 12
  v4f accW[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) accW[j] = (v4f){0.f, 0.f, 0.f, 0.f};
  float accBq = 0.f, accG = 0.f, accBt = 0.f;
  for (int r0 = chunk * NODE_RC; r0 < min(N, (chunk + 1) * NODE_RC); r0 += NODE_RC) {  // one chunk
    const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
    const size_t row0 = (size_t)b * N + r0;
    __syncthreads();
    stage_rows(xs, ld, a.h + row0 * Dh, Dh, nr, nrp);
    // dQKV rows: packed dq + dK/dV partials summed over the row-ranges (16-byte units)
    {
      const int U = nrp * 48, NT = (int)blockDim.x;
      for (int i0 = t; i0 < U; i0 += 4 * NT) {   // up to 4 units x (NQP | NLR) 16-byte loads in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT, r = i / 48, pos4 = (i % 48) * 4, s = pos4 >> 6;
          float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i < U && r < nr) {
            if (s == 0) {
#pragma unroll 4
              for (int qp = 0; qp < a.NQP; ++qp) {
                const float4 w = *reinterpret_cast<const float4*>(
                    a.dqp + (((size_t)b * a.NQP + qp) * N + r0 + r) * 64 + pos4);
                acc4.x += w.x; acc4.y += w.y; acc4.z += w.z; acc4.w += w.w;
              }
            } else {
#pragma unroll 4
              for (int lr = 0; lr < a.NLR; ++lr) {
                const float4 w = *reinterpret_cast<const float4*>(
                    a.dkvp + ((((size_t)b * a.NLR + lr) * N + r0 + r) * 2 + (s - 1)) * 64 + (pos4 & 63));
                acc4.x += w.x; acc4.y += w.y; acc4.z += w.z; acc4.w += w.w;
              }
            }
          }
          v[u] = acc4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NT, r = i / 48, pos4 = (i % 48) * 4, s = pos4 >> 6;
          if (i < U) {
            const int qq = (pos4 >> 4) & 3, k0 = (pos4 >> 1) & 7;
            float* d = dqs + r * ld3 + s * Dh + k0 * 8 + 2 * qq;
            if (k0 < a.DK) { d[0] = v[u].x; d[1] = v[u].y; }
            if (k0 + 1 < a.DK) { d[8] = v[u].z; d[9] = v[u].w; }
          }
        }
      }
    }
    __syncthreads();
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // LN forward statistics -> xhat in place
      float* x = xs + (rb + q) * ld;
      float v[4], s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; v[i] = c < Dh ? x[c] : 0.f; s += v[i]; }
      const float mu = sum16(s) / Dh;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) { v[i] -= mu; ss = fmaf(v[i], v[i], ss); } }
      const float rstd = rsqrtf(sum16(ss) / Dh + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) x[c] = v[i] * rstd; }
      if (p == 0) rs[rb + q] = rstd;
    }
    // d(h_ln)[row][kk] = sum_c dQKV[row][c] * Wqkv[kk][c]
    const int ntr = nrp / 16;
    for (int tile = wave; tile < ntr * nkt; tile += NW) {
      const int rt = tile / nkt, kt = tile % nkt, kk = kt * 16 + p;
      const bool colok = kk < Dh;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = mm_xwt(dqs + rt * 16 * ld3, ld3, ws + kt * 16 * ld3, ld3, D3, p, q, colok, acc);
      if (colok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dls[(rt * 16 + 4 * q + r) * ld + kk] = acc[r];
      }
    }
    __syncthreads();
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // LayerNorm backward + residual
      const int row = rb + q;
      const float* x = xs + row * ld;
      const float* dl = dls + row * ld;
      float dx[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = p + 16 * i;
        dx[i] = c < Dh ? dl[c] * a.nm_g[c] : 0.f;
        m1 += dx[i];
        m2 = fmaf(dx[i], c < Dh ? x[c] : 0.f, m2);
      }
      m1 = sum16(m1) / Dh;
      m2 = sum16(m2) / Dh;
      const float rstd = rs[row];
      if (row < nr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = p + 16 * i;
          if (c < Dh) {
            const size_t o = (row0 + row) * Dh + c;
            a.dh[o] = a.dh_out[o] + rstd * (dx[i] - m1 - x[c] * m2);
          }
        }
      }
    }
    // dWqkv[kk][c] += sum_rows h_ln[row][kk] * dQKV[row][c]
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int idx = wave + NW * j;
      if (idx < ntl) {
        const int kt = idx / nct, ct = idx % nct, kk = kt * 16 + p;
        const float g = kk < Dh ? a.nm_g[kk] : 0.f, bt = kk < Dh ? a.nm_b[kk] : 0.f;
#pragma unroll 4
        for (int rr = 0; rr < nrp; rr += 4)
          accW[j] = MFMA(fmaf(xs[(rr + q) * ld + kk], g, bt), dqs[(rr + q) * ld3 + ct * 16 + p], accW[j]);
      }
    }
    // column sums with independent partial accumulators (rows >= nr are zero-padded in LDS)
    if (t < D3) {
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      for (int r = 0; r < nrp; r += 4) {
        b0 += dqs[r * ld3 + t]; b1 += dqs[(r + 1) * ld3 + t]; b2 += dqs[(r + 2) * ld3 + t]; b3 += dqs[(r + 3) * ld3 + t];
      }
      accBq += (b0 + b1) + (b2 + b3);
    } else if (t >= 256 && t < 256 + Dh) {   // a different wavefront takes the LayerNorm parameter sums
      const int c = t - 256;
      float g0 = 0.f, g1 = 0.f, s0 = 0.f, s1 = 0.f;
      for (int r = 0; r < nrp; r += 2) {
        const float d0 = dls[r * ld + c], d1 = dls[(r + 1) * ld + c];
        g0 = fmaf(d0, xs[r * ld + c], g0); g1 = fmaf(d1, xs[(r + 1) * ld + c], g1);
        s0 += d0; s1 += d1;
      }
      accG += g0 + g1;
      accBt += s0 + s1;
    }
  }
  float* part = a.npart + (size_t)blockIdx.x * (Dh * D3 + D3 + 2 * Dh + Dh * Dh + Dh);
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int idx = wave + NW * j;
    if (idx < ntl) {
      const int kt = idx / nct, ct = idx % nct, c = ct * 16 + p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt * 16 + 4 * q + r;
        if (kk < Dh && c < D3) part[kk * D3 + c] = accW[j][r];
      }
    }
  }
  if (t < D3) part[Dh * D3 + t] = accBq;
  else if (t >= 256 && t < 256 + Dh) {
    part[Dh * D3 + D3 + (t - 256)] = accG;
    part[Dh * D3 + D3 + Dh + (t - 256)] = accBt;
  }
}

// ------------------------------------------------------------ final reduce -----
struct SumSeg { const float* src; float* dst; int n, np, stride, nblk; };
#define SUM_MAX_SEG 77  // 11 layers x 7 segments: fits the 4 KiB kernel-argument block
struct SumArgs { SumSeg seg[SUM_MAX_SEG]; };

// 64 outputs per workgroup, the partial axis split over 4 wavefronts
__global__ void __launch_bounds__(256) k_sum_segments(SumArgs s) {
  __shared__ float red[4][64];
  const SumSeg sg = s.seg[blockIdx.y];
  if ((int)blockIdx.x >= sg.nblk) return;
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (o < sg.n) {
    int pi = pg;
    for (; pi + 12 < sg.np; pi += 16) {
      v0 += sg.src[(size_t)pi * sg.stride + o];
      v1 += sg.src[(size_t)(pi + 4) * sg.stride + o];
      v2 += sg.src[(size_t)(pi + 8) * sg.stride + o];
      v3 += sg.src[(size_t)(pi + 12) * sg.stride + o];
    }
    for (; pi < sg.np; pi += 4) v0 += sg.src[(size_t)pi * sg.stride + o];
  }
  red[pg][threadIdx.x & 63] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (pg == 0 && o < sg.n) sg.dst[o] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// T[c][i], s[i], R[c][h|8] -> grads of norm_edge, attention_gates, dense_edge_b, dense_edge_r.
// One workgroup per layer (blockIdx.x), so a whole stack finishes in one launch.
struct EdgeGradLayer {
  const float *ne_g, *ne_b, *Wg, *We, *ered;
  float *g_ne_g, *g_ne_b, *g_Wg, *g_bg, *g_We, *g_be, *g_Wr, *g_br;
};
#define EPG_MAX_LAYERS 16
struct EdgeGradArgs { EdgeGradLayer L[EPG_MAX_LAYERS]; int De; uint32_t flags; };

__global__ void __launch_bounds__(256) k_edge_param_grads(EdgeGradArgs ga) {
  const EdgeGradLayer a = ga.L[blockIdx.x];
  const int DE = ga.De, DEP = ((DE + 15) / 16) * 16;
  const bool gated = (ga.flags & EGT_BF_GATE) != 0;
  const float* T = a.ered;
  const float* s = a.ered + DEP * 16;
  const float* R = s + 16;
  for (int idx = threadIdx.x; idx < DE * 16; idx += 256) {
    const int c = idx >> 4, i = idx & 15, hd = col_head(i);
    const float v = a.ne_g[c] * T[c * 16 + i] + a.ne_b[c] * s[i];
    if (col_is_gate(i)) { if (gated) a.g_Wg[c * BH + hd] = v; }
    else a.g_We[c * BH + hd] = v;
  }
  if (threadIdx.x < 16) {
    const int i = threadIdx.x, hd = col_head(i);
    if (col_is_gate(i)) { if (gated) a.g_bg[hd] = s[i]; }
    else a.g_be[hd] = s[i];
  }
  for (int c = threadIdx.x; c < DE; c += 256) {
    float dg = 0.f, db = 0.f;
    for (int i = 0; i < 16; ++i) {
      const int hd = col_head(i);
      float w;
      if (col_is_gate(i)) w = gated ? a.Wg[c * BH + hd] : 0.f;
      else w = a.We[c * BH + hd];
      dg = fmaf(w, T[c * 16 + i], dg);
      db = fmaf(w, s[i], db);
    }
    a.g_ne_g[c] = dg;
    a.g_ne_b[c] = db;
    a.g_br[c] = R[c * 16 + 8];
  }
  for (int idx = threadIdx.x; idx < BH * DE; idx += 256) {
    const int hd = idx / DE, c = idx % DE;
    a.g_Wr[idx] = R[c * 16 + hd];
  }
}

// ---------------------------------------------------------------- launchers ----
static int node_chunks(const BlockArgs& a) { return (a.N + NODE_RC - 1) / NODE_RC; }
static size_t lds_rows(int Dh, int mult) { return (size_t)mult * NODE_RC * (Dh + LDP) * 4; }

void egt_node_launch_pre(BlockArgs& a, hipStream_t st) {
  size_t lds = lds_rows(a.Dh, 1) + (size_t)a.Dh * (3 * a.Dh + LDP) * 4;
  if (lds < 1024) lds = 1024;  // prep workgroup scratch
  (void)hipFuncSetAttribute((const void*)k_node_pre, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_node_pre", k_node_pre, dim3(a.B * node_chunks(a) + (a.prep ? 1 : 0)), dim3(512), lds, st, a);
}

void egt_node_launch_post(BlockArgs& a, hipStream_t st) {
  const size_t lds = lds_rows(a.Dh, 1) + (size_t)a.Dh * (a.Dh + LDP) * 4;
  EGT_LAUNCH("k_node_post", k_node_post, dim3(a.B * node_chunks(a)), dim3(512), lds, st, a);
}

void egt_node_launch_post_bwd(BlockArgs& a, hipStream_t st) {
  const size_t lds = lds_rows(a.Dh, 2) + (size_t)a.Dh * (a.Dh + LDP) * 4 + (size_t)4 * NODE_RC * 8 * 4;
  EGT_LAUNCH("k_node_post_bwd", k_node_post_bwd, dim3(a.B * node_chunks(a) + (a.prep ? 1 : 0)), dim3(512), lds, st, a);
}

void egt_node_launch_pre_bwd(BlockArgs& a, hipStream_t st) {
  const size_t lds = lds_rows(a.Dh, 2) + ((size_t)NODE_RC * (3 * a.Dh + LDP) + NODE_RC +
                                           (size_t)a.Dh * (3 * a.Dh + LDP)) * 4;
  (void)hipFuncSetAttribute((const void*)k_node_pre_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_node_pre_bwd", k_node_pre_bwd, dim3(a.B * node_chunks(a)), dim3(512), lds, st, a);
}

void egt_node_launch_prep(BlockArgs* as, int n, hipStream_t st) {
  PrepArgs pa{};
  pa.De = as[0].De; pa.flags = as[0].flags;
  for (int l = 0; l < n; ++l) {
    const BlockArgs& a = as[l];
    pa.L[l] = PrepLayer{a.ne_g, a.ne_b, a.Wg, a.bg, a.We, a.be, a.pw};
  }
  EGT_LAUNCH("k_edge_prep", k_edge_prep, dim3(n), dim3(256), 0, st, pa);
}

// Reduce the per-workgroup partials of `n` layers (one BlockArgs each, with their own
// npart / epart / ered and gradient pointers) and finish the edge-parameter gradients.
void egt_node_launch_reduce(BlockArgs* as, int n, int nwg_bwd, int EP, int npart_stride, hipStream_t st) {
  for (int l0 = 0; l0 < n; l0 += 11) {
    const int nl = (n - l0 < 11) ? (n - l0) : 11;
    SumArgs s{};
    int maxblk = 0, k = 0;
    auto seg = [&](const float* src, float* dst, int cnt, int npart, int stride) {
      s.seg[k] = SumSeg{src, dst, cnt, npart, stride, (cnt + 63) / 64};
      if (s.seg[k].nblk > maxblk) maxblk = s.seg[k].nblk;
      ++k;
    };
    for (int l = l0; l < l0 + nl; ++l) {
      BlockArgs& a = as[l];
      const int Dh = a.Dh, D3 = 3 * Dh, nnp = a.B * node_chunks(a);
      const float* np = a.npart;
      int o = 0;
      seg(np + o, a.g_Wqkv, Dh * D3, nnp, npart_stride); o += Dh * D3;
      seg(np + o, a.g_bqkv, D3, nnp, npart_stride); o += D3;
      seg(np + o, a.g_nm_g, Dh, nnp, npart_stride); o += Dh;
      seg(np + o, a.g_nm_b, Dh, nnp, npart_stride); o += Dh;
      seg(np + o, a.g_Wo, Dh * Dh, nnp, npart_stride); o += Dh * Dh;
      seg(np + o, a.g_bo, Dh, nnp, npart_stride);
      seg(a.epart, a.ered, EP, nwg_bwd, EP);
    }
    EGT_LAUNCH("k_sum_segments", k_sum_segments, dim3(maxblk, k), dim3(256), 0, st, s);
  }
  for (int l0 = 0; l0 < n; l0 += EPG_MAX_LAYERS) {
    const int nl = (n - l0 < EPG_MAX_LAYERS) ? (n - l0) : EPG_MAX_LAYERS;
    EdgeGradArgs ga{};
    ga.De = as[0].De; ga.flags = as[0].flags;
    for (int l = 0; l < nl; ++l) {
      const BlockArgs& a = as[l0 + l];
      ga.L[l] = EdgeGradLayer{a.ne_g, a.ne_b, a.Wg, a.We, a.ered, a.g_ne_g, a.g_ne_b, a.g_Wg, a.g_bg,
                              a.g_We, a.g_be, a.g_Wr, a.g_br};
    }
    EGT_LAUNCH("k_edge_param_grads", k_edge_param_grads, dim3(nl), dim3(256), 0, st, ga);
  }
}
