// Node-side kernels of the fused attention block (O(N*Dh) work, once per layer):
//   k_node_pre      : norm_mha -> dense_qkv, written packed per head-pair
//                     (graph_xformer_model_base.py:109,113)         [+ edge-weight prep]
//   k_node_post     : dense_mha + res_mha (:136,140)
//   k_node_bwd      : [layer l] dQKV -> d(h_ln) -> LN backward -> dh ; bias / LN-parameter sums
//                     [layer l-1] dV_att = dh.Wo^T (packed), delta = sum_k dV_att*V_att
//                     (on the headline geometry both run as the prologue of the backward pair
//                      kernel, egt_block_dev.h:bwd_node_prologue; this kernel then only closes the chain)
//   k_node_wgrads   : dWqkv, dWo of every layer in one launch (deferred, off the critical path)
//   k_sum_segments  : deterministic reduction of all per-workgroup partials
//   k_edge_param_grads : T,s,R -> grads of norm_edge / attention_gates / dense_edge_b / dense_edge_r
// One workgroup per 32 node rows; every contraction is a 16x16 tile on
// v_mfma_f32_16x16x4_f32 with the activation rows staged in LDS.
#include <stdlib.h>

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
struct PrepLayer { const float *ne_g, *ne_b, *Wg, *bg, *We, *be; float* pw; const float *Wqkv, *Wo; float* wfrag; };

// one 16-byte piece of the fragment-major weight copies (egt_block.h: WFRAG_*); o = index of the piece
__device__ __forceinline__ float4 wfrag_piece(const PrepLayer& a, int o, int Dh) {
  const int D3 = 3 * Dh;
  float v[4];
  if (o < 3072) {                       // BWQ
    const int lane = o & 63, s = (o >> 6) % 12, w = o / 768, p = lane & 15, q = lane >> 4, kk = 16 * w + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 48 * q + 4 * s + i, cc = c & 63;
      v[i] = (kk < Dh && cc < Dh) ? a.Wqkv[(size_t)kk * D3 + (c >> 6) * Dh + cc] : 0.f;
    }
  } else if (o < 4096) {                // BWO
    const int x = o - 3072, lane = x & 63, s = (x >> 6) & 3, w = x >> 8, p = lane & 15, q = lane >> 4, row = 16 * w + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = 16 * q + 4 * s + i;
      v[i] = (row < Dh && col < Dh) ? a.Wo[(size_t)row * Dh + col] : 0.f;
    }
  } else if (o < 5120) {                // FWO
    const int x = o - 4096, lane = x & 63, s4 = (x >> 6) & 3, w = x >> 8, p = lane & 15, q = lane >> 4, c = 16 * w + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * (4 * s4 + i) + q;
      v[i] = (c < Dh && k < Dh) ? a.Wo[(size_t)k * Dh + c] : 0.f;
    }
  } else {                              // FWQ
    const int x = o - 5120, lane = x & 63, s4 = (x >> 6) & 3, j = (x >> 8) % 3, w = x / 768, p = lane & 15, q = lane >> 4;
    const int cq = (w + 4 * j) * 16 + p, gc = (cq >> 6) * Dh + (cq & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * (4 * s4 + i) + q;
      v[i] = ((cq & 63) < Dh && k < Dh) ? a.Wqkv[(size_t)k * D3 + gc] : 0.f;
    }
  }
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ void prep_wfrag(const PrepLayer& a, int Dh) {
  if (!a.wfrag || Dh > 64) return;
  const int NT = blockDim.x;
  for (int o0 = threadIdx.x; o0 < WFRAG_FLOATS / 4; o0 += 4 * NT) {   // four pieces (16 scalar loads) in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = wfrag_piece(a, min(o0 + u * NT, WFRAG_FLOATS / 4 - 1), Dh);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (o0 + u * NT < WFRAG_FLOATS / 4) reinterpret_cast<float4*>(a.wfrag)[o0 + u * NT] = v[u];
  }
}

__device__ void prep_layer(const PrepLayer& a, int De, int Dh, bool gated, float* red) {
  prep_wfrag(a, Dh);
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
  const PrepLayer L{a.ne_g, a.ne_b, a.Wg, a.bg, a.We, a.be, a.pw, a.Wqkv, a.Wo, a.wfrag};
  prep_layer(L, a.De, a.Dh, (a.flags & EGT_BF_GATE) != 0, red);
}

// the same preparation for every layer of a stack in one launch (one workgroup per layer)
#define PREP_MAX_LAYERS 48   // (80 bytes per layer: the 4 KiB kernel-argument block)
struct PrepArgs { PrepLayer L[PREP_MAX_LAYERS]; int De, Dh; uint32_t flags; };
__global__ void __launch_bounds__(256) k_edge_prep(PrepArgs pa) {
  __shared__ float red[256];
  prep_layer(pa.L[blockIdx.x], pa.De, pa.Dh, (pa.flags & EGT_BF_GATE) != 0, red);
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

// Two-phase staging for kernels with several independent inputs: issue every global load
// first (unconditional, clamped addresses), commit to LDS afterwards, so that one memory
// round trip covers all of them.  Unit i = threadIdx.x + u * blockDim.x is one 16-byte load.
template <int NU>
struct Stg { float4 v[NU]; };

// Pin a staged value: the compiler may neither sink the load below this point nor reorder the
// commits above it, so all loads issued before the first pin share one memory round trip.
__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <int NU>
__device__ __forceinline__ void stg_pin(Stg<NU>& s) {
#pragma unroll
  for (int u = 0; u < NU; ++u) pin4(s.v[u]);
}
__device__ __forceinline__ float4 sel4(bool ok, float4 v) {   // component-wise (a float4 ternary goes through scratch)
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// weights: `Wa` is W rounded down to 16 bytes by the host (parameter views need not be aligned;
// the unaligned case loads harmlessly from Wa and commits with scalar copies from W)
template <int NU>
__device__ __forceinline__ void stg_issue_weight(Stg<NU>& s, const float* Wa, int n4) {
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = threadIdx.x + u * blockDim.x;
    s.v[u] = *reinterpret_cast<const float4*>(Wa + (size_t)(i < n4 ? i : 0) * 4);
  }
}
template <int NU>
__device__ __forceinline__ void stg_commit_weight(const Stg<NU>& s, float* ws, int ldw, const float* W, bool al,
                                                  int rows, int width) {
  const int n4 = rows * width / 4;
  if (al) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      if (i < n4) {
        const int r = (i * 4) / width, c = (i * 4) % width;
        *reinterpret_cast<float4*>(ws + r * ldw + c) = s.v[u];
      }
    }
  } else {
    for (int i = threadIdx.x; i < rows * width; i += blockDim.x) ws[(i / width) * ldw + (i % width)] = W[i];
  }
}
// rows of a [.., width] tensor: rows >= nr read row 0 and are committed as zeros (up to nrp rows)
template <int NU>
__device__ __forceinline__ void stg_issue_rows(Stg<NU>& s, const float* src, int width, int nr, int nrp) {
  const int w4 = width >> 2, n4 = nrp * w4;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = threadIdx.x + u * blockDim.x, r = i / w4, c4 = i % w4;
    const bool ok = i < n4 && r < nr;
    s.v[u] = *reinterpret_cast<const float4*>(src + (ok ? (size_t)r * width + c4 * 4 : 0));
  }
}
template <int NU>
__device__ __forceinline__ void stg_commit_rows(const Stg<NU>& s, float* xs, int ld, int width, int nr, int nrp) {
  const int w4 = width >> 2, n4 = nrp * w4;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = threadIdx.x + u * blockDim.x, r = i / w4, c4 = i % w4;
    if (i < n4) *reinterpret_cast<float4*>(xs + r * ld + c4 * 4) = sel4(r < nr, s.v[u]);
  }
}

// --------------------------------------------------------------- node: pre -----
__device__ __forceinline__ void node_pre_rows(const BlockArgs& a, float* sm, int NCH) {
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, ld = Dh + LDP, D3 = 3 * Dh;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;                 // [NODE_RC][ld]
  float* ws = xs + NODE_RC * ld;  // Wqkv [Dh][ldw]
  const int ldw = D3 + LDP;
  const int nct = (D3 + 15) / 16;
  for (int r0 = chunk * NODE_RC; r0 < min(N, (chunk + 1) * NODE_RC); r0 += NODE_RC) {  // one chunk
    const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
    const size_t row0 = (size_t)b * N + r0;
    if (Dh * D3 / 4 <= 6 * (int)blockDim.x && NODE_RC * (Dh >> 2) <= (int)blockDim.x) {
      // Wqkv (<= 6 16-byte units per thread) and the chunk's rows (1) in ONE memory round trip: issue, then commit (the staging
      // helpers one after the other were three dependent round trips at the head of every forward)
      Stg<6> sW;
      Stg<1> sX;
      const bool al = (reinterpret_cast<uintptr_t>(a.Wqkv) & 15) == 0;
      stg_issue_weight(sW, reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(a.Wqkv) & ~(uintptr_t)15), Dh * D3 / 4);
      stg_issue_rows(sX, a.h + row0 * Dh, Dh, nr, nrp);
      stg_pin(sW); stg_pin(sX);
      stg_commit_weight(sW, ws, ldw, a.Wqkv, al, Dh, D3);
      stg_commit_rows(sX, xs, ld, Dh, nr, nrp);
    } else {
      stage_weight(ws, ldw, a.Wqkv, Dh, D3);
      stage_rows(xs, ld, a.h + row0 * Dh, Dh, nr, nrp);
    }
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

__global__ void __launch_bounds__(512) k_node_pre(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  if ((int)blockIdx.x == a.B * NCH) { prep_device(a, sm); return; }
  node_pre_rows(a, sm, NCH);
}
// The first launch of a stack's forward: layer 0's packed QKV rows plus the edge-weight preparation of EVERY layer as NL extra
// workgroups (k_edge_prep as a launch of its own was ~6.5 us of latency for 10 workgroups in front of every step).
template <int NL> struct PrepArgsN { PrepLayer L[NL]; int n; };
template <int NL>
__global__ void __launch_bounds__(512) k_node_pre_stack(BlockArgs a, PrepArgsN<NL> pa) {
  static_assert(sizeof(BlockArgs) + sizeof(PrepArgsN<NL>) <= 4096, "kernel-argument block");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC, extra = (int)blockIdx.x - a.B * NCH;
  if (extra >= 0) { prep_layer(pa.L[extra], a.De, a.Dh, (a.flags & EGT_BF_GATE) != 0, sm); return; }
  node_pre_rows(a, sm, NCH);
}

// -------------------------------------------------------------- node: post -----
__global__ void __launch_bounds__(512) k_node_post(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, ld = Dh + LDP;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
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

// ------------------------------------------------------------ node: backward -----
// One launch per layer on the d_h critical path (plus one at the top of the chain):
//   pre part (layer l):  dQKV rows = packed dQ + dK/dV partial sums  -> d(h_ln) = dQKV.Wqkv^T
//                        -> LayerNorm backward + residual -> dh(l);  dQKV rows are kept for
//                        the deferred weight-gradient kernel; bias / LN-parameter column sums
//   dv part (layer l-1, or the top layer itself in the first call):
//                        dV_att = dh.Wo^T (packed), delta = sum_k dV_att*V_att, dbo column sums
// The GEMM-shaped weight gradients (dWqkv, dWo) are NOT computed here: k_node_wgrads does them
// for every layer of the stack in one launch, off the critical path.
// small per-workgroup partials of a layer: spart [dbqkv 3Dh | dgamma Dh | dbeta Dh], sbo [dbo Dh]
struct NodeBwdArgs {
  const float *wq_a, *wo_a;   // Wqkv / dv_Wo rounded down to 16 bytes
  int wq_al, wo_al;           // ... and whether they were aligned to begin with
  const float *dv_Wo, *dv_v_att, *dv_dh_src;   // dv_dh_src: rows of dh' when do_pre == 0
  float *dv_stats, *dv_dvp, *dv_sbo;
};

template <bool PRE, bool DV, int NP>   // compile-time roles; NP: partials per gathered unit (0: run-time counts)
__global__ void __launch_bounds__(512, 2) k_node_bwd(BlockArgs a, NodeBwdArgs x) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NCH = (a.N + NODE_RC - 1) / NODE_RC;
  if ((int)blockIdx.x == a.B * NCH) { prep_device(a, sm); return; }
  const int Dh = a.Dh, N = a.N, b = blockIdx.x / NCH, chunk = blockIdx.x % NCH, t = threadIdx.x, D3 = 3 * Dh;
  const int ld = Dh + LDP, ld3 = D3 + LDP, SP = D3 + 2 * Dh;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;                        // xhat   [NODE_RC][ld]
  float* dls = xs + NODE_RC * ld;        // d h_ln [NODE_RC][ld]
  float* dhs = dls + NODE_RC * ld;       // dh     [NODE_RC][ld]
  float* vs = dhs + NODE_RC * ld;        // v_att  [NODE_RC][ld]   (dv layer)
  float* rs = vs + NODE_RC * ld;         // rstd   [NODE_RC]
  float* dlp = rs + NODE_RC;             // delta partials [4][NODE_RC][8]
  float* wo = dlp + 4 * NODE_RC * 8;     // Wo     [Dh][ld]        (dv layer)
  float* dqs = wo + Dh * ld;             // dQKV   [NODE_RC][ld3]
  float* ws = dqs + NODE_RC * ld3;       // Wqkv   [Dh][ld3]
  const int r0 = chunk * NODE_RC;
  const int nr = min(NODE_RC, N - r0), nrp = (nr + 15) & ~15;
  const size_t row0 = (size_t)b * N + r0;
  const int nkt = (Dh + 15) / 16;

  // ---- every global input of the kernel in ONE memory round trip: issue, then commit ----
  // (512 threads: Wqkv <= 6 units/thread, Wo <= 2, the 32-row tensors 1, the dQKV gather 3)
  Stg<6> sWq;
  Stg<2> sWo;
  Stg<1> sV, sX;
  float4 gq[3];
  float dho[4], gmm[4];   // dh' and gamma of the lane's LayerNorm-backward elements: row 4*wave + q, columns p + 16 i
  if (DV) {
    stg_issue_weight(sWo, x.wo_a, Dh * Dh / 4);
    stg_issue_rows(sV, x.dv_v_att + row0 * Dh, Dh, nr, nrp);
  }
  if (PRE) {
    stg_issue_weight(sWq, x.wq_a, Dh * D3 / 4);
    stg_issue_rows(sX, a.h + row0 * Dh, Dh, nr, nrp);
    {
      const int row = 4 * wave + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = p + 16 * i;
        const bool ok = row < nr && c < Dh;
        dho[i] = a.dh_out[ok ? (row0 + row) * Dh + c : row0 * Dh];
        gmm[i] = a.nm_g[c < Dh ? c : 0];
      }
    }
    // dQKV rows: packed dq + dK/dV partials summed over the row-ranges (16-byte units)
    const int U = nrp * 48;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = t + u * 512, r = i / 48, pos4 = (i % 48) * 4, sx = pos4 >> 6;
      const bool ok = i < U && r < nr;
      const int rc = ok ? r : 0;
      float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const int np = sx == 0 ? a.NQP : a.NLR;
      const float* base = sx == 0 ? a.dqp + ((size_t)b * a.NQP * N + r0 + rc) * 64 + pos4
                                  : a.dkvp + (((size_t)b * a.NLR * N + r0 + rc) * 2 + (sx - 1)) * 64 + (pos4 & 63);
      const size_t pstride = sx == 0 ? (size_t)N * 64 : (size_t)N * 128;
      if (NP > 0) {   // NQP == NLR == NP: every load of the gather is issued back to back
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
          const float4 w = *reinterpret_cast<const float4*>(base + pi * pstride);
          acc4.x += w.x; acc4.y += w.y; acc4.z += w.z; acc4.w += w.w;
        }
      } else {
        const int npmax = max(a.NQP, a.NLR);   // uniform trip count; lanes past their own count re-read and drop
#pragma unroll 4
        for (int pi = 0; pi < npmax; ++pi) {
          const float4 w = *reinterpret_cast<const float4*>(base + min(pi, np - 1) * pstride);
          const float f = pi < np ? 1.0f : 0.0f;
          acc4.x = fmaf(w.x, f, acc4.x); acc4.y = fmaf(w.y, f, acc4.y);
          acc4.z = fmaf(w.z, f, acc4.z); acc4.w = fmaf(w.w, f, acc4.w);
        }
      }
      gq[u] = sel4(ok, acc4);
    }
  } else {
    stg_issue_rows(sX, x.dv_dh_src + row0 * Dh, Dh, nr, nrp);
  }
  if (DV) { stg_pin(sWo); stg_pin(sV); }
  stg_pin(sX);
  if (PRE) {
    stg_pin(sWq);
#pragma unroll
    for (int u = 0; u < 3; ++u) pin4(gq[u]);
  }
  if (DV) {
    stg_commit_weight(sWo, wo, ld, x.dv_Wo, x.wo_al != 0, Dh, Dh);
    stg_commit_rows(sV, vs, ld, Dh, nr, nrp);
  }
  if (PRE) {
    stg_commit_weight(sWq, ws, ld3, a.Wqkv, x.wq_al != 0, Dh, D3);
    stg_commit_rows(sX, xs, ld, Dh, nr, nrp);
    const int U = nrp * 48;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = t + u * 512, r = i / 48, pos4 = (i % 48) * 4, sx = pos4 >> 6;
      if (i < U) {
        const int qq = (pos4 >> 4) & 3, k0 = (pos4 >> 1) & 7;
        float* d = dqs + r * ld3 + sx * Dh + k0 * 8 + 2 * qq;
        if (k0 < a.DK) { d[0] = gq[u].x; d[1] = gq[u].y; }
        if (k0 + 1 < a.DK) { d[8] = gq[u].z; d[9] = gq[u].w; }
      }
    }
  } else {
    stg_commit_rows(sX, dhs, ld, Dh, nr, nrp);
  }
  __syncthreads();

  if (PRE) {
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // LN forward statistics -> xhat in place
      float* xr = xs + (rb + q) * ld;
      float v[4], s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; v[i] = c < Dh ? xr[c] : 0.f; s1 += v[i]; }
      const float mu = sum16(s1) / Dh;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) { v[i] -= mu; ss = fmaf(v[i], v[i], ss); } }
      const float rstd = rsqrtf(sum16(ss) / Dh + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) xr[c] = v[i] * rstd; }
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
    // dQKV rows out, natural channel order, for k_node_wgrads
    for (int i = t; i < nr * (D3 / 4); i += blockDim.x) {
      const int r = i / (D3 / 4), c4 = (i % (D3 / 4)) * 4;
      *reinterpret_cast<float4*>(a.dqkv_sv + (row0 + r) * D3 + c4) = *reinterpret_cast<const float4*>(dqs + r * ld3 + c4);
    }
    __syncthreads();
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // LayerNorm backward + residual
      const int row = rb + q;
      const float* xr = xs + row * ld;
      const float* dl = dls + row * ld;
      float dx[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = p + 16 * i;
        dx[i] = c < Dh ? dl[c] * gmm[i] : 0.f;
        m1 += dx[i];
        m2 = fmaf(dx[i], c < Dh ? xr[c] : 0.f, m2);
      }
      m1 = sum16(m1) / Dh;
      m2 = sum16(m2) / Dh;
      const float rstd = rs[row];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = p + 16 * i;
        if (c < Dh) {
          float dv = 0.f;
          if (row < nr) {
            const size_t o = (row0 + row) * Dh + c;
            dv = dho[i] + rstd * (dx[i] - m1 - xr[c] * m2);
            a.dh[o] = dv;
          }
          dhs[row * ld + c] = dv;
        }
      }
    }
    // column sums (rows >= nr are zero-padded in LDS): dbqkv | dgamma, dbeta
    float* sp = a.spart + (size_t)blockIdx.x * SP;
    if (t < D3) {
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      for (int r = 0; r < nrp; r += 4) {
        b0 += dqs[r * ld3 + t]; b1 += dqs[(r + 1) * ld3 + t]; b2 += dqs[(r + 2) * ld3 + t]; b3 += dqs[(r + 3) * ld3 + t];
      }
      sp[t] = (b0 + b1) + (b2 + b3);
    } else if (t >= 256 && t < 256 + Dh) {   // a different wavefront takes the LayerNorm parameter sums
      const int c = t - 256;
      float g0 = 0.f, g1 = 0.f, s0 = 0.f, s1 = 0.f;
      for (int r = 0; r < nrp; r += 2) {
        const float d0 = dls[r * ld + c], d1 = dls[(r + 1) * ld + c];
        g0 = fmaf(d0, xs[r * ld + c], g0); g1 = fmaf(d1, xs[(r + 1) * ld + c], g1);
        s0 += d0; s1 += d1;
      }
      sp[D3 + c] = g0 + g1;
      sp[D3 + Dh + c] = s0 + s1;
    }
    if (DV) __syncthreads();
  }

  if (DV) {
    // dV_att = dh'.Wo^T: (row tile, i tile) pairs spread over the waves; the per-head delta
    // contributions of each i tile go through LDS and are summed in fixed order
    const int nit = nkt;
    for (int tile = wave; tile < (nrp / 16) * nit; tile += NW) {
      const int rt = tile / nit, it = tile % nit;
      const int i = it * 16 + p;
      const bool colok = i < Dh;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      acc = mm_xwt(dhs + rt * 16 * ld, ld, wo + it * 16 * ld, ld, Dh, p, q, colok, acc);
      const int k = i >> 3, hh = i & 7;
      const int pos = (hh >> 1) * 16 + k * 2 + (hh & 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        if (colok && row < nr) x.dv_dvp[(row0 + row) * 64 + pos] = acc[r];
        float pr = colok ? acc[r] * vs[row * ld + i] : 0.f;
        pr += lane_xor<8>(pr);   // the tile's two k values of head p&7
        if (p < 8) dlp[(it * NODE_RC + row) * 8 + p] = pr;
      }
    }
    if (t >= 256 && t < 256 + Dh) {   // dbo of the dv layer
      const int c = t - 256;
      float b0 = 0.f, b1 = 0.f;
      for (int r = 0; r < nrp; r += 2) { b0 += dhs[r * ld + c]; b1 += dhs[(r + 1) * ld + c]; }
      x.dv_sbo[(size_t)blockIdx.x * Dh + c] = b0 + b1;
    }
    __syncthreads();
    for (int i = t; i < nr * 8; i += blockDim.x) {
      float dl = 0.f;
      for (int it = 0; it < nit; ++it) dl += dlp[it * NODE_RC * 8 + i];
      x.dv_stats[(row0 * BH + i) * 4 + 2] = dl;
    }
    if (a.DK < 8) {
      for (int i = t; i < nr * 64; i += blockDim.x)
        if ((((i & 63) >> 1) & 7) >= a.DK) x.dv_dvp[row0 * 64 + i] = 0.f;
    }
  }
}

// ----------------------------------------------------- node: weight gradients -----
// dWqkv[l] = h_ln(l)^T . dQKV(l)   and   dWo[l] = V_att(l)^T . dh'(l)   for EVERY layer in one
// launch: grid = (row chunks, layers).  A workgroup walks WG_ROWS node rows (rows are flat over
// the batch: no graph structure is needed) in steps of 32, recomputes norm_mha for them, and
// accumulates its 16x16 output tiles on MFMA with the row index as the contraction axis.
// Partials [layer][chunk][Dh*3Dh + Dh*Dh] are reduced by k_sum_segments.
#define WG_ROWS 128
struct WGradLayer { const float *h, *nm_g, *nm_b, *dqkv, *v_att, *dh_out; float* part; };
struct WGradArgs { WGradLayer L[64]; int rows, Dh, rows_per_wg; float ln_eps; };

template <bool D64>   // D64: Dh == 64, every tile exists -> no guards around the MFMAs
__global__ void __launch_bounds__(512, 4) k_node_wgrads(WGradArgs wa) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const WGradLayer L = wa.L[blockIdx.y];
  const int Dh = D64 ? 64 : wa.Dh, D3 = 3 * Dh, t = threadIdx.x;
  const int ld = Dh + 16, ld3 = D3 + 16;   // row shift of 16 banks: the transposed operand reads are conflict-free
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), p = lane & 15, q = lane >> 4, NW = blockDim.x >> 6;
  float* xs = sm;                 // h -> h_ln [32][ld]
  float* vs = xs + 32 * ld;       // v_att     [32][ld]
  float* ds = vs + 32 * ld;       // dh'       [32][ld]
  float* dqs = ds + 32 * ld;      // dQKV      [32][ld3]
  const int nkt = (Dh + 15) / 16, nct = (D3 + 15) / 16;  // <= 4 x 12
  const int kt = wave >> 1, ct0 = (wave & 1) * 6, co0 = (wave & 1) * 2;
  v4f accW[6], accO[2];
#pragma unroll
  for (int j = 0; j < 6; ++j) accW[j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) accO[j] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int rbeg = blockIdx.x * wa.rows_per_wg, rend = min(wa.rows, rbeg + wa.rows_per_wg);
  // per-thread 16-byte units of a 32-row step: one of h / v_att / dh' each, three of dQKV.  The
  // loads of step i+1 are issued before the arithmetic of step i (register prefetch); addresses
  // are clamped so that every load is unconditional, rows past the end are zeroed at the LDS store
  const int w4 = Dh >> 2, w43 = 3 * w4;
  const bool xok = t < 32 * w4;
  const int xr = xok ? t / w4 : 0, xc = xok ? t % w4 : 0;
  int qr[3], qc[3];
  bool qok[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int i = t + 512 * u;
    qok[u] = i < 32 * w43;
    qr[u] = qok[u] ? i / w43 : 0;
    qc[u] = qok[u] ? i % w43 : 0;
  }
  float gam[4], bet[4];   // norm_mha parameters of the lane's LayerNorm columns
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = p + 16 * i;
    gam[i] = c < Dh ? L.nm_g[c] : 0.f;
    bet[i] = c < Dh ? L.nm_b[c] : 0.f;
  }
  // (D64: the LayerNorm runs on the prefetched registers: the thread's four channels 4 xc .. 4 xc + 3 of row xr)
  // (scalar loads: parameter views need not be 16-byte aligned)
  float4 gam4 = make_float4(0.f, 0.f, 0.f, 0.f), bet4 = gam4;
  if (D64) {
    gam4 = make_float4(L.nm_g[xc * 4], L.nm_g[xc * 4 + 1], L.nm_g[xc * 4 + 2], L.nm_g[xc * 4 + 3]);
    bet4 = make_float4(L.nm_b[xc * 4], L.nm_b[xc * 4 + 1], L.nm_b[xc * 4 + 2], L.nm_b[xc * 4 + 3]);
  }
  float4 ph, pv, pd, pq[3];
#define WG_LOAD(R0)                                                                              \
  do {                                                                                           \
    const size_t rx = (size_t)min((R0) + xr, wa.rows - 1);                                       \
    ph = *reinterpret_cast<const float4*>(L.h + rx * Dh + xc * 4);                               \
    pv = *reinterpret_cast<const float4*>(L.v_att + rx * Dh + xc * 4);                           \
    pd = *reinterpret_cast<const float4*>(L.dh_out + rx * Dh + xc * 4);                          \
    _Pragma("unroll") for (int u = 0; u < 3; ++u)                                                \
      pq[u] = *reinterpret_cast<const float4*>(L.dqkv + (size_t)min((R0) + qr[u], wa.rows - 1) * D3 + qc[u] * 4); \
  } while (0)
  WG_LOAD(rbeg);
  for (int r0 = rbeg; r0 < rend; r0 += 32) {
    const int nr = min(32, rend - r0), nrp = 32;
    __syncthreads();
    {
      // component-wise selects (a float4 ternary is lowered through scratch memory)
#define SEL4(ok, v) make_float4((ok) ? (v).x : 0.f, (ok) ? (v).y : 0.f, (ok) ? (v).z : 0.f, (ok) ? (v).w : 0.f)
      if (xok) {
        const bool ok = xr < nr;
        if (D64) {
          // norm_mha of the row from the prefetched registers: 16 consecutive lanes hold one row (4 channels each), so the two
          // row reductions are 16-lane DPP sums and the LDS tile receives h_ln directly -- no in-place LayerNorm pass over the
          // tile, one barrier less per 32-row step
          const float mu = sum16((ph.x + ph.y) + (ph.z + ph.w)) * (1.0f / 64);
          const float4 v = make_float4(ph.x - mu, ph.y - mu, ph.z - mu, ph.w - mu);
          const float rstd = rsqrtf(sum16(fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)))) * (1.0f / 64) + wa.ln_eps);
          ph = make_float4(fmaf(v.x * rstd, gam4.x, bet4.x), fmaf(v.y * rstd, gam4.y, bet4.y),
                           fmaf(v.z * rstd, gam4.z, bet4.z), fmaf(v.w * rstd, gam4.w, bet4.w));
        }
        *reinterpret_cast<float4*>(xs + xr * ld + xc * 4) = SEL4(ok, ph);
        *reinterpret_cast<float4*>(vs + xr * ld + xc * 4) = SEL4(ok, pv);
        *reinterpret_cast<float4*>(ds + xr * ld + xc * 4) = SEL4(ok, pd);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (qok[u]) *reinterpret_cast<float4*>(dqs + qr[u] * ld3 + qc[u] * 4) = SEL4(qr[u] < nr, pq[u]);
#undef SEL4
    }
    if (r0 + 32 < rend) WG_LOAD(r0 + 32);
    __syncthreads();
    if (!D64)
    for (int rb = wave * 4; rb < nrp; rb += 4 * NW) {  // norm_mha forward, in place
      float* xr = xs + (rb + q) * ld;
      float v[4], s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; v[i] = c < Dh ? xr[c] : 0.f; s1 += v[i]; }
      const float mu = sum16(s1) / Dh;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) { v[i] -= mu; ss = fmaf(v[i], v[i], ss); } }
      const float rstd = rsqrtf(sum16(ss) / Dh + wa.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int c = p + 16 * i; if (c < Dh) xr[c] = fmaf(v[i] * rstd, gam[i], bet[i]); }
    }
    if (!D64) __syncthreads();
    // wave -> (k tile kt = wave/2, half of the column tiles): the h_ln / v_att operand is read
    // once per row step and feeds all of the wave's output tiles
    if (D64 || kt < nkt) {
#pragma unroll
      for (int rr = 0; rr < 32; rr += 4) {
        const float av = xs[(rr + q) * ld + kt * 16 + p];
        const float vv = vs[(rr + q) * ld + kt * 16 + p];
        const float* dr = dqs + (rr + q) * ld3 + p;
        const float* dd = ds + (rr + q) * ld + p;
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (D64 || ct0 + j < nct) accW[j] = MFMA(av, dr[(ct0 + j) * 16], accW[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (D64 || co0 + j < nkt) accO[j] = MFMA(vv, dd[(co0 + j) * 16], accO[j]);
      }
    }
  }
  float* part = L.part + (size_t)blockIdx.x * (Dh * D3 + Dh * Dh);
  if (kt < nkt) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int c = (ct0 + j) * 16 + p;
      if (ct0 + j < nct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = kt * 16 + 4 * q + r;
          if (kk < Dh && c < D3) part[kk * D3 + c] = accW[j][r];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (co0 + j) * 16 + p;
      if (co0 + j < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kt * 16 + 4 * q + r;
          if (i < Dh && c < Dh) part[Dh * D3 + i * Dh + c] = accO[j][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------ final reduce -----
struct SumSeg { const float* src; float* dst; int n, np, stride, blk0; };   // blk0: first workgroup of the segment in the flat grid
#define SUM_MAX_SEG 77  // 11 layers x 7 segments: fits the 4 KiB kernel-argument block
struct SumArgs { SumSeg seg[SUM_MAX_SEG]; int nseg; };

// 32 outputs per workgroup, the partial axis split over SUM_G groups of 32 lanes, SUM_U loads in flight per lane and ONE wait per
// round of SUM_U -- the last, partial round included (a plain `for (pi ...) v += src[pi]` tail waits on every load).  With hundreds
// of partials per output the loop is latency-bound: what counts is loads in flight per output (64).  SUM_G = 16 (512 threads) measured
// the same at the headline (24.4-24.8 us either way: the kernel moves its ~78 MB at ~4 TB/s) and 1-2 us slower on the small
// launches of configs 1 and 4.  Fixed association order: deterministic.
#define SUM_G 8
#define SUM_U 8
__global__ void __launch_bounds__(32 * SUM_G) k_sum_segments(SumArgs s) {
  __shared__ float red[SUM_G][32];
  // flat grid: one workgroup per 32 outputs of some segment (a (max blocks) x (segments) grid launched
  // four empty workgroups for every working one); the segment is found by a scalar binary search
  int lo = 0, hi = s.nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (s.seg[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SumSeg sg = s.seg[lo];
  const int ol = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int o = ((int)blockIdx.x - sg.blk0) * 32 + ol;
  float v[SUM_U];
#pragma unroll
  for (int k = 0; k < SUM_U; ++k) v[k] = 0.f;
  if (o < sg.n) {
    const float* __restrict__ src = sg.src + o;
    int pi = pg;
    for (; pi + SUM_G * (SUM_U - 1) < sg.np; pi += SUM_G * SUM_U) {
      float w[SUM_U];
#pragma unroll
      for (int k = 0; k < SUM_U; ++k) w[k] = src[(size_t)(pi + SUM_G * k) * sg.stride];
#pragma unroll
      for (int k = 0; k < SUM_U; ++k) v[k] += w[k];
    }
    if (pi < sg.np) {   // the last, partial round: an index past the end re-reads the last partial (a cache hit) and adds nothing
      float w[SUM_U];
#pragma unroll
      for (int k = 0; k < SUM_U; ++k) w[k] = src[(size_t)min(pi + SUM_G * k, sg.np - 1) * sg.stride];
#pragma unroll
      for (int k = 0; k < SUM_U; ++k) v[k] += pi + SUM_G * k < sg.np ? w[k] : 0.f;
    }
  }
  red[pg][ol] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  __syncthreads();
  if (pg == 0 && o < sg.n) {
    float r[SUM_G];
#pragma unroll
    for (int g = 0; g < SUM_G; ++g) r[g] = red[g][ol];
#pragma unroll
    for (int w = 1; w < SUM_G; w *= 2)
#pragma unroll
      for (int g = 0; g < SUM_G; g += 2 * w) r[g] += r[g + w];
    sg.dst[o] = r[0];
  }
}

// T[c][i], s[i], R[c][h|8] -> grads of norm_edge, attention_gates, dense_edge_b, dense_edge_r.
// One workgroup per layer (blockIdx.x), so a whole stack finishes in one launch.  Every input (the reduced sums and the layer's
// parameters) is fetched in ONE round of loads into LDS; the arithmetic then reads LDS only (the straight loops -- load, use,
// store per iteration -- were a chain of ~10 dependent memory round trips for a few KB of work).
struct EdgeGradLayer {
  const float *ne_g, *ne_b, *Wg, *We, *ered;
  float *g_ne_g, *g_ne_b, *g_Wg, *g_bg, *g_We, *g_be, *g_Wr, *g_br;
};
#define EPG_MAX_LAYERS 16
#define EPG_MAX_DE 64
struct EdgeGradArgs { EdgeGradLayer L[EPG_MAX_LAYERS]; int De; uint32_t flags; };

__global__ void __launch_bounds__(256) k_edge_param_grads(EdgeGradArgs ga) {
  constexpr int EPM = 2 * EPG_MAX_DE * 16 + 16, EU = (EPM + 255) / 256, WU = (EPG_MAX_DE * BH + 255) / 256;
  __shared__ float er[EPM], wg[EPG_MAX_DE * BH], we[EPG_MAX_DE * BH], gam[EPG_MAX_DE], bet[EPG_MAX_DE];
  const EdgeGradLayer a = ga.L[blockIdx.x];
  const int DE = ga.De, DEP = ((DE + 15) / 16) * 16, EP = 2 * DEP * 16 + 16, t = threadIdx.x;
  const bool gated = (ga.flags & EGT_BF_GATE) != 0;
  {
    float e[EU], g[WU], w[WU];
#pragma unroll
    for (int u = 0; u < EU; ++u) e[u] = a.ered[min(t + 256 * u, EP - 1)];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = min(t + 256 * u, DE * BH - 1);
      g[u] = gated ? a.Wg[i] : 0.f;
      w[u] = a.We[i];
    }
    const float pg = a.ne_g[min(t, DE - 1)], pb = a.ne_b[min(t, DE - 1)];
#pragma unroll
    for (int u = 0; u < EU; ++u) if (t + 256 * u < EP) er[t + 256 * u] = e[u];
#pragma unroll
    for (int u = 0; u < WU; ++u) if (t + 256 * u < DE * BH) { wg[t + 256 * u] = g[u]; we[t + 256 * u] = w[u]; }
    if (t < DE) { gam[t] = pg; bet[t] = pb; }
  }
  __syncthreads();
  const float* T = er;
  const float* s = er + DEP * 16;
  const float* R = s + 16;
  for (int idx = t; idx < DE * 16; idx += 256) {
    const int c = idx >> 4, i = idx & 15, hd = col_head(i);
    const float v = gam[c] * T[c * 16 + i] + bet[c] * s[i];
    if (col_is_gate(i)) { if (gated) a.g_Wg[c * BH + hd] = v; }
    else a.g_We[c * BH + hd] = v;
  }
  if (t < 16) {
    const int i = t, hd = col_head(i);
    if (col_is_gate(i)) { if (gated) a.g_bg[hd] = s[i]; }
    else a.g_be[hd] = s[i];
  }
  for (int c = t; c < DE; c += 256) {
    float dg = 0.f, db = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int hd = col_head(i);
      const float w = col_is_gate(i) ? wg[c * BH + hd] : we[c * BH + hd];   // (wg holds zeros when the block is not gated)
      dg = fmaf(w, T[c * 16 + i], dg);
      db = fmaf(w, s[i], db);
    }
    a.g_ne_g[c] = dg;
    a.g_ne_b[c] = db;
    a.g_br[c] = R[c * 16 + 8];
  }
  for (int idx = t; idx < BH * DE; idx += 256) {
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
  EGT_MAX_LDS_ONCE(k_node_pre);
  EGT_LAUNCH("k_node_pre", k_node_pre, dim3(a.B * node_chunks(a) + (a.prep ? 1 : 0)), dim3(512), lds, st, a);
}

// layer 0's k_node_pre with the edge-weight preparation of all `n` layers riding along; false: more layers than the argument
// block takes (the caller launches k_edge_prep + k_node_pre instead)
template <int NL>
static void launch_pre_stack(BlockArgs* as, int n, size_t lds, hipStream_t st) {
  PrepArgsN<NL> pa{};
  pa.n = n;
  for (int l = 0; l < n; ++l) {
    const BlockArgs& a = as[l];
    pa.L[l] = PrepLayer{a.ne_g, a.ne_b, a.Wg, a.bg, a.We, a.be, a.pw, a.Wqkv, a.Wo, a.wfrag};
  }
  EGT_MAX_LDS_ONCE(k_node_pre_stack<NL>);
  EGT_LAUNCH("k_node_pre", k_node_pre_stack<NL>, dim3(as[0].B * node_chunks(as[0]) + n), dim3(512), lds, st, as[0], pa);
}
bool egt_node_launch_pre_stack(BlockArgs* as, int n, hipStream_t st) {
  if (n > 40) return false;
  const BlockArgs& a = as[0];
  size_t lds = lds_rows(a.Dh, 1) + (size_t)a.Dh * (3 * a.Dh + LDP) * 4;
  if (lds < 1024) lds = 1024;  // prep workgroup scratch
  if (n <= 16) launch_pre_stack<16>(as, n, lds, st);
  else launch_pre_stack<40>(as, n, lds, st);
  return true;
}

void egt_node_launch_post(BlockArgs& a, hipStream_t st) {
  const size_t lds = lds_rows(a.Dh, 1) + (size_t)a.Dh * (a.Dh + LDP) * 4;
  EGT_LAUNCH("k_node_post", k_node_post, dim3(a.B * node_chunks(a)), dim3(512), lds, st, a);
}

// dv_layer: the layer whose dense_mha is differentiated in this launch (NULL: none).  do_pre == 0
// is the first call of a chain: the dh' rows come from a.dh_out and dv_layer must be &a.
void egt_node_launch_bwd(BlockArgs& a, const BlockArgs* dv_layer, bool do_pre, hipStream_t st) {
  NodeBwdArgs x{};
  auto down16 = [](const float* p) { return reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)15); };
  x.wq_a = down16(a.Wqkv); x.wq_al = x.wq_a == a.Wqkv;
  if (dv_layer) {
    x.wo_a = down16(dv_layer->Wo); x.wo_al = x.wo_a == dv_layer->Wo;
    x.dv_Wo = dv_layer->Wo; x.dv_v_att = dv_layer->v_att; x.dv_dh_src = a.dh_out;
    x.dv_stats = dv_layer->stats; x.dv_dvp = dv_layer->dvp; x.dv_sbo = dv_layer->sbo;
  }
  const int Dh = a.Dh, ld = Dh + LDP, ld3 = 3 * Dh + LDP;
  size_t lds = ((size_t)4 * NODE_RC * ld + NODE_RC + 4 * NODE_RC * 8 + (size_t)Dh * ld) * 4;
  if (do_pre) lds += ((size_t)NODE_RC * ld3 + (size_t)Dh * ld3) * 4;
  const bool prep = !do_pre && a.prep;
  const dim3 grid(a.B * node_chunks(a) + (prep ? 1 : 0));
#define NODE_BWD(PRE_, DV_, NP_, NAME)                                                                     \
  do {                                                                                                 \
    EGT_MAX_LDS_ONCE(k_node_bwd<PRE_, DV_, NP_>); \
    EGT_LAUNCH(NAME, (k_node_bwd<PRE_, DV_, NP_>), grid, dim3(512), lds, st, a, x);                    \
  } while (0)
  const bool np4 = a.NQP == 4 && a.NLR == 4;   // N = 64 with 16-row workgroups
  if (!do_pre) NODE_BWD(false, true, 0, "k_node_bwd_top");
  else if (dv_layer) { if (np4) NODE_BWD(true, true, 4, "k_node_bwd"); else NODE_BWD(true, true, 0, "k_node_bwd"); }
  else { if (np4) NODE_BWD(true, false, 4, "k_node_bwd"); else NODE_BWD(true, false, 0, "k_node_bwd"); }
#undef NODE_BWD
}

// Node rows per workgroup of k_node_wgrads: WG_ROWS, or more (whole 32-row steps) when that keeps the launch's
// (chunks x layers) workgroups within ONE round of the two workgroups per CU its 57 KB of LDS allow -- the headline's ten layers
// of 8 192 rows are 640 workgroups of four steps at 128 rows (a second, quarter-full round) and 430 of six steps at 192.
// `layers` = 1 gives the largest chunk count (what the workspace layout reserves).
static int wgrad_rows_per_wg(int rows, int layers) {
  int per_layer = (2 * egt_device_cus()) / (layers > 0 ? layers : 1);
  if (per_layer < 1) per_layer = 1;
  const int r = (((rows + per_layer - 1) / per_layer + 31) / 32) * 32;
  return r < WG_ROWS ? WG_ROWS : r;
}
int egt_node_wgrad_chunks(int rows, int layers) { const int r = wgrad_rows_per_wg(rows, layers); return (rows + r - 1) / r; }

// deferred GEMM-shaped weight gradients of `n` layers (n <= 64) in one launch
void egt_node_launch_wgrads(BlockArgs* as, int n, hipStream_t st) {
  WGradArgs wa{};
  const BlockArgs& a0 = as[0];
  wa.rows = a0.B * a0.N; wa.Dh = a0.Dh; wa.ln_eps = a0.ln_eps; wa.rows_per_wg = wgrad_rows_per_wg(wa.rows, n);
  for (int l = 0; l < n; ++l) {
    const BlockArgs& a = as[l];
    wa.L[l] = WGradLayer{a.h, a.nm_g, a.nm_b, a.dqkv_sv, a.v_att, a.dh_out, a.wpart};
  }
  const int Dh = a0.Dh;
  const size_t lds = ((size_t)3 * 32 * (Dh + 16) + (size_t)32 * (3 * Dh + 16)) * 4;
  const dim3 grid(egt_node_wgrad_chunks(wa.rows, n), n);
  for (int l = 0; l < n; ++l) as[l].wpart_n = (int)grid.x;   // what the reduction sums (never re-derived from rows / layers there)
  if (Dh == 64) EGT_LAUNCH("k_node_wgrads", k_node_wgrads<true>, grid, dim3(512), lds, st, wa);
  else EGT_LAUNCH("k_node_wgrads", k_node_wgrads<false>, grid, dim3(512), lds, st, wa);
}

void egt_node_launch_prep(BlockArgs* as, int n, hipStream_t st) {
  for (int l0 = 0; l0 < n; l0 += PREP_MAX_LAYERS) {
    const int nl = n - l0 < PREP_MAX_LAYERS ? n - l0 : PREP_MAX_LAYERS;
    PrepArgs pa{};
    pa.De = as[0].De; pa.Dh = as[0].Dh; pa.flags = as[0].flags;
    for (int l = 0; l < nl; ++l) {
      const BlockArgs& a = as[l0 + l];
      pa.L[l] = PrepLayer{a.ne_g, a.ne_b, a.Wg, a.bg, a.We, a.be, a.pw, a.Wqkv, a.Wo, a.wfrag};
    }
    EGT_LAUNCH("k_edge_prep", k_edge_prep, dim3(nl), dim3(256), 0, st, pa);
  }
}

// Reduce the per-workgroup partials of `n` layers (one BlockArgs each, with their own
// spart / sbo / wpart / epart / ered and gradient pointers) and finish the edge-parameter gradients.
void egt_node_launch_reduce(BlockArgs* as, int n, int nwg_bwd, int EP, hipStream_t st) {
  for (int l0 = 0; l0 < n; l0 += 11) {
    const int nl = (n - l0 < 11) ? (n - l0) : 11;
    SumArgs s{};
    int nblk = 0, k = 0;
    auto seg = [&](const float* src, float* dst, int cnt, int npart, int stride) {
      s.seg[k] = SumSeg{src, dst, cnt, npart, stride, nblk};
      nblk += (cnt + 31) / 32;
      ++k;
    };
    for (int l = l0; l < l0 + nl; ++l) {
      BlockArgs& a = as[l];
      const int Dh = a.Dh, D3 = 3 * Dh, SP = D3 + 2 * Dh;
      const int nwc = a.wpart_n, WS = Dh * D3 + Dh * Dh;   // the chunk count egt_node_launch_wgrads stored
      seg(a.wpart, a.g_Wqkv, Dh * D3, nwc, WS);
      seg(a.wpart + Dh * D3, a.g_Wo, Dh * Dh, nwc, WS);
      seg(a.spart, a.g_bqkv, D3, a.spart_n, SP);          // written by k_node_bwd or by the pair kernel's prologue
      seg(a.spart + D3, a.g_nm_g, Dh, a.spart_n, SP);
      seg(a.spart + D3 + Dh, a.g_nm_b, Dh, a.spart_n, SP);
      seg(a.sbo, a.g_bo, Dh, a.sbo_n, Dh);
      seg(a.epart, a.ered, EP, nwg_bwd, EP);
    }
    s.nseg = k;
    EGT_LAUNCH("k_sum_segments", k_sum_segments, dim3(nblk), dim3(32 * SUM_G), 0, st, s);
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
