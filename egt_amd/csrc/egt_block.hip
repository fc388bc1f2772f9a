// Fused EGT attention block for gfx950 — the data-parallel hot path.
//   (h', e') = edge_update_residual(h, e) around mha_block, pre-norm
//   (lib/models/graph_xformer_model_base.py:192-223 + :106-145, inner op
//    lib/models/egt_layers.py:57-143).
//
// Data flow per 16-pair tile (one query row l, 16 consecutive keys m; the
// [B,N,N,De] edge tensor makes that 16*De*4 contiguous bytes):
//   HBM --coalesced 16B loads--> LDS tile (XOR-swizzled 16B slots)
//   LDS --ds_read_b128--> MFMA B-fragments: lane (p = lane&15, q = lane>>4) owns
//        pair p and channels {16*t + 4*q + r}
//   LayerNorm (norm_edge) in registers (two-pass moments; 2 cross-lane adds)
//   v_mfma_f32_16x16x4_f32 x (De/4):  [Wg|We]^T(16 x De) . ehat^T(De x 16 pairs)
//        -> lane (p,q) receives G,E of pair p for heads 2q,2q+1  (no shuffles)
//   QK^T (d <= 8: 16 FMAs/lane), clip, +E, additive masks, per-lane ONLINE
//   softmax x sigmoid gate, A.V accumulated per lane; merged across the 16 key
//   lanes once per query row.
//   v_mfma x (De/16*2): Wr^T . H_hat^T -> residual update written back into the
//   LDS tile, streamed out with coalesced 16B stores.
// E, G, H_hat, A_tild never touch HBM.  Backward recomputes all of it from e,
// keeps per-(row,head) softmax statistics + V_att from the forward (flash-style
// delta), and does every weight gradient as MFMA contractions over the pair
// axis with deterministic per-workgroup partials.
//
// Lane roles follow the 16x16x4 f32 MFMA register maps (A: row=lane&15,
// k=lane>>4; B: k=lane>>4, col=lane&15; D: row=4*(lane>>4)+reg, col=lane&15).
#include "egt_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define BH 8        // heads (all reference configs)
#define QKVP 192    // packed floats per node row: [3][4 head-pairs][8 k][2]
#define NODE_RC 32  // node rows per chunk in the node kernels

struct BlockArgs {
  int B, N, De, DK, Dh;  // DK = per-head dim (<= 8)
  uint32_t flags;
  float clip_lo, clip_hi, scale, ln_eps;
  uint32_t rm_thr, s0, s1;
  int rng_rm;
  int TL, NLR;  // backward: query rows per workgroup, row-ranges per graph
  // params
  const float *ne_g, *ne_b, *Wg, *bg, *We, *be, *nm_g, *nm_b, *Wqkv, *bqkv, *Wo, *bo, *Wr, *br;
  // tensors
  const float *h, *e, *M;
  const uint8_t *km, *rm;
  float *h_out, *e_out;
  // saved (forward -> backward)
  float *v_att, *stats, *qkvp;
  // workspace
  float *pw;        // prepared edge weights: Wp[DEP][16], c2[16]
  float *dvp, *dqp, *dkvp, *epart, *npart, *ered;
  // backward
  const float *dh_out, *de_out;
  float *dh, *de;
  float *g_ne_g, *g_ne_b, *g_Wg, *g_bg, *g_We, *g_be, *g_nm_g, *g_nm_b, *g_Wqkv, *g_bqkv, *g_Wo,
      *g_bo, *g_Wr, *g_br;
};

template <int DE>
struct Geo {
  static constexpr int TILES = (DE + 15) / 16;
  static constexpr int DEP = TILES * 16;
  static constexpr int NSLOT = DE / 4;          // 16-byte slots per pair row
  static constexpr int TILE_FLOATS = 16 * DE;   // one 16-pair tile
  static constexpr int NF4 = 4 * DE;            // float4s per tile
  static constexpr int EP = DEP * 16 + 16 + DEP * 16;  // edge partial: T, s, R
};

// row order of the 16 projection columns: i = 4*q + r ->
//   r=0: gate head 2q, r=1: edge-bias head 2q, r=2: gate head 2q+1, r=3: edge-bias head 2q+1
__host__ __device__ inline int col_is_gate(int i) { return ((i & 1) == 0); }
__host__ __device__ inline int col_head(int i) { return 2 * (i >> 2) + ((i >> 1) & 1); }

template <int DE>
__device__ __forceinline__ int swz(int row) { return DE == 64 ? (row & 15) : 0; }

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------- prep ------
// Wp[c][i] = gamma_c * Wsel[c][head(i)], c2[i] = sum_c beta_c * Wsel[c][head(i)] + bias
template <int DE>
__global__ void __launch_bounds__(256) k_block_prep(BlockArgs a) {
  using G = Geo<DE>;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  for (int idx = threadIdx.x; idx < G::DEP * 16; idx += 256) {
    const int c = idx >> 4, i = idx & 15;
    float v = 0.f;
    if (c < DE) {
      const int hd = col_head(i);
      if (col_is_gate(i)) v = gated ? a.ne_g[c] * a.Wg[c * BH + hd] : 0.f;
      else v = a.ne_g[c] * a.We[c * BH + hd];
    }
    a.pw[idx] = v;
  }
  if (threadIdx.x < 16) {
    const int i = threadIdx.x, hd = col_head(i);
    float v = 0.f;
    if (col_is_gate(i)) {
      if (gated) { v = a.bg[hd]; for (int c = 0; c < DE; ++c) v = fmaf(a.ne_b[c], a.Wg[c * BH + hd], v); }
    } else {
      v = a.be[hd];
      for (int c = 0; c < DE; ++c) v = fmaf(a.ne_b[c], a.We[c * BH + hd], v);
    }
    a.pw[G::DEP * 16 + i] = v;
  }
}

// --------------------------------------------------------------- node: pre -----
// norm_mha -> dense_qkv, written in the packed head-pair layout
// qkvp[row][s][q][k][j] = QKV[row][s*Dh + k*8 + 2q + j] (zero for k >= DK)
__global__ void __launch_bounds__(256) k_node_pre(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = a.Dh, N = a.N, b = blockIdx.x, t = threadIdx.x;
  float* xs = sm;  // [NODE_RC][Dh+1]
  for (int r0 = 0; r0 < N; r0 += NODE_RC) {
    const int nr = min(NODE_RC, N - r0);
    __syncthreads();
    for (int i = t; i < nr * Dh; i += 256) {
      const int r = i / Dh, c = i % Dh;
      xs[r * (Dh + 1) + c] = a.h[((size_t)b * N + r0 + r) * Dh + c];
    }
    __syncthreads();
    if (t < nr) {
      float* x = xs + t * (Dh + 1);
      float mu = 0.f;
      for (int c = 0; c < Dh; ++c) mu += x[c];
      mu /= Dh;
      float var = 0.f;
      for (int c = 0; c < Dh; ++c) { const float dlt = x[c] - mu; var = fmaf(dlt, dlt, var); }
      var /= Dh;
      const float rstd = rsqrtf(var + a.ln_eps);
      for (int c = 0; c < Dh; ++c) x[c] = fmaf((x[c] - mu) * rstd, a.nm_g[c], a.nm_b[c]);
    }
    __syncthreads();
    if (t < QKVP) {
      const int s = t / 64, qq = (t >> 4) & 3, k = (t >> 1) & 7, j = t & 1;
      const bool live = k < a.DK;
      const int c = s * Dh + k * 8 + 2 * qq + j;
      for (int rb = 0; rb < nr; rb += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = live ? a.bqkv[c] : 0.f;
        if (live) {
          for (int kk = 0; kk < Dh; ++kk) {
            const float w = a.Wqkv[(size_t)kk * 3 * Dh + c];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(xs[(rb + i) * (Dh + 1) + kk], w, acc[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (rb + i < nr) a.qkvp[((size_t)b * N + r0 + rb + i) * QKVP + t] = acc[i];
      }
    }
  }
}

// -------------------------------------------------------------- node: post -----
// dense_mha + res_mha: h' = h + V_att.Wo + bo
__global__ void __launch_bounds__(256) k_node_post(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = a.Dh, N = a.N, b = blockIdx.x, t = threadIdx.x;
  float* xs = sm;  // v_att rows [NODE_RC][Dh+1]
  const int c = t % Dh, rs = t / Dh, RS = 256 / Dh;
  for (int r0 = 0; r0 < N; r0 += NODE_RC) {
    const int nr = min(NODE_RC, N - r0);
    __syncthreads();
    for (int i = t; i < nr * Dh; i += 256) {
      const int r = i / Dh, cc = i % Dh;
      xs[r * (Dh + 1) + cc] = a.v_att[((size_t)b * N + r0 + r) * Dh + cc];
    }
    __syncthreads();
    if (rs < RS) {
      for (int r = rs; r < nr; r += RS) {
        float acc = a.bo[c];
        for (int i = 0; i < Dh; ++i) acc = fmaf(xs[r * (Dh + 1) + i], a.Wo[i * Dh + c], acc);
        const size_t o = ((size_t)b * N + r0 + r) * Dh + c;
        a.h_out[o] = acc + a.h[o];
      }
    }
  }
}

// ------------------------------------------------------------ tile helpers -----
template <int DE>
__device__ __forceinline__ void tile_to_lds(float* tl, const float* src, int lane, int rows_valid) {
  using G = Geo<DE>;
  // src: 16 contiguous pair rows; lane f covers float4 #f of the tile
#pragma unroll
  for (int f0 = 0; f0 < G::NF4; f0 += 64) {
    const int f = f0 + lane;
    if (G::NF4 >= 64 || f < G::NF4) {
      const int row = f / G::NSLOT, slot = f % G::NSLOT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows_valid) v = *reinterpret_cast<const float4*>(src + (size_t)f * 4);
      *reinterpret_cast<float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2)) = v;
    }
  }
}

template <int DE>
__device__ __forceinline__ void tile_from_lds(const float* tl, float* dst, int lane, int rows_valid) {
  using G = Geo<DE>;
#pragma unroll
  for (int f0 = 0; f0 < G::NF4; f0 += 64) {
    const int f = f0 + lane;
    if (G::NF4 >= 64 || f < G::NF4) {
      const int row = f / G::NSLOT, slot = f % G::NSLOT;
      if (row < rows_valid)
        *reinterpret_cast<float4*>(dst + (size_t)f * 4) =
            *reinterpret_cast<const float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2));
    }
  }
}

// fragment (p,q): channels 16*t + 4*q + {0..3}
template <int DE>
__device__ __forceinline__ float4 frag_read(const float* tl, int p, int q, int t) {
  if (16 * t + 4 * q < DE)
    return *reinterpret_cast<const float4*>(tl + p * DE + (((4 * t + q) ^ swz<DE>(p)) << 2));
  return make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int DE>
__device__ __forceinline__ void frag_write(float* tl, int p, int q, int t, float4 v) {
  if (16 * t + 4 * q < DE)
    *reinterpret_cast<float4*>(tl + p * DE + (((4 * t + q) ^ swz<DE>(p)) << 2)) = v;
}
// scalar element [row][c] of a swizzled tile
template <int DE>
__device__ __forceinline__ float elem_read(const float* tl, int row, int c) {
  return tl[row * DE + ((((c >> 2) ^ swz<DE>(row)) << 2) | (c & 3))];
}

__device__ __forceinline__ float sum_over_q(float v) {  // lanes p, p+16, p+32, p+48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// LayerNorm of the lane's fragments (two-pass moments like tf.nn.moments); returns rstd
template <int DE>
__device__ __forceinline__ float ln_frags(float4 (&x)[Geo<DE>::TILES], int q, float eps) {
  using G = Geo<DE>;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) s += (x[t].x + x[t].y) + (x[t].z + x[t].w);
  const float mu = sum_over_q(s) * (1.0f / DE);
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    if (16 * t + 4 * q < DE) {
      x[t].x -= mu; x[t].y -= mu; x[t].z -= mu; x[t].w -= mu;
      v = fmaf(x[t].x, x[t].x, v); v = fmaf(x[t].y, x[t].y, v);
      v = fmaf(x[t].z, x[t].z, v); v = fmaf(x[t].w, x[t].w, v);
    }
  }
  const float rstd = rsqrtf(sum_over_q(v) * (1.0f / DE) + eps);
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { x[t].x *= rstd; x[t].y *= rstd; x[t].z *= rstd; x[t].w *= rstd; }
  return rstd;
}

// 16 projection columns of the lane's pair: acc[r] = column 4q+r
template <int DE>
__device__ __forceinline__ v4f project(const float4 (&x)[Geo<DE>::TILES],
                                       const float (&wA)[4 * Geo<DE>::TILES], v4f acc) {
  using G = Geo<DE>;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    acc = MFMA(wA[4 * t + 0], x[t].x, acc);
    acc = MFMA(wA[4 * t + 1], x[t].y, acc);
    acc = MFMA(wA[4 * t + 2], x[t].z, acc);
    acc = MFMA(wA[4 * t + 3], x[t].w, acc);
  }
  return acc;
}

struct PairMask {
  float add0, add1;  // total additive mask for heads 2q, 2q+1 (applied stepwise below)
};

// logits x and gate-logits gl for the lane's pair, heads 2q+j; masks ADDED in the
// reference's order (egt_layers.py:91-108)
__device__ __forceinline__ void apply_masks(const BlockArgs& a, bool key_ok, bool valid, size_t idx8,
                                            int q, float (&x)[2], float (&gl)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float xv = x[j], gv = gl[j];
    if (a.km) { const float mk = key_ok ? 0.0f : -EGT_NEG; xv += mk; gv += mk; }
    if (a.M) {
      const float mm = valid ? (a.M[idx8 + 2 * q + j] - 1.0f) * EGT_NEG : 0.0f;
      xv += mm; gv += mm;
    }
    if (a.rm || a.rng_rm) {
      bool hit = false;
      if (valid) {
        if (a.rm) hit = a.rm[idx8 + 2 * q + j] != 0;
        else hit = (egt_hash32((uint32_t)(idx8 + 2 * q + j), a.s0, a.s1) >> 8) < a.rm_thr;
      }
      const float mr = hit ? -EGT_NEG : 0.0f;
      xv += mr; gv += mr;
    }
    x[j] = xv; gl[j] = gv;
  }
}

// ================================================================= forward =====
template <int DE>
__global__ void __launch_bounds__(256, 2) k_block_fwd(BlockArgs a) {
  using G = Geo<DE>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N;
  const int lgroups = (N + 15) / 16;
  const int b = blockIdx.x / lgroups, lg = blockIdx.x % lgroups;
  float* tl = sm + wave * G::TILE_FLOATS;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;

  // lane-constant MFMA operands
  float wA[4 * G::TILES], wrA[G::TILES][2], c2r[4];
  float4 brv[G::TILES];
#pragma unroll
  for (int t = 0; t < 4 * G::TILES; ++t) wA[t] = a.pw[(16 * (t >> 2) + 4 * q + (t & 3)) * 16 + p];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    const int c = 16 * t + p;
    wrA[t][0] = c < DE ? a.Wr[(2 * q + 0) * DE + c] : 0.f;
    wrA[t][1] = c < DE ? a.Wr[(2 * q + 1) * DE + c] : 0.f;
    brv[t] = (16 * t + 4 * q < DE) ? *reinterpret_cast<const float4*>(a.br + 16 * t + 4 * q)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (int li = 0; li < 4; ++li) {
    const int l = lg * 16 + wave + 4 * li;
    if (l >= N) break;
    const size_t rowl = (size_t)b * N + l;
    float Qf[16];
    {
      const float4* qp = reinterpret_cast<const float4*>(a.qkvp + (rowl * 3 + 0) * 64 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = qp[i]; Qf[4*i] = v.x; Qf[4*i+1] = v.y; Qf[4*i+2] = v.z; Qf[4*i+3] = v.w; }
    }
    float mx[2] = {-3.0e38f, -3.0e38f}, sum[2] = {0.f, 0.f}, O[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) O[i] = 0.f;

    for (int m0 = 0; m0 < N; m0 += 16) {
      const int m = m0 + p;
      const bool valid = m < N;
      const int rows_valid = min(16, N - m0);
      const size_t pair0 = rowl * N + m0;
      // ---- K/V fragments + edge tile ----
      float Kf[16], Vf[16];
      {
        const size_t rowm = (size_t)b * N + (valid ? m : 0);
        const float4* kp = reinterpret_cast<const float4*>(a.qkvp + (rowm * 3 + 1) * 64 + q * 16);
        const float4* vp = reinterpret_cast<const float4*>(a.qkvp + (rowm * 3 + 2) * 64 + q * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 kv = kp[i], vv = vp[i];
          Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
          Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
        }
      }
      wave_lds_fence();
      tile_to_lds<DE>(tl, a.e + pair0 * DE, lane, rows_valid);
      wave_lds_fence();
      float4 x[G::TILES];
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(tl, p, q, t);
      // ---- norm_edge + [attention_gates | dense_edge_b] ----
      ln_frags<DE>(x, q, a.ln_eps);
      v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
      acc = project<DE>(x, wA, acc);
      // ---- scaled QK^T, clip, + E (egt_layers.py:79-86) ----
      float hh[2], xl[2], gl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
        float ah = dot * a.scale;
        if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
        hh[j] = ah + acc[2 * j + 1];
        xl[j] = hh[j];
        gl[j] = acc[2 * j];
      }
      const bool key_ok = (a.km && valid) ? (a.km[(size_t)b * N + m] != 0) : true;
      apply_masks(a, key_ok, valid, (pair0 + p) * BH, q, xl, gl);
      // ---- online softmax x gate, A.V (per lane) ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float xv = xl[j];
        const float mn = valid ? fmaxf(mx[j], xv) : mx[j];
        const float alpha = __expf(mx[j] - mn);
        const float pe = valid ? __expf(xv - mn) : 0.f;
        mx[j] = mn;
        sum[j] = fmaf(sum[j], alpha, pe);
        const float av = gated ? pe * egt_sigmoid(gl[j]) : pe;
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(O[2 * k + j], alpha, av * Vf[2 * k + j]);
      }
      // ---- dense_edge_r + res_edge: e' = e + H_hat.Wr + br ----
      const float h0 = valid ? hh[0] : 0.f, h1 = valid ? hh[1] : 0.f;
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        v4f d = {brv[t].x, brv[t].y, brv[t].z, brv[t].w};
        d = MFMA(wrA[t][0], h0, d);
        d = MFMA(wrA[t][1], h1, d);
        const float4 ev = frag_read<DE>(tl, p, q, t);
        frag_write<DE>(tl, p, q, t, make_float4(ev.x + d[0], ev.y + d[1], ev.z + d[2], ev.w + d[3]));
      }
      wave_lds_fence();
      tile_from_lds<DE>(tl, a.e_out + pair0 * DE, lane, rows_valid);
    }

    // ---- merge the 16 key lanes (same q): max, then sums ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float mr = mx[j];
      mr = fmaxf(mr, __shfl_xor(mr, 1, 64)); mr = fmaxf(mr, __shfl_xor(mr, 2, 64));
      mr = fmaxf(mr, __shfl_xor(mr, 4, 64)); mr = fmaxf(mr, __shfl_xor(mr, 8, 64));
      const float f = __expf(mx[j] - mr);
      mx[j] = mr;
      float s = sum[j] * f;
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      sum[j] = s;
#pragma unroll
      for (int k = 0; k < 8; ++k) O[2 * k + j] *= f;
    }
    // transpose-reduce O over the 16 key lanes: lane p ends with element i = p
    {
      float w8[8], w4[4], w2[2];
      const bool b3 = (p & 8) != 0, b2 = (p & 4) != 0, b1 = (p & 2) != 0, b0 = (p & 1) != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float snd = b3 ? O[i] : O[i + 8], kp = b3 ? O[i + 8] : O[i];
        w8[i] = kp + __shfl_xor(snd, 8, 64);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float snd = b2 ? w8[i] : w8[i + 4], kp = b2 ? w8[i + 4] : w8[i];
        w4[i] = kp + __shfl_xor(snd, 4, 64);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float snd = b1 ? w4[i] : w4[i + 2], kp = b1 ? w4[i + 2] : w4[i];
        w2[i] = kp + __shfl_xor(snd, 2, 64);
      }
      const float snd = b0 ? w2[0] : w2[1], kp = b0 ? w2[1] : w2[0];
      const float o = kp + __shfl_xor(snd, 1, 64);
      // element i = p: k = p>>1, j = p&1 -> head 2q + j
      const int k = p >> 1, j = p & 1;
      const float sj = j ? sum[1] : sum[0];
      if (k < a.DK) a.v_att[rowl * a.Dh + k * BH + 2 * q + j] = o / sj;
      if (p < 2) {
        float* st = a.stats + (rowl * BH + 2 * q + p) * 4;
        st[0] = p ? mx[1] : mx[0];
        st[1] = sj;
      }
    }
  }
}

// ====================================================== node: post backward ====
// dV_att = dh'.Wo^T (packed), delta[l,h] = sum_k dV_att*V_att, partial dWo, dbo
__global__ void __launch_bounds__(256) k_node_post_bwd(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = a.Dh, N = a.N, b = blockIdx.x, t = threadIdx.x;
  float* ds = sm;                           // dh' rows   [RC][Dh+1]
  float* vs = ds + NODE_RC * (Dh + 1);      // v_att rows [RC][Dh+1]
  float* wl = vs + NODE_RC * (Dh + 1);      // Wo         [Dh][Dh+1]
  float* pr = wl + Dh * (Dh + 1);           // dv*v       [RC][64+1]
  for (int i = t; i < Dh * Dh; i += 256) wl[(i / Dh) * (Dh + 1) + (i % Dh)] = a.Wo[i];
  const int c = t % Dh, ib = t / Dh, IB = 256 / Dh;  // weight-grad role
  const int IPT = (Dh + IB - 1) / IB;                 // i's per thread
  float accW[16];
  float accB = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) accW[i] = 0.f;
  for (int r0 = 0; r0 < N; r0 += NODE_RC) {
    const int nr = min(NODE_RC, N - r0);
    __syncthreads();
    for (int i = t; i < nr * Dh; i += 256) {
      const int r = i / Dh, cc = i % Dh;
      const size_t o = ((size_t)b * N + r0 + r) * Dh + cc;
      ds[r * (Dh + 1) + cc] = a.dh_out[o];
      vs[r * (Dh + 1) + cc] = a.v_att[o];
    }
    __syncthreads();
    // packed dV: thread <-> packed position (64 per row), rows strided by 4
    {
      const int pos = t & 63, rsub = t >> 6;
      const int qq = pos >> 4, k = (pos >> 1) & 7, j = pos & 1;
      const bool live = k < a.DK;
      const int i = k * 8 + 2 * qq + j;
      for (int r = rsub; r < nr; r += 4) {
        float dv = 0.f;
        if (live)
          for (int cc = 0; cc < Dh; ++cc) dv = fmaf(ds[r * (Dh + 1) + cc], wl[i * (Dh + 1) + cc], dv);
        a.dvp[((size_t)b * N + r0 + r) * 64 + pos] = dv;
        pr[r * 65 + pos] = live ? dv * vs[r * (Dh + 1) + i] : 0.f;
      }
    }
    __syncthreads();
    if (t < nr * BH) {
      const int r = t / BH, hd = t % BH;
      const int qq = hd >> 1, j = hd & 1;
      float dl = 0.f;
      for (int k = 0; k < 8; ++k) dl += pr[r * 65 + qq * 16 + k * 2 + j];
      a.stats[(((size_t)b * N + r0 + r) * BH + hd) * 4 + 2] = dl;
    }
    // dWo[i][c] += v_att[r][i] * dh'[r][c]
    if (ib < IB) {
      for (int r = 0; r < nr; ++r) {
        const float dv = ds[r * (Dh + 1) + c];
        if (ib == 0) accB += dv;
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
          const int i = ib * IPT + ii;
          if (ii < IPT && i < Dh) accW[ii] = fmaf(vs[r * (Dh + 1) + i], dv, accW[ii]);
        }
      }
    }
  }
  // node partial layout per graph: [dWqkv Dh*3Dh | dbqkv 3Dh | dgamma Dh | dbeta Dh | dWo Dh*Dh | dbo Dh]
  float* part = a.npart + (size_t)b * (Dh * 3 * Dh + 3 * Dh + 2 * Dh + Dh * Dh + Dh) + Dh * 3 * Dh + 3 * Dh + 2 * Dh;
  if (ib < IB) {
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
      const int i = ib * IPT + ii;
      if (ii < IPT && i < Dh) part[i * Dh + c] = accW[ii];
    }
    if (ib == 0) part[Dh * Dh + c] = accB;
  }
}

// ================================================================ backward =====
template <int DE>
__global__ void __launch_bounds__(256, 1) k_block_bwd(BlockArgs a) {
  using G = Geo<DE>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;
  const int b = blockIdx.x / a.NLR, lr = blockIdx.x % a.NLR;
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL);
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  // LDS carve: per wave [e tile | de' tile | dGE scratch 16x16 | H_hat scratch 16x12], then dq[4][TL][64]
  constexpr int PW = 2 * G::TILE_FLOATS + 256 + 192;
  float* et = sm + wave * PW;
  float* dt = et + G::TILE_FLOATS;
  float* sc1 = dt + G::TILE_FLOATS;
  float* sc2 = sc1 + 256;
  float* dql = sm + 4 * PW + wave * TL * 64;
  for (int i = lane; i < TL * 64; i += 64) dql[i] = 0.f;

  // lane-constant MFMA operands
  float wA[4 * G::TILES], wrB[4 * G::TILES], wD[G::TILES][4], c2r[4];
#pragma unroll
  for (int t = 0; t < 4 * G::TILES; ++t) {
    const int c = 16 * (t >> 2) + 4 * q + (t & 3);
    wA[t] = a.pw[c * 16 + p];
    // dH_ext rows: i = 4q'+0 -> head 2q', i = 4q'+1 -> head 2q'+1, rows 4q'+2,3 empty
    const int hd = 2 * (p >> 2) + (p & 1);
    wrB[t] = ((p & 2) == 0 && c < DE) ? a.Wr[hd * DE + c] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < G::TILES; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) wD[t][s] = a.pw[(16 * t + p) * 16 + 4 * q + s];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];

  v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f};

  const int ntile = (N + 15) / 16;
  for (int mt = wave; mt < ntile; mt += 4) {
    const int m0 = mt * 16, m = m0 + p;
    const bool valid = m < N;
    const int rows_valid = min(16, N - m0);
    float Kf[16], Vf[16], dKa[16], dVa[16];
    {
      const size_t rowm = (size_t)b * N + (valid ? m : 0);
      const float4* kp = reinterpret_cast<const float4*>(a.qkvp + (rowm * 3 + 1) * 64 + q * 16);
      const float4* vp = reinterpret_cast<const float4*>(a.qkvp + (rowm * 3 + 2) * 64 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i], vv = vp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
        Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
    }
    const bool key_ok = (a.km && valid) ? (a.km[(size_t)b * N + m] != 0) : true;

    for (int l = l_begin; l < l_end; ++l) {
      const size_t rowl = (size_t)b * N + l;
      const size_t pair0 = rowl * N + m0;
      float Qf[16], dVf[16], st[8];
      {
        const float4* qp = reinterpret_cast<const float4*>(a.qkvp + (rowl * 3 + 0) * 64 + q * 16);
        const float4* dp = reinterpret_cast<const float4*>(a.dvp + rowl * 64 + q * 16);
        const float4* sp = reinterpret_cast<const float4*>(a.stats + (rowl * BH + 2 * q) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 u = qp[i], v = dp[i];
          Qf[4*i] = u.x; Qf[4*i+1] = u.y; Qf[4*i+2] = u.z; Qf[4*i+3] = u.w;
          dVf[4*i] = v.x; dVf[4*i+1] = v.y; dVf[4*i+2] = v.z; dVf[4*i+3] = v.w;
        }
        const float4 s0 = sp[0], s1 = sp[1];
        st[0] = s0.x; st[1] = s0.y; st[2] = s0.z; st[4] = s1.x; st[5] = s1.y; st[6] = s1.z;
      }
      wave_lds_fence();
      tile_to_lds<DE>(et, a.e + pair0 * DE, lane, rows_valid);
      tile_to_lds<DE>(dt, a.de_out + pair0 * DE, lane, rows_valid);
      wave_lds_fence();
      float4 x[G::TILES], dy[G::TILES];
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) { x[t] = frag_read<DE>(et, p, q, t); dy[t] = frag_read<DE>(dt, p, q, t); }
      const float rstd = ln_frags<DE>(x, q, a.ln_eps);
      // xhat back into the tile: the weight-gradient MFMAs read it pair-major
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) frag_write<DE>(et, p, q, t, x[t]);
      v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
      acc = project<DE>(x, wA, acc);
      // dH_ext = de'.Wr^T (rows 4q, 4q+1 of D = heads 2q, 2q+1)
      v4f dhx = {0.f, 0.f, 0.f, 0.f};
      dhx = project<DE>(dy, wrB, dhx);

      float hh[2], xl[2], gl[2], inr[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
        const float araw = dot * a.scale;
        float ah = araw;
        inr[j] = 1.0f;
        if (clip) {
          inr[j] = (araw >= a.clip_lo && araw <= a.clip_hi) ? 1.0f : 0.0f;
          ah = fminf(fmaxf(araw, a.clip_lo), a.clip_hi);
        }
        hh[j] = ah + acc[2 * j + 1];
        xl[j] = hh[j];
        gl[j] = acc[2 * j];
      }
      apply_masks(a, key_ok, valid, (pair0 + p) * BH, q, xl, gl);
      float dge[4], dq[16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float S = valid ? __expf(xl[j] - st[4 * j]) / st[4 * j + 1] : 0.f;
        const float g = gated ? egt_sigmoid(gl[j]) : 1.0f;
        float dAd = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dAd = fmaf(dVf[2 * k + j], Vf[2 * k + j], dAd);
        const float dS = dAd * g;
        const float dGl = gated ? dAd * S * g * (1.0f - g) : 0.f;
        float dH = S * (dS - st[4 * j + 2]) + dhx[j];
        if (!valid) dH = 0.f;
        const float dA = dH * inr[j] * a.scale;
        const float at = S * g;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          dq[2 * k + j] = dA * Kf[2 * k + j];
          dKa[2 * k + j] = fmaf(dA, Qf[2 * k + j], dKa[2 * k + j]);
          dVa[2 * k + j] = fmaf(at, dVf[2 * k + j], dVa[2 * k + j]);
        }
        dge[2 * j] = valid ? dGl : 0.f;
        dge[2 * j + 1] = dH;
        if (!valid) hh[j] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[r] += dge[r];
      // scratch: dGE[pair][16] and H_hat[pair][8] (+ validity column) for the pair-contractions
      *reinterpret_cast<float4*>(sc1 + p * 16 + 4 * q) = make_float4(dge[0], dge[1], dge[2], dge[3]);
      *reinterpret_cast<float2*>(sc2 + p * 12 + 2 * q) = make_float2(hh[0], hh[1]);
      if (q == 0) sc2[p * 12 + 8] = valid ? 1.0f : 0.0f;

      // d(ehat) = Wp . dGE  -> same fragment layout as x / dy
      float4 dxh[G::TILES];
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        v4f d = {0.f, 0.f, 0.f, 0.f};
        d = MFMA(wD[t][0], dge[0], d);
        d = MFMA(wD[t][1], dge[1], d);
        d = MFMA(wD[t][2], dge[2], d);
        d = MFMA(wD[t][3], dge[3], d);
        dxh[t] = make_float4(d[0], d[1], d[2], d[3]);
      }
      // LayerNorm backward
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        m1 += (dxh[t].x + dxh[t].y) + (dxh[t].z + dxh[t].w);
        m2 = fmaf(dxh[t].x, x[t].x, m2); m2 = fmaf(dxh[t].y, x[t].y, m2);
        m2 = fmaf(dxh[t].z, x[t].z, m2); m2 = fmaf(dxh[t].w, x[t].w, m2);
      }
      m1 = sum_over_q(m1) * (1.0f / DE);
      m2 = sum_over_q(m2) * (1.0f / DE);

      wave_lds_fence();
      // ---- weight-gradient contractions over the 16 pairs of the tile ----
      float bT[4], bR[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bT[s] = sc1[(q + 4 * s) * 16 + p];
        bR[s] = (p < 9) ? sc2[(q + 4 * s) * 12 + p] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        const int c = 16 * t + p;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float ax = c < DE ? elem_read<DE>(et, q + 4 * s, c) : 0.f;
          const float ad = c < DE ? elem_read<DE>(dt, q + 4 * s, c) : 0.f;
          accT[t] = MFMA(ax, bT[s], accT[t]);
          accR[t] = MFMA(ad, bR[s], accR[t]);
        }
      }
      wave_lds_fence();
      // ---- de = de' + LN_bwd(d ehat), written in place over the de' tile ----
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        float4 o;
        o.x = dy[t].x + rstd * (dxh[t].x - m1 - x[t].x * m2);
        o.y = dy[t].y + rstd * (dxh[t].y - m1 - x[t].y * m2);
        o.z = dy[t].z + rstd * (dxh[t].z - m1 - x[t].z * m2);
        o.w = dy[t].w + rstd * (dxh[t].w - m1 - x[t].w * m2);
        frag_write<DE>(dt, p, q, t, o);
      }
      wave_lds_fence();
      tile_from_lds<DE>(dt, a.de + pair0 * DE, lane, rows_valid);

      // ---- dQ[l] partial over this tile's 16 keys: transpose-reduce, lane p keeps element p ----
      {
        float w8[8], w4[4], w2[2];
        const bool b3 = (p & 8) != 0, b2 = (p & 4) != 0, b1 = (p & 2) != 0, b0 = (p & 1) != 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float snd = b3 ? dq[i] : dq[i + 8], kp = b3 ? dq[i + 8] : dq[i];
          w8[i] = kp + __shfl_xor(snd, 8, 64);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float snd = b2 ? w8[i] : w8[i + 4], kp = b2 ? w8[i + 4] : w8[i];
          w4[i] = kp + __shfl_xor(snd, 4, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float snd = b1 ? w4[i] : w4[i + 2], kp = b1 ? w4[i + 2] : w4[i];
          w2[i] = kp + __shfl_xor(snd, 2, 64);
        }
        const float snd = b0 ? w2[0] : w2[1], kp = b0 ? w2[1] : w2[0];
        dql[(l - l_begin) * 64 + q * 16 + p] += kp + __shfl_xor(snd, 1, 64);
      }
    }
    // dK, dV partial of this (row-range, key tile)
    if (valid) {
      float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
      float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
        vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
      }
    }
  }
  // ---- column sums s[i] over the wave's pairs ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = ssum[r];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    ssum[r] = v;
  }
  __syncthreads();
  // dQ: sum the four waves' slots, write packed
  for (int i = threadIdx.x; i < (l_end - l_begin) * 64; i += 256) {
    const float* d0 = sm + 4 * PW;
    a.dqp[((size_t)b * N + l_begin) * 64 + i] =
        (d0[i] + d0[TL * 64 + i]) + (d0[2 * TL * 64 + i] + d0[3 * TL * 64 + i]);
  }
  // edge-parameter partials: per wave into LDS (reusing the tile area), summed over the 4 waves
  __syncthreads();
  float* ep = sm + wave * G::EP;  // EP floats per wave; 4*EP <= 4*PW holds for all DE (checked on host)
#pragma unroll
  for (int t = 0; t < G::TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ep[(16 * t + 4 * q + r) * 16 + p] = accT[t][r];
      ep[G::DEP * 16 + 16 + (16 * t + 4 * q + r) * 16 + p] = accR[t][r];
    }
  if (p == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ep[G::DEP * 16 + 4 * q + r] = ssum[r];
  }
  __syncthreads();
  float* out = a.epart + (size_t)blockIdx.x * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 256)
    out[i] = (sm[i] + sm[G::EP + i]) + (sm[2 * G::EP + i] + sm[3 * G::EP + i]);
}

// ======================================================= node: pre backward ====
// dQKV (packed dq + summed dK/dV partials) -> d(h_ln) = dQKV.Wqkv^T -> LN bwd -> dh
__global__ void __launch_bounds__(256) k_node_pre_bwd(BlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = a.Dh, N = a.N, b = blockIdx.x, t = threadIdx.x, D3 = 3 * a.Dh;
  float* xs = sm;                              // xhat rows   [RC][Dh+1]
  float* dqs = xs + NODE_RC * (Dh + 1);        // dQKV rows   [RC][D3+1]
  float* dls = dqs + NODE_RC * (D3 + 1);       // d(h_ln)     [RC][Dh+1]
  float* wl = dls + NODE_RC * (Dh + 1);        // Wqkv        [Dh][D3+1]
  float* rsx = wl + Dh * (D3 + 1);             // per-row rstd, m1, m2 [RC][4]
  for (int i = t; i < Dh * D3; i += 256) wl[(i / D3) * (D3 + 1) + (i % D3)] = a.Wqkv[i];
  // weight-grad roles: thread <-> column c of dQKV (t < D3), accumulators over kk blocks
  float accW[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) accW[i] = 0.f;
  float accBq = 0.f, accG = 0.f, accBt = 0.f;
  for (int r0 = 0; r0 < N; r0 += NODE_RC) {
    const int nr = min(NODE_RC, N - r0);
    __syncthreads();
    for (int i = t; i < nr * Dh; i += 256) {
      const int r = i / Dh, cc = i % Dh;
      xs[r * (Dh + 1) + cc] = a.h[((size_t)b * N + r0 + r) * Dh + cc];
    }
    // dQKV rows from the packed buffers
    for (int i = t; i < nr * QKVP; i += 256) {
      const int r = i / QKVP, pos = i % QKVP;
      const int s = pos / 64, qq = (pos >> 4) & 3, k = (pos >> 1) & 7, j = pos & 1;
      if (k < a.DK) {
        const size_t row = (size_t)b * N + r0 + r;
        float v;
        if (s == 0) v = a.dqp[row * 64 + (pos & 63)];
        else {
          v = 0.f;
          for (int lr = 0; lr < a.NLR; ++lr)
            v += a.dkvp[(((((size_t)b * a.NLR + lr) * N + r0 + r) * 2 + (s - 1)) * 4 + qq) * 16 + k * 2 + j];
        }
        dqs[r * (D3 + 1) + s * Dh + k * 8 + 2 * qq + j] = v;
      }
    }
    __syncthreads();
    if (t < nr) {  // LN forward statistics -> xhat in place
      float* x = xs + t * (Dh + 1);
      float mu = 0.f;
      for (int c = 0; c < Dh; ++c) mu += x[c];
      mu /= Dh;
      float var = 0.f;
      for (int c = 0; c < Dh; ++c) { const float dlt = x[c] - mu; var = fmaf(dlt, dlt, var); }
      var /= Dh;
      const float rstd = rsqrtf(var + a.ln_eps);
      for (int c = 0; c < Dh; ++c) x[c] = (x[c] - mu) * rstd;
      rsx[t * 4] = rstd;
    }
    // d(h_ln)[r][kk] = sum_c dQKV[r][c] * Wqkv[kk][c]
    {
      const int kk = t % Dh, rs = t / Dh, RS = 256 / Dh;
      if (rs < RS)
        for (int r = rs; r < nr; r += RS) {
          float v = 0.f;
          for (int c = 0; c < D3; ++c) v = fmaf(dqs[r * (D3 + 1) + c], wl[kk * (D3 + 1) + c], v);
          dls[r * (Dh + 1) + kk] = v;
        }
    }
    __syncthreads();
    if (t < nr) {
      float m1 = 0.f, m2 = 0.f;
      for (int c = 0; c < Dh; ++c) {
        const float dxh = dls[t * (Dh + 1) + c] * a.nm_g[c];
        m1 += dxh;
        m2 = fmaf(dxh, xs[t * (Dh + 1) + c], m2);
      }
      rsx[t * 4 + 1] = m1 / Dh;
      rsx[t * 4 + 2] = m2 / Dh;
    }
    __syncthreads();
    for (int i = t; i < nr * Dh; i += 256) {
      const int r = i / Dh, cc = i % Dh;
      const float dxh = dls[r * (Dh + 1) + cc] * a.nm_g[cc];
      const size_t o = ((size_t)b * N + r0 + r) * Dh + cc;
      a.dh[o] = a.dh_out[o] + rsx[r * 4] * (dxh - rsx[r * 4 + 1] - xs[r * (Dh + 1) + cc] * rsx[r * 4 + 2]);
    }
    // parameter partials
    if (t < D3) {
      for (int r = 0; r < nr; ++r) {
        const float dv = dqs[r * (D3 + 1) + t];
        accBq += dv;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk)
          if (kk < Dh) {
            const float hl = fmaf(xs[r * (Dh + 1) + kk], a.nm_g[kk], a.nm_b[kk]);
            accW[kk] = fmaf(hl, dv, accW[kk]);
          }
      }
    }
    if (t < Dh) {
      for (int r = 0; r < nr; ++r) {
        const float dl = dls[r * (Dh + 1) + t];
        accG = fmaf(dl, xs[r * (Dh + 1) + t], accG);
        accBt += dl;
      }
    }
  }
  float* part = a.npart + (size_t)b * (Dh * D3 + D3 + 2 * Dh + Dh * Dh + Dh);
  if (t < D3) {
#pragma unroll
    for (int kk = 0; kk < 64; ++kk)
      if (kk < Dh) part[kk * D3 + t] = accW[kk];
    part[Dh * D3 + t] = accBq;
  }
  if (t < Dh) {
    part[Dh * D3 + D3 + t] = accG;
    part[Dh * D3 + D3 + Dh + t] = accBt;
  }
}

// ============================================================ final reduce =====
struct SumSeg { const float* src; float* dst; int n, np, stride; };
struct SumArgs { SumSeg seg[8]; int nseg; };

__global__ void __launch_bounds__(256) k_sum_segments(SumArgs s) {
  const SumSeg sg = s.seg[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < sg.n; i += gridDim.x * 256) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int pidx = 0;
    for (; pidx + 3 < sg.np; pidx += 4) {
      v0 += sg.src[(size_t)pidx * sg.stride + i];
      v1 += sg.src[(size_t)(pidx + 1) * sg.stride + i];
      v2 += sg.src[(size_t)(pidx + 2) * sg.stride + i];
      v3 += sg.src[(size_t)(pidx + 3) * sg.stride + i];
    }
    for (; pidx < sg.np; ++pidx) v0 += sg.src[(size_t)pidx * sg.stride + i];
    sg.dst[i] = (v0 + v1) + (v2 + v3);
  }
}

// T[c][i], s[i], R[c][h|8] -> grads of norm_edge, attention_gates, dense_edge_b, dense_edge_r
template <int DE>
__global__ void __launch_bounds__(256) k_edge_param_grads(BlockArgs a) {
  using G = Geo<DE>;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const float* T = a.ered;
  const float* s = a.ered + G::DEP * 16;
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

// ================================================================ host glue ====
#define BWD_TL 16

static int block_check(const egt_block_desc* d, bool report) {
#define BAD(code, ...) do { if (report) egt_set_error(__VA_ARGS__); return (code); } while (0)
  if (!d) BAD(EGT_E_NULL, "desc is NULL");
  if (d->dtype != EGT_F32) BAD(EGT_E_DTYPE, "only EGT_F32 is supported (got %d)", d->dtype);
  if (d->B <= 0 || d->N <= 0) BAD(EGT_E_SHAPE, "B and N must be positive");
  if (d->H != BH) BAD(EGT_E_SHAPE, "fused block is built for num_heads=8 (got %d)", d->H);
  if (d->d < 1 || d->d > 8) BAD(EGT_E_SHAPE, "fused block covers per-head dim <= 8 (got %d)", d->d);
  switch (d->De) {
    case 8: case 16: case 32: case 48: case 64: break;
    default: BAD(EGT_E_SHAPE, "fused block covers edge_width in {8,16,32,48,64} (got %d)", d->De);
  }
  if ((size_t)d->B * d->N * d->N * d->H > 0xFFFFFFFFull) BAD(EGT_E_SHAPE, "B*N*N*H exceeds the 32-bit RNG counter");
  return EGT_OK;
#undef BAD
}

extern "C" int egt_block_supported(const egt_block_desc* d) { return block_check(d, false) == EGT_OK; }

static size_t al(size_t x) { return (x + 63) & ~(size_t)63; }  // in floats

struct BlockLayout {
  size_t v_att, stats, qkvp, saved_total;
  size_t pw, dvp, dqp, dkvp, epart, npart, ered, ws_total;
  int NLR, nwg_bwd, EP, npart_stride;
};

static BlockLayout layout(const egt_block_desc* d) {
  BlockLayout L{};
  const size_t rows = (size_t)d->B * d->N;
  const int Dh = d->d * d->H;
  const int TILES = (d->De + 15) / 16, DEP = TILES * 16;
  L.EP = DEP * 16 + 16 + DEP * 16;
  size_t o = 0;
  L.v_att = o; o += al(rows * Dh);
  L.stats = o; o += al(rows * BH * 4);
  L.qkvp = o; o += al(rows * QKVP);
  L.saved_total = o;
  L.NLR = (d->N + BWD_TL - 1) / BWD_TL;
  L.nwg_bwd = d->B * L.NLR;
  L.npart_stride = Dh * 3 * Dh + 3 * Dh + 2 * Dh + Dh * Dh + Dh;
  o = 0;
  L.pw = o; o += al((size_t)DEP * 16 + 16);
  L.dvp = o; o += al(rows * 64);
  L.dqp = o; o += al(rows * 64);
  L.dkvp = o; o += al((size_t)d->B * L.NLR * d->N * 128);
  L.epart = o; o += al((size_t)L.nwg_bwd * L.EP);
  L.npart = o; o += al((size_t)d->B * L.npart_stride);
  L.ered = o; o += al(L.EP);
  L.ws_total = o;
  return L;
}

extern "C" size_t egt_block_saved_bytes(const egt_block_desc* d) {
  if (block_check(d, false)) return 0;
  return layout(d).saved_total * sizeof(float);
}
extern "C" size_t egt_block_workspace_bytes(const egt_block_desc* d) {
  if (block_check(d, false)) return 0;
  return layout(d).ws_total * sizeof(float);
}

static int fill_block(const egt_block_desc* d, const egt_block_params* p, BlockArgs& a) {
  int rc = block_check(d, true);
  if (rc) return rc;
  if (!p) EGT_FAIL(EGT_E_NULL, "params is NULL");
  const void* const* pp = reinterpret_cast<const void* const*>(p);
  const bool gated = (d->flags & EGT_BF_GATE) != 0;
  for (int i = 0; i < 14; ++i) {
    if (!gated && (i == 2 || i == 3)) continue;
    if (!pp[i]) EGT_FAIL(EGT_E_NULL, "block parameter #%d is NULL", i);
  }
  a = BlockArgs{};
  a.B = d->B; a.N = d->N; a.De = d->De; a.DK = d->d; a.Dh = d->d * d->H;
  a.flags = d->flags;
  a.clip_lo = d->clip_lo; a.clip_hi = d->clip_hi;
  a.scale = 1.0f / sqrtf((float)d->d);
  a.ln_eps = d->ln_eps;
  a.rm_thr = egt_threshold24(d->random_mask_prob);
  a.s0 = (uint32_t)(d->seed & 0xFFFFFFFFull);
  a.s1 = (uint32_t)(d->seed >> 32);
  a.ne_g = (const float*)p->norm_edge_gamma; a.ne_b = (const float*)p->norm_edge_beta;
  a.Wg = (const float*)p->attention_gates_kernel; a.bg = (const float*)p->attention_gates_bias;
  a.We = (const float*)p->dense_edge_b_kernel; a.be = (const float*)p->dense_edge_b_bias;
  a.nm_g = (const float*)p->norm_mha_gamma; a.nm_b = (const float*)p->norm_mha_beta;
  a.Wqkv = (const float*)p->dense_qkv_kernel; a.bqkv = (const float*)p->dense_qkv_bias;
  a.Wo = (const float*)p->dense_mha_kernel; a.bo = (const float*)p->dense_mha_bias;
  a.Wr = (const float*)p->dense_edge_r_kernel; a.br = (const float*)p->dense_edge_r_bias;
  return EGT_OK;
}

static void bind_common(const egt_block_desc* d, BlockArgs& a, const void* h, const void* e,
                        const uint8_t* km, const void* M, const uint8_t* rm, float* saved, float* ws) {
  const BlockLayout L = layout(d);
  a.h = (const float*)h; a.e = (const float*)e; a.km = km;
  a.M = (d->flags & EGT_BF_ATTN_MASK) ? (const float*)M : nullptr;
  a.rm = nullptr; a.rng_rm = 0;
  if ((d->flags & EGT_BF_TRAINING) && d->random_mask_prob > 0.0f) {
    if (rm) a.rm = rm; else a.rng_rm = 1;
  }
  a.v_att = saved + L.v_att; a.stats = saved + L.stats; a.qkvp = saved + L.qkvp;
  a.pw = ws + L.pw; a.dvp = ws + L.dvp; a.dqp = ws + L.dqp; a.dkvp = ws + L.dkvp;
  a.epart = ws + L.epart; a.npart = ws + L.npart; a.ered = ws + L.ered;
  a.TL = BWD_TL; a.NLR = L.NLR;
}

#define DISPATCH_BDE(De, CALL)                        \
  switch (De) {                                       \
    case 8: { constexpr int DE = 8; CALL; } break;    \
    case 16: { constexpr int DE = 16; CALL; } break;  \
    case 32: { constexpr int DE = 32; CALL; } break;  \
    case 48: { constexpr int DE = 48; CALL; } break;  \
    default: { constexpr int DE = 64; CALL; } break;  \
  }

static size_t node_lds_pre(int Dh) { return (size_t)NODE_RC * (Dh + 1) * 4; }
static size_t node_lds_post_bwd(int Dh) {
  return ((size_t)2 * NODE_RC * (Dh + 1) + (size_t)Dh * (Dh + 1) + (size_t)NODE_RC * 65) * 4;
}
static size_t node_lds_pre_bwd(int Dh) {
  return ((size_t)2 * NODE_RC * (Dh + 1) + (size_t)NODE_RC * (3 * Dh + 1) + (size_t)Dh * (3 * Dh + 1) +
          (size_t)NODE_RC * 4) * 4;
}

template <int DE>
static void launch_fwd(BlockArgs& a, hipStream_t st) {
  const int lgroups = (a.N + 15) / 16;
  EGT_LAUNCH("k_block_prep", k_block_prep<DE>, dim3(1), dim3(256), 0, st, a);
  EGT_LAUNCH("k_node_pre", k_node_pre, dim3(a.B), dim3(256), node_lds_pre(a.Dh), st, a);
  const size_t lds = (size_t)4 * Geo<DE>::TILE_FLOATS * 4;
  EGT_LAUNCH("k_block_fwd", k_block_fwd<DE>, dim3(a.B * lgroups), dim3(256), lds, st, a);
  EGT_LAUNCH("k_node_post", k_node_post, dim3(a.B), dim3(256), node_lds_pre(a.Dh), st, a);
}

template <int DE>
static void launch_bwd(BlockArgs& a, const BlockLayout& L, hipStream_t st) {
  using GG = Geo<DE>;
  const int Dh = a.Dh, D3 = 3 * Dh;
  EGT_LAUNCH("k_block_prep", k_block_prep<DE>, dim3(1), dim3(256), 0, st, a);
  (void)hipFuncSetAttribute((const void*)k_node_post_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_node_post_bwd", k_node_post_bwd, dim3(a.B), dim3(256), node_lds_post_bwd(Dh), st, a);
  constexpr int PW = 2 * GG::TILE_FLOATS + 256 + 192;
  static_assert(4 * GG::EP <= 4 * PW, "edge partial staging must fit the LDS tile area");
  const size_t lds = ((size_t)4 * PW + (size_t)4 * BWD_TL * 64) * 4;
  (void)hipFuncSetAttribute((const void*)k_block_bwd<DE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_block_bwd", k_block_bwd<DE>, dim3(L.nwg_bwd), dim3(256), lds, st, a);
  (void)hipFuncSetAttribute((const void*)k_node_pre_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_node_pre_bwd", k_node_pre_bwd, dim3(a.B), dim3(256), node_lds_pre_bwd(Dh), st, a);
  SumArgs s{};
  const float* np = a.npart;
  int o = 0;
  s.seg[0] = SumSeg{np + o, a.g_Wqkv, Dh * D3, a.B, L.npart_stride}; o += Dh * D3;
  s.seg[1] = SumSeg{np + o, a.g_bqkv, D3, a.B, L.npart_stride}; o += D3;
  s.seg[2] = SumSeg{np + o, a.g_nm_g, Dh, a.B, L.npart_stride}; o += Dh;
  s.seg[3] = SumSeg{np + o, a.g_nm_b, Dh, a.B, L.npart_stride}; o += Dh;
  s.seg[4] = SumSeg{np + o, a.g_Wo, Dh * Dh, a.B, L.npart_stride}; o += Dh * Dh;
  s.seg[5] = SumSeg{np + o, a.g_bo, Dh, a.B, L.npart_stride};
  s.seg[6] = SumSeg{a.epart, a.ered, L.EP, L.nwg_bwd, L.EP};
  s.nseg = 7;
  EGT_LAUNCH("k_sum_segments", k_sum_segments, dim3(16, 7), dim3(256), 0, st, s);
  EGT_LAUNCH("k_edge_param_grads", k_edge_param_grads<DE>, dim3(1), dim3(256), 0, st, a);
}

extern "C" int egt_block_fwd(const egt_block_desc* desc, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, const uint8_t* rand_mask, void* h_out,
                             void* e_out, void* saved, void* workspace, void* stream) {
  BlockArgs a;
  int rc = fill_block(desc, params, a);
  if (rc) return rc;
  if (!h || !e || !h_out || !e_out || !saved || !workspace)
    EGT_FAIL(EGT_E_NULL, "h/e/h_out/e_out/saved/workspace is NULL");
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  bind_common(desc, a, h, e, key_mask, attn_mask, rand_mask, (float*)saved, (float*)workspace);
  a.h_out = (float*)h_out; a.e_out = (float*)e_out;
  DISPATCH_BDE(desc->De, launch_fwd<DE>(a, (hipStream_t)stream));
  EGT_HIP_LAUNCH_CHECK("egt_block_fwd");
  return EGT_OK;
}

extern "C" int egt_block_bwd(const egt_block_desc* desc, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, const uint8_t* rand_mask, const void* saved,
                             const void* d_h_out, const void* d_e_out, void* d_h, void* d_e,
                             const egt_block_params* grads, void* workspace, void* stream) {
  BlockArgs a;
  int rc = fill_block(desc, params, a);
  if (rc) return rc;
  if (!h || !e || !saved || !d_h_out || !d_e_out || !d_h || !d_e || !grads || !workspace)
    EGT_FAIL(EGT_E_NULL, "h/e/saved/d_h_out/d_e_out/d_h/d_e/grads/workspace is NULL");
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  const bool gated = (desc->flags & EGT_BF_GATE) != 0;
  {
    const void* const* gp = reinterpret_cast<const void* const*>(grads);
    for (int i = 0; i < 14; ++i) {
      if (!gated && (i == 2 || i == 3)) continue;
      if (!gp[i]) EGT_FAIL(EGT_E_NULL, "gradient pointer #%d is NULL", i);
    }
  }
  bind_common(desc, a, h, e, key_mask, attn_mask, rand_mask, (float*)saved, (float*)workspace);
  a.dh_out = (const float*)d_h_out; a.de_out = (const float*)d_e_out;
  a.dh = (float*)d_h; a.de = (float*)d_e;
  a.g_ne_g = (float*)grads->norm_edge_gamma; a.g_ne_b = (float*)grads->norm_edge_beta;
  a.g_Wg = (float*)grads->attention_gates_kernel; a.g_bg = (float*)grads->attention_gates_bias;
  a.g_We = (float*)grads->dense_edge_b_kernel; a.g_be = (float*)grads->dense_edge_b_bias;
  a.g_nm_g = (float*)grads->norm_mha_gamma; a.g_nm_b = (float*)grads->norm_mha_beta;
  a.g_Wqkv = (float*)grads->dense_qkv_kernel; a.g_bqkv = (float*)grads->dense_qkv_bias;
  a.g_Wo = (float*)grads->dense_mha_kernel; a.g_bo = (float*)grads->dense_mha_bias;
  a.g_Wr = (float*)grads->dense_edge_r_kernel; a.g_br = (float*)grads->dense_edge_r_bias;
  const BlockLayout L = layout(desc);
  DISPATCH_BDE(desc->De, launch_bwd<DE>(a, L, (hipStream_t)stream));
  EGT_HIP_LAUNCH_CHECK("egt_block_bwd");
  return EGT_OK;
}
