// Inner op: EGT([QKV,E,G,M], mask) -> (V_att, H_hat, A_tild) and its backward.
// General-shape kernels (any N, d, power-of-two H <= 64, every operator attribute
// of lib/models/egt_layers.py:5-16).  One 64-lane wavefront owns one query row
// (b,l); lane = (m_sub, h) with h innermost so that the [B,N,N,H] streams
// (E, G, M, H_hat, A_tild and their grads) are read/written as contiguous
// 256-byte wavefront transactions.  The softmax over the key axis is a
// three-pass row softmax staged in LDS (max, sum, normalise) exactly like
// tf.nn.softmax (egt_layers.py:111); masks are ADDED (-1e9) in the reference's
// order so all-masked rows and double masking behave as in fp32 TensorFlow.
//
// The headline (fused, MFMA) path lives in egt_block.hip; this file is the
// feature-complete operator and the building block of the composed block.
#include "egt_common.h"

struct AttnArgs {
  int B, N, H, d, logH;
  uint32_t flags;
  float clip_lo, clip_hi, scale;
  uint32_t rm_thr, dk_thr, s0, s1;
  int rng_rm, rng_dk;  // draw the sample in-kernel
  float dk_scale;      // 1/(1-attn_dropout)
  int nvn;
  int lds_per_wave;    // floats
  const float *qkv, *E, *G, *M;
  const uint8_t *km, *rm, *dk;
  float *v_att, *h_hat, *a_tild, *rowstats;
  // backward
  const float *d_v_att, *d_h_ext, *v_att_in, *rowstats_in;
  float *d_qkv, *d_E, *d_G, *ws_dA, *ws_Ad;
};

// masks are applied in the reference's order: key padding (:91-94), attention
// mask (:96-101), random mask (:103-108).
__device__ __forceinline__ float add_masks(const AttnArgs& a, float v, bool key_ok, float mval,
                                           bool rnd_hit) {
  if (a.km) v += key_ok ? 0.0f : -EGT_NEG;
  if (a.M) v += (mval - 1.0f) * EGT_NEG;
  if (a.rm || a.rng_rm) v += rnd_hit ? -EGT_NEG : 0.0f;
  return v;
}

__device__ __forceinline__ bool rnd_hit_of(const AttnArgs& a, size_t idx) {
  if (a.rm) return a.rm[idx] != 0;
  if (a.rng_rm) return (egt_hash32((uint32_t)idx, a.s0, a.s1) >> 8) < a.rm_thr;
  return false;
}

__device__ __forceinline__ float drop_factor(const AttnArgs& a, size_t idx) {
  if (a.dk) return a.dk[idx] ? a.dk_scale : 0.0f;
  if (a.rng_dk)
    return ((egt_hash32((uint32_t)idx, a.s0, a.s1 ^ EGT_DROPOUT_STREAM) >> 8) >= a.dk_thr) ? a.dk_scale
                                                                                         : 0.0f;
  return 1.0f;
}

__device__ __forceinline__ float degree_scaler(const AttnArgs& a, float deg, int l) {
  if (!(a.flags & EGT_F_SCALE_DEGREE)) return 1.0f;
  if (l < a.nvn) return 1.0f;  // tf.pad(..., constant_values=1), egt_layers.py:131-135
  return (a.flags & EGT_F_SCALER_LINEAR) ? deg : logf(1.0f + deg);
}

template <typename F>
__device__ __forceinline__ float reduce_same_head(float v, int H, F op) {
  for (int off = H; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
  return v;
}

// ------------------------------------------------------------------ forward ---
__global__ void __launch_bounds__(256) k_attn_fwd(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long row = (long)blockIdx.x * wpb + wave;
  const bool active = row < (long)a.B * a.N;
  const int N = a.N, H = a.H, d = a.d, dH = d * H;
  const int b = active ? (int)(row / N) : 0, l = active ? (int)(row % N) : 0;
  const int h = lane & (H - 1), ms = lane >> a.logH, MS = 64 >> a.logH;
  float* xs = smem + (size_t)wave * a.lds_per_wave;
  float* gs = xs + (size_t)N * H;
  float* qs = gs + (size_t)N * H;
  const bool gated = (a.flags & EGT_F_GATE_INPUT) != 0;

  if (active) {
    const float* qrow = a.qkv + ((size_t)b * N + l) * 3 * dH;
    for (int i = lane; i < dH; i += 64) qs[i] = qrow[i];
  }
  __syncthreads();
  if (!active) return;

  const size_t rowbase = ((size_t)b * N + l) * N * H;
  // pass 1: logits (egt_layers.py:79-108), H_hat out, running max
  float mx = -3.0e38f;
  for (int m0 = 0; m0 < N; m0 += MS) {
    const int m = m0 + ms;
    if (m < N) {
      const float* kp = a.qkv + ((size_t)b * N + m) * 3 * dH + dH + h;
      float dot = 0.0f;
      for (int k = 0; k < d; ++k) dot = fmaf(qs[k * H + h], kp[k * H], dot);
      float ah = dot * a.scale;
      if (a.flags & EGT_F_CLIP) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
      const size_t idx = rowbase + (size_t)m * H + h;
      float hh = ah;
      if (a.E) hh += a.E[idx];
      a.h_hat[idx] = hh;
      const bool key_ok = a.km ? (a.km[(size_t)b * N + m] != 0) : true;
      const float mval = a.M ? a.M[idx] : 1.0f;
      const bool hit = rnd_hit_of(a, idx);
      const float x = add_masks(a, hh, key_ok, mval, hit);
      xs[m * H + h] = x;
      if (gated) gs[m * H + h] = add_masks(a, a.G[idx], key_ok, mval, hit);
      mx = fmaxf(mx, x);
    }
  }
  mx = reduce_same_head(mx, H, [](float p, float q) { return fmaxf(p, q); });

  // pass 2: exp / sum, gates (egt_layers.py:111-112)
  float sum = 0.0f, deg = 0.0f;
  for (int m0 = 0; m0 < N; m0 += MS) {
    const int m = m0 + ms;
    if (m < N) {
      const float p = __expf(xs[m * H + h] - mx);
      xs[m * H + h] = p;
      sum += p;
      if (gated) {
        const float g = egt_sigmoid(gs[m * H + h]);
        gs[m * H + h] = g;
        deg += g;
      }
    }
  }
  sum = reduce_same_head(sum, H, [](float p, float q) { return p + q; });
  deg = reduce_same_head(deg, H, [](float p, float q) { return p + q; });
  const float inv = 1.0f / sum;
  const float sc = degree_scaler(a, deg, l);
  if (ms == 0) {
    float* rs = a.rowstats + (((size_t)b * N + l) * H + h) * 4;
    rs[0] = mx; rs[1] = sum; rs[2] = deg; rs[3] = 0.0f;
  }

  // pass 3: A_tild (:113), dropout (:116-117), A·V (:120), degree scaler (:123-136)
  constexpr int KB = 8;
  for (int kb = 0; kb < d; kb += KB) {
    float acc[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) acc[kk] = 0.0f;
    for (int m0 = 0; m0 < N; m0 += MS) {
      const int m = m0 + ms;
      if (m < N) {
        float ad;
        if (kb == 0) {
          const size_t idx = rowbase + (size_t)m * H + h;
          float at = xs[m * H + h] * inv;
          if (gated) at *= gs[m * H + h];
          ad = at * drop_factor(a, idx);
          if (a.a_tild) a.a_tild[idx] = ad;   // the reference REASSIGNS A_tild = dropout(A_tild) (:116-117) and returns that
          xs[m * H + h] = ad;
        } else {
          ad = xs[m * H + h];
        }
        const float* vp = a.qkv + ((size_t)b * N + m) * 3 * dH + 2 * dH + h;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
          if (kb + kk < d) acc[kk] = fmaf(ad, vp[(kb + kk) * H], acc[kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      float v = reduce_same_head(acc[kk], H, [](float p, float q) { return p + q; });
      if (ms == 0 && kb + kk < d) a.v_att[((size_t)b * N + l) * dH + (kb + kk) * H + h] = v * sc;
    }
  }
}

// ----------------------------------------------------------------- backward ---
// Row pass: every per-pair gradient of SURVEY §8(a18).  Uses the identity
// sum_m S*dS = sum_k dO*O (O = pre-scaler output) so the row is a single pass.
__global__ void __launch_bounds__(256) k_attn_bwd_row(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long row = (long)blockIdx.x * wpb + wave;
  const bool active = row < (long)a.B * a.N;
  const int N = a.N, H = a.H, d = a.d, dH = d * H;
  const int b = active ? (int)(row / N) : 0, l = active ? (int)(row % N) : 0;
  const int h = lane & (H - 1), ms = lane >> a.logH, MS = 64 >> a.logH;
  float* qs = smem + (size_t)wave * a.lds_per_wave;  // Q row
  float* dos = qs + dH;                               // dO row (dOut * scaler)
  const bool gated = (a.flags & EGT_F_GATE_INPUT) != 0;

  float mx = 0.f, inv_sum = 0.f, rowdot = 0.f, ddeg = 0.f;
  if (active) {
    const float* rs = a.rowstats_in + (((size_t)b * N + l) * H + h) * 4;
    mx = rs[0];
    inv_sum = 1.0f / rs[1];
    const float deg = rs[2];
    const float sc = degree_scaler(a, deg, l);
    const float inv_sc = (sc != 0.0f) ? 1.0f / sc : 0.0f;
    const float* qrow = a.qkv + ((size_t)b * N + l) * 3 * dH;
    const float* dvo = a.d_v_att + ((size_t)b * N + l) * dH;
    const float* vo = a.v_att_in + ((size_t)b * N + l) * dH;
    float dsc = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float dout = dvo[k * H + h];
      const float opre = vo[k * H + h] * inv_sc;
      dsc = fmaf(dout, opre, dsc);
    }
    rowdot = dsc * sc;  // sum_k (dOut*sc) * O_pre
    if ((a.flags & EGT_F_SCALE_DEGREE) && l >= a.nvn)
      ddeg = (a.flags & EGT_F_SCALER_LINEAR) ? dsc : dsc / (1.0f + deg);
    for (int i = lane; i < dH; i += 64) {
      qs[i] = qrow[i];
      // scaler of head (i % H): every lane needs its own head's sc; recompute
      const float* rsi = a.rowstats_in + (((size_t)b * N + l) * H + (i & (H - 1))) * 4;
      dos[i] = dvo[i] * degree_scaler(a, rsi[2], l);
    }
  }
  __syncthreads();
  if (!active) return;

  const size_t rowbase = ((size_t)b * N + l) * N * H;
  for (int m0 = 0; m0 < N; m0 += MS) {
    const int m = m0 + ms;
    if (m >= N) continue;
    const float* kp = a.qkv + ((size_t)b * N + m) * 3 * dH + dH + h;
    const float* vp = kp + dH;
    float dot = 0.0f, dAd = 0.0f;
    for (int k = 0; k < d; ++k) {
      dot = fmaf(qs[k * H + h], kp[k * H], dot);
      dAd = fmaf(dos[k * H + h], vp[k * H], dAd);
    }
    const float araw = dot * a.scale;
    float ah = araw;
    bool inrange = true;
    if (a.flags & EGT_F_CLIP) {
      inrange = (araw >= a.clip_lo) && (araw <= a.clip_hi);
      ah = fminf(fmaxf(araw, a.clip_lo), a.clip_hi);
    }
    const size_t idx = rowbase + (size_t)m * H + h;
    float hh = ah;
    if (a.E) hh += a.E[idx];
    const bool key_ok = a.km ? (a.km[(size_t)b * N + m] != 0) : true;
    const float mval = a.M ? a.M[idx] : 1.0f;
    const bool hit = rnd_hit_of(a, idx);
    const float x = add_masks(a, hh, key_ok, mval, hit);
    const float S = __expf(x - mx) * inv_sum;
    float g = 1.0f;
    if (gated) g = egt_sigmoid(add_masks(a, a.G[idx], key_ok, mval, hit));
    const float Dk = drop_factor(a, idx);
    const float dAt = dAd * Dk;
    const float dS = dAt * g;
    float dH = S * (dS - rowdot);
    if (a.d_h_ext) dH += a.d_h_ext[idx];
    if (a.d_E) a.d_E[idx] = dH;
    if (gated && a.d_G) {
      const float dg = fmaf(dAt, S, ddeg);
      a.d_G[idx] = dg * g * (1.0f - g);
    }
    a.ws_dA[idx] = inrange ? dH * a.scale : 0.0f;
    a.ws_Ad[idx] = S * g * Dk;
  }
}

// dQ[l,k,h] = sum_m dA[l,m,h] * K[m,k,h]
__global__ void __launch_bounds__(256) k_attn_bwd_dq(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long row = (long)blockIdx.x * wpb + wave;
  if (row >= (long)a.B * a.N) return;
  const int N = a.N, H = a.H, d = a.d, dH = d * H;
  const int b = (int)(row / N), l = (int)(row % N);
  const int h = lane & (H - 1), ms = lane >> a.logH, MS = 64 >> a.logH;
  const size_t rowbase = ((size_t)b * N + l) * N * H;
  constexpr int KB = 8;
  for (int kb = 0; kb < d; kb += KB) {
    float acc[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) acc[kk] = 0.0f;
    for (int m0 = 0; m0 < N; m0 += MS) {
      const int m = m0 + ms;
      if (m < N) {
        const float da = a.ws_dA[rowbase + (size_t)m * H + h];
        const float* kp = a.qkv + ((size_t)b * N + m) * 3 * dH + dH + h;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
          if (kb + kk < d) acc[kk] = fmaf(da, kp[(kb + kk) * H], acc[kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      float v = reduce_same_head(acc[kk], H, [](float p, float q) { return p + q; });
      if (ms == 0 && kb + kk < d) a.d_qkv[((size_t)b * N + l) * 3 * dH + (kb + kk) * H + h] = v;
    }
  }
}

// dK[m,k,h] = sum_l dA[l,m,h] * Q[l,k,h] ; dV[m,k,h] = sum_l A_drop[l,m,h] * dO[l,k,h]
__global__ void __launch_bounds__(256) k_attn_bwd_dkv(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const long col = (long)blockIdx.x * wpb + wave;
  if (col >= (long)a.B * a.N) return;
  const int N = a.N, H = a.H, d = a.d, dH = d * H;
  const int b = (int)(col / N), m = (int)(col % N);
  const int h = lane & (H - 1), ls = lane >> a.logH, LS = 64 >> a.logH;
  constexpr int KB = 8;
  for (int kb = 0; kb < d; kb += KB) {
    float accK[KB], accV[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) accK[kk] = accV[kk] = 0.0f;
    for (int l0 = 0; l0 < N; l0 += LS) {
      const int l = l0 + ls;
      if (l < N) {
        const size_t idx = (((size_t)b * N + l) * N + m) * H + h;
        const float da = a.ws_dA[idx];
        const float ad = a.ws_Ad[idx];
        const float sc = degree_scaler(a, a.rowstats_in[(((size_t)b * N + l) * H + h) * 4 + 2], l);
        const float* qp = a.qkv + ((size_t)b * N + l) * 3 * dH + h;
        const float* dop = a.d_v_att + ((size_t)b * N + l) * dH + h;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
          if (kb + kk < d) {
            accK[kk] = fmaf(da, qp[(kb + kk) * H], accK[kk]);
            accV[kk] = fmaf(ad * sc, dop[(kb + kk) * H], accV[kk]);
          }
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      float vk = reduce_same_head(accK[kk], H, [](float p, float q) { return p + q; });
      float vv = reduce_same_head(accV[kk], H, [](float p, float q) { return p + q; });
      if (ls == 0 && kb + kk < d) {
        float* o = a.d_qkv + ((size_t)b * N + m) * 3 * dH + (kb + kk) * H + h;
        o[dH] = vk;
        o[2 * dH] = vv;
      }
    }
  }
}

__global__ void k_mask_sample(int which, uint32_t s0, uint32_t s1, uint32_t thr, size_t n,
                              uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (which == 0)
    out[i] = ((egt_hash32((uint32_t)i, s0, s1) >> 8) < thr) ? 1 : 0;
  else
    out[i] = ((egt_hash32((uint32_t)i, s0, s1 ^ EGT_DROPOUT_STREAM) >> 8) >= thr) ? 1 : 0;
}

// ------------------------------------------------------------------ host glue --
static int ilog2(int v) {
  int r = 0;
  while ((1 << r) < v) ++r;
  return r;
}

static int fill_args(const egt_attn_desc* d, AttnArgs& a) {
  if (!d) EGT_FAIL(EGT_E_NULL, "desc is NULL");
  if (d->dtype != EGT_F32) EGT_FAIL(EGT_E_DTYPE, "only EGT_F32 is supported (got %d)", d->dtype);
  if (d->B <= 0 || d->N <= 0 || d->d <= 0 || d->H <= 0)
    EGT_FAIL(EGT_E_SHAPE, "B,N,H,d must be positive (B=%d N=%d H=%d d=%d)", d->B, d->N, d->H, d->d);
  if (d->H > 64 || (d->H & (d->H - 1)))
    EGT_FAIL(EGT_E_SHAPE, "num_heads must be a power of two <= 64 (got %d)", d->H);
  if ((d->flags & EGT_F_SCALE_DEGREE) && !(d->flags & EGT_F_GATE_INPUT))
    EGT_FAIL(EGT_E_FLAGS, "scale_degree requires gate_input");  // egt_layers.py:20-21
  if ((size_t)d->B * d->N * d->N * d->H > 0xFFFFFFFFull && (d->flags & EGT_F_TRAINING))
    EGT_FAIL(EGT_E_SHAPE, "B*N*N*H exceeds the 32-bit RNG counter");
  a = AttnArgs{};
  a.B = d->B; a.N = d->N; a.H = d->H; a.d = d->d; a.logH = ilog2(d->H);
  a.flags = d->flags;
  a.clip_lo = d->clip_lo; a.clip_hi = d->clip_hi;
  a.scale = 1.0f / sqrtf((float)d->d);
  a.s0 = (uint32_t)(d->seed & 0xFFFFFFFFull);
  a.s1 = (uint32_t)(d->seed >> 32);
  a.rm_thr = egt_threshold24(d->random_mask_prob);
  a.dk_thr = egt_threshold24(d->attn_dropout);
  a.dk_scale = 1.0f / (1.0f - d->attn_dropout);
  a.nvn = d->num_virtual_nodes;
  return EGT_OK;
}

static void bind_stochastic(const egt_attn_desc* d, AttnArgs& a, const uint8_t* rand_mask,
                            const uint8_t* drop_keep) {
  const bool training = (d->flags & EGT_F_TRAINING) != 0;
  a.rm = nullptr; a.dk = nullptr; a.rng_rm = 0; a.rng_dk = 0;
  if (training && d->random_mask_prob > 0.0f) {  // egt_layers.py:103
    if (rand_mask) a.rm = rand_mask; else a.rng_rm = 1;
  }
  if (training && d->attn_dropout > 0.0f) {      // egt_layers.py:116
    if (drop_keep) a.dk = drop_keep; else a.rng_dk = 1;
  }
}

static int pick_waves(size_t floats_per_wave, int& waves, size_t& lds_bytes) {
  const size_t cap = 160 * 1024;
  waves = 4;
  while (waves > 1 && waves * floats_per_wave * 4 > cap) waves >>= 1;
  lds_bytes = (size_t)waves * floats_per_wave * 4;
  if (lds_bytes > cap) EGT_FAIL(EGT_E_SHAPE, "row of N*H does not fit the 160 KiB LDS");
  return EGT_OK;
}

extern "C" int egt_attn_fwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                            const void* G, const uint8_t* key_mask, const void* attn_mask,
                            const uint8_t* rand_mask, const uint8_t* drop_keep, void* v_att,
                            void* h_hat, void* a_tild, void* rowstats, void* stream) {
  AttnArgs a;
  int rc = fill_args(desc, a);
  if (rc) return rc;
  if (!qkv || !v_att || !h_hat || !rowstats) EGT_FAIL(EGT_E_NULL, "qkv/v_att/h_hat/rowstats is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  a.qkv = (const float*)qkv;
  a.E = (desc->flags & EGT_F_EDGE_INPUT) ? (const float*)E : nullptr;
  a.G = (desc->flags & EGT_F_GATE_INPUT) ? (const float*)G : nullptr;
  a.M = (desc->flags & EGT_F_ATTN_MASK) ? (const float*)attn_mask : nullptr;
  a.km = key_mask;
  bind_stochastic(desc, a, rand_mask, drop_keep);
  a.v_att = (float*)v_att; a.h_hat = (float*)h_hat; a.a_tild = (float*)a_tild;
  a.rowstats = (float*)rowstats;
  size_t fpw = (size_t)2 * a.N * a.H + (size_t)a.d * a.H;
  fpw = (fpw + 3) & ~(size_t)3;
  int waves; size_t lds;
  rc = pick_waves(fpw, waves, lds);
  if (rc) return rc;
  a.lds_per_wave = (int)fpw;
  const long rows = (long)a.B * a.N;
  dim3 grid((unsigned)((rows + waves - 1) / waves)), block(64 * waves);
  EGT_MAX_LDS_ONCE(k_attn_fwd);
  EGT_LAUNCH("k_attn_fwd", k_attn_fwd, grid, block, lds, (hipStream_t)stream, a);
  EGT_HIP_LAUNCH_CHECK("egt_attn_fwd");
  return EGT_OK;
}

extern "C" size_t egt_attn_bwd_workspace_bytes(const egt_attn_desc* d) {
  if (!d) return 0;
  return (size_t)2 * d->B * d->N * d->N * d->H * sizeof(float);
}

extern "C" int egt_attn_bwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                            const void* G, const uint8_t* key_mask, const void* attn_mask,
                            const uint8_t* rand_mask, const uint8_t* drop_keep, const void* v_att,
                            const void* rowstats, const void* d_v_att, const void* d_h_ext,
                            void* d_qkv, void* d_E, void* d_G, void* workspace, void* stream) {
  AttnArgs a;
  int rc = fill_args(desc, a);
  if (rc) return rc;
  if (!qkv || !v_att || !rowstats || !d_v_att || !d_qkv || !workspace)
    EGT_FAIL(EGT_E_NULL, "qkv/v_att/rowstats/d_v_att/d_qkv/workspace is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  a.qkv = (const float*)qkv;
  a.E = (desc->flags & EGT_F_EDGE_INPUT) ? (const float*)E : nullptr;
  a.G = (desc->flags & EGT_F_GATE_INPUT) ? (const float*)G : nullptr;
  a.M = (desc->flags & EGT_F_ATTN_MASK) ? (const float*)attn_mask : nullptr;
  a.km = key_mask;
  bind_stochastic(desc, a, rand_mask, drop_keep);
  a.v_att_in = (const float*)v_att; a.rowstats_in = (const float*)rowstats;
  a.d_v_att = (const float*)d_v_att; a.d_h_ext = (const float*)d_h_ext;
  a.d_qkv = (float*)d_qkv;
  a.d_E = (desc->flags & EGT_F_EDGE_INPUT) ? (float*)d_E : nullptr;
  a.d_G = (desc->flags & EGT_F_GATE_INPUT) ? (float*)d_G : nullptr;
  const size_t nelem = (size_t)a.B * a.N * a.N * a.H;
  a.ws_dA = (float*)workspace;
  a.ws_Ad = a.ws_dA + nelem;
  size_t fpw = (size_t)2 * a.d * a.H;
  fpw = (fpw + 3) & ~(size_t)3;
  int waves; size_t lds;
  rc = pick_waves(fpw, waves, lds);
  if (rc) return rc;
  a.lds_per_wave = (int)fpw;
  const long rows = (long)a.B * a.N;
  dim3 grid((unsigned)((rows + waves - 1) / waves)), block(64 * waves);
  EGT_MAX_LDS_ONCE(k_attn_bwd_row);
  EGT_LAUNCH("k_attn_bwd_row", k_attn_bwd_row, grid, block, lds, (hipStream_t)stream, a);
  EGT_HIP_LAUNCH_CHECK("egt_attn_bwd(row)");
  dim3 grid4((unsigned)((rows + 3) / 4)), block4(256);
  EGT_LAUNCH("k_attn_bwd_dq", k_attn_bwd_dq, grid4, block4, 0, (hipStream_t)stream, a);
  EGT_HIP_LAUNCH_CHECK("egt_attn_bwd(dq)");
  EGT_LAUNCH("k_attn_bwd_dkv", k_attn_bwd_dkv, grid4, block4, 0, (hipStream_t)stream, a);
  EGT_HIP_LAUNCH_CHECK("egt_attn_bwd(dkv)");
  return EGT_OK;
}

extern "C" int egt_mask_sample(int which, uint64_t seed, float prob, int32_t B, int32_t N,
                               int32_t H, uint8_t* out, void* stream) {
  if (!out) EGT_FAIL(EGT_E_NULL, "out is NULL");
  if (which != 0 && which != 1) EGT_FAIL(EGT_E_FLAGS, "which must be 0 or 1");
  const size_t n = (size_t)B * N * N * H;
  if (n > 0xFFFFFFFFull) EGT_FAIL(EGT_E_SHAPE, "B*N*N*H exceeds the 32-bit RNG counter");
  EGT_LAUNCH("k_mask_sample", k_mask_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, which, (uint32_t)(seed & 0xFFFFFFFFull),
                     (uint32_t)(seed >> 32), egt_threshold24(prob), n, out);
  EGT_HIP_LAUNCH_CHECK("egt_mask_sample");
  return EGT_OK;
}

__global__ void k_seed_advance(uint64_t* words, int count, uint64_t inc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) words[i] += inc;
}

extern "C" int egt_seed_advance(uint64_t* words, int32_t count, uint64_t increment, void* stream) {
  if (!words) EGT_FAIL(EGT_E_NULL, "words is NULL");
  if (count < 1) EGT_FAIL(EGT_E_SHAPE, "count must be >= 1");
  EGT_LAUNCH("k_seed_advance", k_seed_advance, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
             words, (int)count, increment);
  EGT_HIP_LAUNCH_CHECK("egt_seed_advance");
  return EGT_OK;
}
