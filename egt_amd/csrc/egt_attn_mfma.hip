// Inner op, MFMA-tiled (flash-style) for the large-head geometry (d % 16 == 0, H = 8):
//   (V_att, H_hat) = EGT([QKV, E?, G?, M?], mask)     lib/models/egt_layers.py:57-213
// This is the shape where QK^T / A.V dominate (BASELINE config 5: N=512, d=64: fp32
// arithmetic intensity above the ridge), so both contractions run on
// v_mfma_f32_16x16x4_f32 and the N x N probabilities never exist outside registers.
//
// Workgroup = (graph b, 16 query rows); wave w owns heads 2w, 2w+1.  Per 16-key tile:
//   K/V rows are staged head-major in LDS ([h][m][k], padded), E/G/(M) tiles are copied
//   coalesced; S^T = K.Q^T is computed with the key index on the MFMA row axis so that a lane
//   holds one query row l = lane&15 and four keys m = 4q+r: the softmax reductions over keys
//   are 3 in-lane ops + two permlane swaps, the online-softmax rescale factor is lane-uniform,
//   and the gated probabilities feed the A.V MFMA as its B operand WITHOUT any data movement
//   (O^T[k][l] += V^T[k][m] . P^T[m][l], contraction order m = 4q + t).
//   H_hat leaves through an LDS tile as whole 512-byte rows.
// Attributes outside this kernel's cover (dropout, degree scalers, A_tild output, H != 8,
// d % 16 != 0) use the general kernels of egt_attn.hip.
#include "egt_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define AH 8
#define PT_LD 132  // pair-tile row stride: [16 l][16 m][8 h] + 4 floats of padding per l

struct AttnMfmaArgs {
  int B, N, d;
  uint32_t flags;
  float clip_lo, clip_hi, scale;
  uint32_t rm_thr, s0, s1;
  int rng_rm;
  const float *qkv, *E, *G, *M;
  const uint8_t *km, *rm;
  float *v_att, *h_hat, *rowstats;
};

__device__ __forceinline__ float pair_max_q(float v) {   // max over lanes l, l+16, l+32, l+48
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float pair_sum_q(float v) { return sum_xor32(sum_xor16(v)); }

template <int D>
__global__ void __launch_bounds__(256, 1) k_attn_mfma_fwd(AttnMfmaArgs a) {
  constexpr int KT = D / 16;        // 16-wide k tiles
  constexpr int KLD = D + 4;        // padded k stride of the staged K/V rows
  constexpr int DH = D * AH;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                          // [8][16][KLD]
  float* Vs = Ks + AH * 16 * KLD;          // [8][16][KLD]
  float* Et = Vs + AH * 16 * KLD;          // [16][PT_LD]
  float* Gt = Et + 16 * PT_LD;
  float* Mt = Gt + 16 * PT_LD;
  float* Ht = Mt + 16 * PT_LD;             // H_hat out
  float* kmadd = Ht + 16 * PT_LD;          // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ll = lane & 15, q = lane >> 4;
  const int N = a.N;
  const int ltiles = (N + 15) / 16;
  const int b = blockIdx.x / ltiles, l0 = (blockIdx.x % ltiles) * 16;
  const int l = l0 + ll, lc = min(l, N - 1);
  const bool gated = (a.flags & EGT_F_GATE_INPUT) != 0;
  const bool clip = (a.flags & EGT_F_CLIP) != 0;

  // Q fragments (B operand of S^T = K.Q^T): lane (l, q) holds Q[l][16T + 4q + u] of its two heads
  float Qr[2][4 * KT];
  {
    const float* qrow = a.qkv + ((size_t)b * N + lc) * 3 * DH;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < 4 * KT; ++t) {
        const int k = 16 * (t >> 2) + 4 * q + (t & 3);
        Qr[hh][t] = qrow[k * AH + 2 * wave + hh];
      }
  }
  v4f oacc[2][KT];
  float mrun[2], lrun[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    mrun[hh] = -3.0e38f; lrun[hh] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) oacc[hh][kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  }

  for (int m0 = 0; m0 < N; m0 += 16) {
    __syncthreads();
    // ---- stage K, V rows m0..m0+15 head-major; E/G/M tiles as whole rows ----
    for (int idx = tid; idx < 16 * (DH / 4); idx += 256) {
      const int row = idx / (DH / 4), c = (idx % (DH / 4)) * 4;   // channel c = k*8 + h
      const int m = m0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (m < N) {
        const float* src = a.qkv + ((size_t)b * N + m) * 3 * DH + DH + c;
        kv = *reinterpret_cast<const float4*>(src);
        vv = *reinterpret_cast<const float4*>(src + DH);
      }
      const int k = c >> 3, h = c & 7;
      float* kd = Ks + (h * 16 + row) * KLD + k;
      float* vd = Vs + (h * 16 + row) * KLD + k;
      kd[0] = kv.x; kd[16 * KLD] = kv.y; kd[32 * KLD] = kv.z; kd[48 * KLD] = kv.w;
      vd[0] = vv.x; vd[16 * KLD] = vv.y; vd[32 * KLD] = vv.z; vd[48 * KLD] = vv.w;
    }
    for (int idx = tid; idx < 16 * 32; idx += 256) {
      const int row = idx >> 5, c4 = (idx & 31) * 4;     // row l0+row, 128 floats = [16 m][8 h]
      const int lr = min(l0 + row, N - 1);
      const int m = m0 + (c4 >> 3);
      const size_t g = (((size_t)b * N + lr) * N + m0) * AH + c4;
      const bool ok = m < N;
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.E) *reinterpret_cast<float4*>(Et + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(a.E + g) : z;
      if (a.G) *reinterpret_cast<float4*>(Gt + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(a.G + g) : z;
      if (a.M) *reinterpret_cast<float4*>(Mt + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(a.M + g) : z;
    }
    if (tid < 16) {
      const int m = m0 + tid;
      kmadd[tid] = (a.km && m < N && a.km[(size_t)b * N + m] == 0) ? -EGT_NEG : 0.0f;
    }
    __syncthreads();

#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      // ---- S^T[m][l] = sum_k K[m][k] Q[l][k]  (A = K rows, B = Q fragments) ----
      v4f s = {0.f, 0.f, 0.f, 0.f};
      const float* krow = Ks + (h * 16 + ll) * KLD + 4 * q;
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        const float4 ka = *reinterpret_cast<const float4*>(krow + 16 * T);
        s = MFMA(ka.x, Qr[hh][4 * T + 0], s);
        s = MFMA(ka.y, Qr[hh][4 * T + 1], s);
        s = MFMA(ka.z, Qr[hh][4 * T + 2], s);
        s = MFMA(ka.w, Qr[hh][4 * T + 3], s);
      }
      // lane (l = ll, q): keys m = m0 + 4q + r
      float x[4], pa[4];
      float tmax = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mi = 4 * q + r, m = m0 + mi;
        const bool valid = m < N;
        float ah = s[r] * a.scale;
        if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
        const int po = ll * PT_LD + mi * AH + h;
        float hv = ah;
        if (a.E) hv += Et[po];
        Ht[po] = hv;                                         // H_hat: post-clip, pre-mask (egt_layers.py:85-86)
        float xv = hv, gv = gated ? Gt[po] : 0.f;
        if (a.km) { xv += kmadd[mi]; gv += kmadd[mi]; }
        if (a.M) { const float mm = (Mt[po] - 1.0f) * EGT_NEG; xv += mm; gv += mm; }
        if (a.rm || a.rng_rm) {
          const size_t gi = (((size_t)b * N + lc) * N + (valid ? m : 0)) * AH + h;
          const bool hit = a.rm ? (a.rm[gi] != 0) : ((egt_hash32((uint32_t)gi, a.s0, a.s1) >> 8) < a.rm_thr);
          const float mr = hit ? -EGT_NEG : 0.0f;
          xv += mr; gv += mr;
        }
        x[r] = valid ? xv : -3.0e38f;
        pa[r] = gated ? egt_sigmoid(gv) : 1.0f;            // gate (multiplied into p below)
        tmax = fmaxf(tmax, x[r]);
      }
      // ---- online softmax over the key axis: in-lane over r, then across q ----
      tmax = pair_max_q(tmax);
      const float mnew = fmaxf(mrun[hh], tmax);
      const float alpha = __expf(mrun[hh] - mnew);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = (x[r] > -2.9e38f) ? __expf(x[r] - mnew) : 0.f;
        psum += pe;
        pa[r] *= pe;
      }
      psum = pair_sum_q(psum);
      lrun[hh] = fmaf(lrun[hh], alpha, psum);
      mrun[hh] = mnew;
      // ---- O^T[k][l] = alpha * O^T + sum_m V[m][k] * P[l][m]  (contraction order m = 4q + t) ----
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        v4f o = oacc[hh][kt];
        o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
        const float* vcol = Vs + (h * 16 + 4 * q) * KLD + 16 * kt + ll;
        o = MFMA(vcol[0], pa[0], o);
        o = MFMA(vcol[KLD], pa[1], o);
        o = MFMA(vcol[2 * KLD], pa[2], o);
        o = MFMA(vcol[3 * KLD], pa[3], o);
        oacc[hh][kt] = o;
      }
    }
    __syncthreads();
    // ---- H_hat tile out: 16 rows of 16*8 floats ----
    for (int idx = tid; idx < 16 * 32; idx += 256) {
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      const int lr = l0 + row, m = m0 + (c4 >> 3);
      if (lr < N && m < N)
        *reinterpret_cast<float4*>(a.h_hat + (((size_t)b * N + lr) * N + m0) * AH + c4) =
            *reinterpret_cast<const float4*>(Ht + row * PT_LD + c4);
    }
  }
  // ---- finalize: V_att[l][k*8+h] = O[l][k] / l_run ; row statistics for the backward ----
  if (l < N) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      const float inv = 1.0f / lrun[hh];
      float* vo = a.v_att + ((size_t)b * N + l) * DH;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vo[(16 * kt + 4 * q + r) * AH + h] = oacc[hh][kt][r] * inv;
      if (q == 0) {
        float* rs = a.rowstats + (((size_t)b * N + l) * AH + h) * 4;
        rs[0] = mrun[hh]; rs[1] = lrun[hh]; rs[2] = 0.f; rs[3] = 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ host glue --
extern "C" int egt_attn_mfma_supported(const egt_attn_desc* d, int need_a_tild) {
  if (!d || d->dtype != EGT_F32 || d->H != AH) return 0;
  if (d->d != 16 && d->d != 32 && d->d != 64) return 0;
  if (d->flags & EGT_F_SCALE_DEGREE) return 0;
  if ((d->flags & EGT_F_TRAINING) && d->attn_dropout > 0.0f) return 0;
  if (need_a_tild) return 0;
  if ((size_t)d->B * d->N * d->N * d->H > 0xFFFFFFFFull) return 0;
  return 1;
}

template <int D>
static void launch_fwd(const AttnMfmaArgs& a, hipStream_t st) {
  const int ltiles = (a.N + 15) / 16;
  const size_t lds = ((size_t)2 * AH * 16 * (D + 4) + 4 * 16 * PT_LD + 16) * 4;
  (void)hipFuncSetAttribute((const void*)k_attn_mfma_fwd<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_attn_mfma_fwd", k_attn_mfma_fwd<D>, dim3(a.B * ltiles), dim3(256), lds, st, a);
}

extern "C" int egt_attn_mfma_fwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                                 const uint8_t* rand_mask, void* v_att, void* h_hat, void* rowstats,
                                 void* stream) {
  if (!egt_attn_mfma_supported(desc, 0)) EGT_FAIL(EGT_E_SHAPE, "configuration not covered by the MFMA inner-op kernel");
  if (!qkv || !v_att || !h_hat || !rowstats) EGT_FAIL(EGT_E_NULL, "qkv/v_att/h_hat/rowstats is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  AttnMfmaArgs a{};
  a.B = desc->B; a.N = desc->N; a.d = desc->d; a.flags = desc->flags;
  a.clip_lo = desc->clip_lo; a.clip_hi = desc->clip_hi;
  a.scale = 1.0f / sqrtf((float)desc->d);
  a.rm_thr = egt_threshold24(desc->random_mask_prob);
  a.s0 = (uint32_t)(desc->seed & 0xFFFFFFFFull); a.s1 = (uint32_t)(desc->seed >> 32);
  a.qkv = (const float*)qkv;
  a.E = (desc->flags & EGT_F_EDGE_INPUT) ? (const float*)E : nullptr;
  a.G = (desc->flags & EGT_F_GATE_INPUT) ? (const float*)G : nullptr;
  a.M = (desc->flags & EGT_F_ATTN_MASK) ? (const float*)attn_mask : nullptr;
  a.km = key_mask;
  if ((desc->flags & EGT_F_TRAINING) && desc->random_mask_prob > 0.0f) {
    if (rand_mask) a.rm = rand_mask; else a.rng_rm = 1;
  }
  a.v_att = (float*)v_att; a.h_hat = (float*)h_hat; a.rowstats = (float*)rowstats;
  switch (desc->d) {
    case 16: launch_fwd<16>(a, (hipStream_t)stream); break;
    case 32: launch_fwd<32>(a, (hipStream_t)stream); break;
    default: launch_fwd<64>(a, (hipStream_t)stream); break;
  }
  EGT_HIP_LAUNCH_CHECK("egt_attn_mfma_fwd");
  return EGT_OK;
}
