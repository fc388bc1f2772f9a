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


// ================================================================= backward =====
// Three launches (flash-attention style, the [N,N,H] probabilities are recomputed):
//   k_attn_mfma_delta : delta[l,h] = sum_k dO[l,k,h] * O[l,k,h]            -> rowstats[...][3]
//   k_attn_mfma_bwd_kv: workgroup = (graph, 16-key tile), walks the query tiles; K/V fragments of
//                       the key tile live in registers; per tile S = Q.K^T and dP = dO.V^T on
//                       MFMA, softmax/gate/clip backward on the VALU, then dV^T += dO^T.A and
//                       dK^T += Q^T.dA on MFMA with the probabilities as B operands in place;
//                       writes dE, dG (through an LDS tile, whole rows) and dA = dH*c*scale
//   k_attn_mfma_bwd_q : workgroup = (graph, 16 query rows), walks the key tiles:
//                       dQ^T += K^T.dA^T on MFMA from the dA tensor
struct AttnMfmaBwdArgs {
  AttnMfmaArgs f;
  const float *v_att, *rowstats_in, *d_v_att, *d_h_ext;
  float *rowstats_rw, *d_qkv, *d_E, *d_G, *ws_dA;
};

__global__ void __launch_bounds__(256) k_attn_mfma_delta(AttnMfmaBwdArgs a) {
  const int d = a.f.d, DH = d * AH;
  const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);   // 8 threads (heads) per row
  const int h = threadIdx.x & 7;
  if (row >= (long)a.f.B * a.f.N) return;
  const float* dv = a.d_v_att + row * DH + h;
  const float* vo = a.v_att + row * DH + h;
  float s = 0.f;
  for (int k = 0; k < d; ++k) s = fmaf(dv[k * AH], vo[k * AH], s);
  a.rowstats_rw[(row * AH + h) * 4 + 3] = s;
}

template <int D>
__global__ void __launch_bounds__(256, 1) k_attn_mfma_bwd_kv(AttnMfmaBwdArgs a) {
  constexpr int KT = D / 16, KLD = D + 4, DH = D * AH;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                          // [8][16][KLD]   Q rows of the query tile, head-major
  float* Os = Qs + AH * 16 * KLD;          // [8][16][KLD]   dO rows
  float* Et = Os + AH * 16 * KLD;          // [16 l][PT_LD]
  float* Gt = Et + 16 * PT_LD;
  float* Mt = Gt + 16 * PT_LD;
  float* Xt = Mt + 16 * PT_LD;             // d_h_ext in
  float* dEt = Xt + 16 * PT_LD;            // dE out
  float* dGt = dEt + 16 * PT_LD;           // dG out
  float* dAt = dGt + 16 * PT_LD;           // dA out
  float* st = dAt + 16 * PT_LD;            // [16 l][8 h][4]
  const AttnMfmaArgs& f = a.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mm = lane & 15, q = lane >> 4;
  const int N = f.N;
  const int mtiles = (N + 15) / 16;
  const int b = blockIdx.x / mtiles, m0 = (blockIdx.x % mtiles) * 16;
  const int m = m0 + mm, mc = min(m, N - 1);
  const bool mvalid = m < N;
  const bool gated = (f.flags & EGT_F_GATE_INPUT) != 0;
  const bool clip = (f.flags & EGT_F_CLIP) != 0;

  // K / V fragments of this lane's key (B operands of S = Q.K^T and dP = dO.V^T)
  float Kr[2][4 * KT], Vr[2][4 * KT];
  {
    const float* krow = f.qkv + ((size_t)b * N + mc) * 3 * DH + DH;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int t = 0; t < 4 * KT; ++t) {
        const int k = 16 * (t >> 2) + 4 * q + (t & 3);
        Kr[hh][t] = mvalid ? krow[k * AH + 2 * wave + hh] : 0.f;
        Vr[hh][t] = mvalid ? krow[DH + k * AH + 2 * wave + hh] : 0.f;
      }
  }
  const float kadd = (f.km && mvalid && f.km[(size_t)b * N + m] == 0) ? -EGT_NEG : 0.0f;
  v4f dKacc[2][KT], dVacc[2][KT];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { dKacc[hh][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; dVacc[hh][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; }

  for (int l0 = 0; l0 < N; l0 += 16) {
    __syncthreads();
    // ---- stage Q and dO rows l0..l0+15 head-major; E/G/M/dH_ext tiles; row statistics ----
    for (int idx = tid; idx < 16 * (DH / 4); idx += 256) {
      const int row = idx / (DH / 4), c = (idx % (DH / 4)) * 4;
      const int l = l0 + row;
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), ov = qv;
      if (l < N) {
        qv = *reinterpret_cast<const float4*>(f.qkv + ((size_t)b * N + l) * 3 * DH + c);
        ov = *reinterpret_cast<const float4*>(a.d_v_att + ((size_t)b * N + l) * DH + c);
      }
      const int k = c >> 3, h = c & 7;
      float* qd = Qs + (h * 16 + row) * KLD + k;
      float* od = Os + (h * 16 + row) * KLD + k;
      qd[0] = qv.x; qd[16 * KLD] = qv.y; qd[32 * KLD] = qv.z; qd[48 * KLD] = qv.w;
      od[0] = ov.x; od[16 * KLD] = ov.y; od[32 * KLD] = ov.z; od[48 * KLD] = ov.w;
    }
    for (int idx = tid; idx < 16 * 32; idx += 256) {
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      const int lr = min(l0 + row, N - 1);
      const int mt = m0 + (c4 >> 3);
      const size_t g = (((size_t)b * N + lr) * N + m0) * AH + c4;
      const bool ok = mt < N;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f.E) *reinterpret_cast<float4*>(Et + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(f.E + g) : z;
      if (f.G) *reinterpret_cast<float4*>(Gt + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(f.G + g) : z;
      if (f.M) *reinterpret_cast<float4*>(Mt + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(f.M + g) : z;
      if (a.d_h_ext) *reinterpret_cast<float4*>(Xt + row * PT_LD + c4) = ok ? *reinterpret_cast<const float4*>(a.d_h_ext + g) : z;
    }
    for (int idx = tid; idx < 16 * AH; idx += 256) {
      const int row = idx >> 3, h = idx & 7;
      const int lr = min(l0 + row, N - 1);
      const float4 v = *reinterpret_cast<const float4*>(a.rowstats_in + (((size_t)b * N + lr) * AH + h) * 4);
      *reinterpret_cast<float4*>(st + idx * 4) = make_float4(v.x, 1.0f / v.y, v.w, 0.f);   // max, 1/sum, delta
    }
    __syncthreads();

#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      // ---- S[l][m] = sum_k Q[l][k] K[m][k] ; dP[l][m] = sum_k dO[l][k] V[m][k] ----
      v4f s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const float* qrow = Qs + (h * 16 + mm) * KLD + 4 * q;   // A rows are query rows: row index = lane&15
      const float* orow = Os + (h * 16 + mm) * KLD + 4 * q;
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        const float4 qa = *reinterpret_cast<const float4*>(qrow + 16 * T);
        const float4 oa = *reinterpret_cast<const float4*>(orow + 16 * T);
        s = MFMA(qa.x, Kr[hh][4 * T + 0], s);   dp = MFMA(oa.x, Vr[hh][4 * T + 0], dp);
        s = MFMA(qa.y, Kr[hh][4 * T + 1], s);   dp = MFMA(oa.y, Vr[hh][4 * T + 1], dp);
        s = MFMA(qa.z, Kr[hh][4 * T + 2], s);   dp = MFMA(oa.z, Vr[hh][4 * T + 2], dp);
        s = MFMA(qa.w, Kr[hh][4 * T + 3], s);   dp = MFMA(oa.w, Vr[hh][4 * T + 3], dp);
      }
      // lane (m = mm, q): query rows l = l0 + 4q + r
      float at[4], da[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int li = 4 * q + r, l = l0 + li;
        const bool valid = mvalid && l < N;
        const float araw = s[r] * f.scale;
        float ah = araw, inr = 1.0f;
        if (clip) {
          inr = (araw >= f.clip_lo && araw <= f.clip_hi) ? 1.0f : 0.0f;
          ah = fminf(fmaxf(araw, f.clip_lo), f.clip_hi);
        }
        const int po = li * PT_LD + mm * AH + h;
        float xv = ah, gv = gated ? Gt[po] : 0.f;
        if (f.E) xv += Et[po];
        if (f.km) { xv += kadd; gv += kadd; }
        if (f.M) { const float mk = (Mt[po] - 1.0f) * EGT_NEG; xv += mk; gv += mk; }
        if (f.rm || f.rng_rm) {
          const size_t gi = (((size_t)b * N + min(l, N - 1)) * N + mc) * AH + h;
          const bool hit = f.rm ? (f.rm[gi] != 0) : ((egt_hash32((uint32_t)gi, f.s0, f.s1) >> 8) < f.rm_thr);
          const float mr = hit ? -EGT_NEG : 0.0f;
          xv += mr; gv += mr;
        }
        const float* sr = st + (li * AH + h) * 4;
        const float S = valid ? __expf(xv - sr[0]) * sr[1] : 0.f;
        const float g = gated ? egt_sigmoid(gv) : 1.0f;
        const float dAt_ = dp[r];
        const float dS = dAt_ * g;
        float dH = S * (dS - sr[2]);
        if (a.d_h_ext) dH += Xt[po];
        if (!valid) dH = 0.f;
        dEt[po] = dH;
        dGt[po] = gated ? dAt_ * S * g * (1.0f - g) : 0.f;
        da[r] = dH * inr * f.scale;
        dAt[po] = da[r];
        at[r] = S * g;
      }
      // ---- dV^T[k][m] += sum_l dO[l][k] A[l][m] ; dK^T[k][m] += sum_l Q[l][k] dA[l][m] ----
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float* ocol = Os + (h * 16 + 4 * q) * KLD + 16 * kt + mm;
        const float* qcol = Qs + (h * 16 + 4 * q) * KLD + 16 * kt + mm;
        v4f dv = dVacc[hh][kt], dk = dKacc[hh][kt];
        dv = MFMA(ocol[0], at[0], dv);           dk = MFMA(qcol[0], da[0], dk);
        dv = MFMA(ocol[KLD], at[1], dv);         dk = MFMA(qcol[KLD], da[1], dk);
        dv = MFMA(ocol[2 * KLD], at[2], dv);     dk = MFMA(qcol[2 * KLD], da[2], dk);
        dv = MFMA(ocol[3 * KLD], at[3], dv);     dk = MFMA(qcol[3 * KLD], da[3], dk);
        dVacc[hh][kt] = dv; dKacc[hh][kt] = dk;
      }
    }
    __syncthreads();
    // ---- dE / dG / dA tiles out as whole rows ----
    for (int idx = tid; idx < 16 * 32; idx += 256) {
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      const int lr = l0 + row, mt = m0 + (c4 >> 3);
      if (lr < N && mt < N) {
        const size_t g = (((size_t)b * N + lr) * N + m0) * AH + c4;
        if (a.d_E) *reinterpret_cast<float4*>(a.d_E + g) = *reinterpret_cast<const float4*>(dEt + row * PT_LD + c4);
        if (a.d_G) *reinterpret_cast<float4*>(a.d_G + g) = *reinterpret_cast<const float4*>(dGt + row * PT_LD + c4);
        *reinterpret_cast<float4*>(a.ws_dA + g) = *reinterpret_cast<const float4*>(dAt + row * PT_LD + c4);
      }
    }
  }
  if (mvalid) {
    float* o = a.d_qkv + ((size_t)b * N + m) * 3 * DH;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kt + 4 * q + r;
          o[DH + k * AH + h] = dKacc[hh][kt][r];
          o[2 * DH + k * AH + h] = dVacc[hh][kt][r];
        }
    }
  }
}

template <int D>
__global__ void __launch_bounds__(256, 2) k_attn_mfma_bwd_q(AttnMfmaBwdArgs a) {
  constexpr int KT = D / 16, KLD = D + 4, DH = D * AH;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                          // [8][16][KLD]
  float* At = Ks + AH * 16 * KLD;          // dA tile [16 l][PT_LD]
  const AttnMfmaArgs& f = a.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ll = lane & 15, q = lane >> 4;
  const int N = f.N;
  const int ltiles = (N + 15) / 16;
  const int b = blockIdx.x / ltiles, l0 = (blockIdx.x % ltiles) * 16;
  const int l = l0 + ll;
  v4f dQacc[2][KT];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) dQacc[hh][kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < N; m0 += 16) {
    __syncthreads();
    for (int idx = tid; idx < 16 * (DH / 4); idx += 256) {
      const int row = idx / (DH / 4), c = (idx % (DH / 4)) * 4;
      const int m = m0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < N) kv = *reinterpret_cast<const float4*>(f.qkv + ((size_t)b * N + m) * 3 * DH + DH + c);
      const int k = c >> 3, h = c & 7;
      float* kd = Ks + (h * 16 + row) * KLD + k;
      kd[0] = kv.x; kd[16 * KLD] = kv.y; kd[32 * KLD] = kv.z; kd[48 * KLD] = kv.w;
    }
    for (int idx = tid; idx < 16 * 32; idx += 256) {
      const int row = idx >> 5, c4 = (idx & 31) * 4;
      const int lr = l0 + row, mt = m0 + (c4 >> 3);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lr < N && mt < N) v = *reinterpret_cast<const float4*>(a.ws_dA + (((size_t)b * N + lr) * N + m0) * AH + c4);
      *reinterpret_cast<float4*>(At + row * PT_LD + c4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      float da[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) da[t] = At[ll * PT_LD + (4 * q + t) * AH + h];   // dA[l][m = 4q + t]
      // dQ^T[k][l] += sum_m K[m][k] dA[l][m]
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float* kcol = Ks + (h * 16 + 4 * q) * KLD + 16 * kt + ll;
        v4f dq = dQacc[hh][kt];
        dq = MFMA(kcol[0], da[0], dq);
        dq = MFMA(kcol[KLD], da[1], dq);
        dq = MFMA(kcol[2 * KLD], da[2], dq);
        dq = MFMA(kcol[3 * KLD], da[3], dq);
        dQacc[hh][kt] = dq;
      }
    }
  }
  if (l < N) {
    float* o = a.d_qkv + ((size_t)b * N + l) * 3 * DH;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(16 * kt + 4 * q + r) * AH + h] = dQacc[hh][kt][r];
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

template <int D>
static void launch_bwd(const AttnMfmaBwdArgs& a, hipStream_t st) {
  const int N = a.f.N, tiles = (N + 15) / 16;
  const long rows = (long)a.f.B * N;
  EGT_LAUNCH("k_attn_mfma_delta", k_attn_mfma_delta, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, st, a);
  const size_t lds_kv = ((size_t)2 * AH * 16 * (D + 4) + 7 * 16 * PT_LD + 16 * AH * 4) * 4;
  (void)hipFuncSetAttribute((const void*)k_attn_mfma_bwd_kv<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  EGT_LAUNCH("k_attn_mfma_bwd_kv", k_attn_mfma_bwd_kv<D>, dim3(a.f.B * tiles), dim3(256), lds_kv, st, a);
  const size_t lds_q = ((size_t)AH * 16 * (D + 4) + 16 * PT_LD) * 4;
  EGT_LAUNCH("k_attn_mfma_bwd_q", k_attn_mfma_bwd_q<D>, dim3(a.f.B * tiles), dim3(256), lds_q, st, a);
}

// rowstats is read AND written (slot 3 receives delta); workspace: egt_attn_bwd_workspace_bytes()
extern "C" int egt_attn_mfma_bwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                                 const uint8_t* rand_mask, const void* v_att, void* rowstats,
                                 const void* d_v_att, const void* d_h_ext, void* d_qkv, void* d_E,
                                 void* d_G, void* workspace, void* stream) {
  if (!egt_attn_mfma_supported(desc, 0)) EGT_FAIL(EGT_E_SHAPE, "configuration not covered by the MFMA inner-op kernel");
  if (!qkv || !v_att || !rowstats || !d_v_att || !d_qkv || !workspace)
    EGT_FAIL(EGT_E_NULL, "qkv/v_att/rowstats/d_v_att/d_qkv/workspace is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  AttnMfmaBwdArgs a{};
  AttnMfmaArgs& f = a.f;
  f.B = desc->B; f.N = desc->N; f.d = desc->d; f.flags = desc->flags;
  f.clip_lo = desc->clip_lo; f.clip_hi = desc->clip_hi;
  f.scale = 1.0f / sqrtf((float)desc->d);
  f.rm_thr = egt_threshold24(desc->random_mask_prob);
  f.s0 = (uint32_t)(desc->seed & 0xFFFFFFFFull); f.s1 = (uint32_t)(desc->seed >> 32);
  f.qkv = (const float*)qkv;
  f.E = (desc->flags & EGT_F_EDGE_INPUT) ? (const float*)E : nullptr;
  f.G = (desc->flags & EGT_F_GATE_INPUT) ? (const float*)G : nullptr;
  f.M = (desc->flags & EGT_F_ATTN_MASK) ? (const float*)attn_mask : nullptr;
  f.km = key_mask;
  if ((desc->flags & EGT_F_TRAINING) && desc->random_mask_prob > 0.0f) {
    if (rand_mask) f.rm = rand_mask; else f.rng_rm = 1;
  }
  a.v_att = (const float*)v_att; a.rowstats_in = (const float*)rowstats; a.rowstats_rw = (float*)rowstats;
  a.d_v_att = (const float*)d_v_att; a.d_h_ext = (const float*)d_h_ext;
  a.d_qkv = (float*)d_qkv;
  a.d_E = (desc->flags & EGT_F_EDGE_INPUT) ? (float*)d_E : nullptr;
  a.d_G = (desc->flags & EGT_F_GATE_INPUT) ? (float*)d_G : nullptr;
  a.ws_dA = (float*)workspace;
  switch (desc->d) {
    case 16: launch_bwd<16>(a, (hipStream_t)stream); break;
    case 32: launch_bwd<32>(a, (hipStream_t)stream); break;
    default: launch_bwd<64>(a, (hipStream_t)stream); break;
  }
  EGT_HIP_LAUNCH_CHECK("egt_attn_mfma_bwd");
  return EGT_OK;
}
