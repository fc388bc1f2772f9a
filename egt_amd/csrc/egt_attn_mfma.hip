// Inner op, MFMA-tiled (flash-style) for the large-head geometry (d in {16,32,64}, H = 8):
//   (V_att, H_hat) = EGT([QKV, E?, G?, M?], mask)     lib/models/egt_layers.py:57-213
// and its backward (SURVEY.md appendix A / egt_layers.py semantics under autodiff).
// This is the shape where QK^T / A.V dominate (BASELINE config 5: N=512, d=64: fp32 arithmetic
// intensity at the ridge), so every contraction runs on v_mfma_f32_16x16x4_f32 and the N x N
// probabilities never exist outside registers.
//
// Layout decision: a pack kernel first rewrites Q/K/V (and dV_att) head-major, both row-major
// [B,H,NP/16,d/16,16,16] and transposed [B,H,NP/16,d,16], both tile-major (NP = N rounded up to 16, zero padded).  With that,
// EVERY MFMA operand that comes from Q/K/V/dO is one aligned 16-byte global load per four
// contraction steps, straight into the lane that feeds the matrix core:
//   operand "rows x contraction":  lane (row = lane&15, q = lane>>4) holds X[row][16T + 4q + u]
//   (the contraction order kappa(T,u,q) = 16T + 4q + u is used on both operands, so it is free).
// There is no LDS staging, no block barrier and no cross-wave dependency anywhere: a wave owns
// one head, a workgroup (8 waves = 8 heads) one 16-row tile; the eight waves touch the same
// 32-byte sectors of the [N,N,8] pair tensors, which the CU's vector L1 merges.
// The pair-tensor elements of a lane sit in the MFMA accumulator layout (row = 4q + r), which is
// exactly what the second contraction of each phase needs as its B operand: probabilities never
// move between lanes.  Softmax reductions over the key axis are 3 in-lane ops + two permlane
// swaps.
// Attributes outside this cover (dropout, degree scalers, A_tild output, H != 8, other d) use
// the general kernels of egt_attn.hip.
#include <stdlib.h>

#include "egt_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define AH 8

// packed operand arrays, each B*H*NP*d floats, in this order inside the workspace
enum { PK_KH = 0, PK_VT, PK_QH, PK_QT, PK_KT, PK_VH, PK_OH, PK_OT, PK_COUNT };

// Timing ablations of the forward (which term costs what): build with -DEGT_ATTN_ABLATION
// (EGT_ATTN_FLAGS, build.py) and set EGT_ATTN_ABLATE=<bits> (1 K/V loads, 2 H_hat stores,
// 4 pair-tile loads, 8 MFMAs, 16 barrier).  Compiled out otherwise: the always-taken branches
// split basic blocks and cost 2-3 %.
#ifdef EGT_ATTN_ABLATION
#define ABL_ON(a, bit) (!((a).guard & (bit)))
#else
#define ABL_ON(a, bit) true
#endif

struct AttnMfmaArgs {
  int B, N, NP, d;
  uint32_t flags;
  float clip_lo, clip_hi, scale;
  uint32_t rm_thr, s0, s1;
  int rng_rm;
  const float *qkv, *E, *G, *M;
  const uint8_t *km, *rm;
  float *v_att, *h_hat, *rowstats;
  float* pk;   // packed arrays
  // backward
  const float *v_att_in, *d_v_att, *d_h_ext;
  float *d_qkv, *d_E, *d_G, *ws_dA;
  int pack_bwd;
  int guard;   // timing ablations (EGT_ATTN_ABLATE), 0 in production
};

__device__ __forceinline__ float pair_max_q(float v) {   // max over lanes l, l+16, l+32, l+48
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float pair_sum_q(float v) { return sum_xor32(sum_xor16(v)); }

// ------------------------------------------------------------------- pack --------
// workgroup = (graph, 16 node rows): the rows' QKV (and dV_att) channels go through an LDS tile
// and leave head-major.  Channel index of the source: c = s*d*H + k*H + h (egt_layers.py:70-76).
template <int D>
__global__ void __launch_bounds__(256) k_attn_pack(AttnMfmaArgs a) {
  // workgroup = (graph, 16 node rows, ONE section of [q | k | v | dO]): forward packs k and v only
  constexpr int DH = D * AH, LD = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [16][LD]
  const int N = a.N, NP = a.NP, tid = threadIdx.x;
  const int tiles = NP / 16;
  const bool bwd = a.pack_bwd != 0;
  const int nsec = bwd ? 4 : 2;
  const int sec = bwd ? (int)(blockIdx.x % nsec) : (int)(blockIdx.x % nsec) + 1;
  const int tile = blockIdx.x / nsec;
  const int b = tile / tiles, n0 = (tile % tiles) * 16;
  for (int i = tid; i < 16 * (DH / 4); i += 256) {
    const int r = i / (DH / 4), c = (i % (DH / 4)) * 4;
    const int n = n0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N)
      v = sec < 3 ? *reinterpret_cast<const float4*>(a.qkv + ((size_t)b * N + n) * 3 * DH + sec * DH + c)
                  : *reinterpret_cast<const float4*>(a.d_v_att + ((size_t)b * N + n) * DH + c);
    *reinterpret_cast<float4*>(sm + r * LD + c) = v;
  }
  __syncthreads();
  const size_t arr = (size_t)a.B * AH * NP * D;
  // the staged rows -> [b,h,n/16,k/16,16 nodes,16 channels] array `which`: a wave's
  // operand fetch (16 nodes x 16 channels of one k-tile) is ONE contiguous 1 KB block
  auto put_rows = [&](int which) {
    float* dst = a.pk + (size_t)which * arr;
    for (int i = tid; i < AH * 16 * (D / 4); i += 256) {
      const int k4 = ((i >> 6) % (D / 16)) * 4 + (i & 3), r = (i >> 2) & 15, h = i / (4 * D);
      const float* src = sm + r * LD + (k4 * 4) * AH + h;
      *reinterpret_cast<float4*>(dst + ((size_t)b * AH + h) * NP * D + (size_t)n0 * D + (k4 >> 2) * 256 + r * 16 + (k4 & 3) * 4) =
          make_float4(src[0], src[AH], src[2 * AH], src[3 * AH]);
    }
  };
  // ... -> transposed, tile-major [b,h,n/16,k,16] array: the 16 nodes of a tile are contiguous per
  // channel and a tile is one 64*D-byte block, so a wave's operand fetch uses whole cache lines
  // (with [b,h,k,n] rows a 16-node access touches half of each 128-byte line)
  auto put_cols = [&](int which) {
    float* dst = a.pk + (size_t)which * arr;
    for (int i = tid; i < AH * D * 4; i += 256) {
      const int r4 = i & 3, k = (i >> 2) % D, h = i / (4 * D);
      const float* src = sm + (r4 * 4) * LD + k * AH + h;
      *reinterpret_cast<float4*>(dst + ((size_t)b * AH + h) * D * NP + (size_t)n0 * D + k * 16 + r4 * 4) =
          make_float4(src[0], src[LD], src[2 * LD], src[3 * LD]);
    }
  };
  // forward: K rows + V^T; backward: Q, K, V, dO rows + K^T (bwd_kv transposes Q / dO in registers)
  if (sec == 0) put_rows(PK_QH);
  else if (sec == 1) { put_rows(PK_KH); if (bwd) put_cols(PK_KT); }
  else if (sec == 2) { if (bwd) put_rows(PK_VH); else put_cols(PK_VT); }
  else {
    put_rows(PK_OH);
    // delta[row, h] = sum_k dO[row, k, h] * O[row, k, h] (flash-style): the dO rows are staged here anyway, so the
    // separate k_attn_mfma_delta launch (15 us of latency at config 5) is folded in -> rowstats[..][3]
    const int r = tid >> 4, hh = (tid >> 1) & 7, half = tid & 1, n = n0 + r;
    float sdel = 0.f;
    if (n < N) {
      const float* vo = a.v_att_in + ((size_t)b * N + n) * DH + hh;
      const float* dr = sm + r * LD + hh;
#pragma unroll 8
      for (int k = half * (D / 2); k < (half + 1) * (D / 2); ++k) sdel = fmaf(dr[k * AH], vo[k * AH], sdel);
    }
    sdel += __shfl_xor(sdel, 1, 64);
    if (n < N && half == 0) a.rowstats[(((size_t)b * N + n) * AH + hh) * 4 + 3] = sdel;
  }
}

// Feature set of a kernel instance.  V = 0 reads every switch at run time (any combination);
// V = 1 / 2 are the straight-line instances of the main configuration (edge bias + gates + key
// padding + clip, no attention-mask tensor, no injected mask bytes; 2 = in-kernel random mask):
// without the per-feature branches the compiler interleaves the MFMAs with the VALU work.
template <int V>
struct Feat {
  bool E, G, M, km, clip, rmb, rng, X;
  __device__ __forceinline__ Feat(const AttnMfmaArgs& a) {
    E = V ? true : a.E != nullptr;
    G = V ? true : (a.flags & EGT_F_GATE_INPUT) != 0;
    M = V ? false : a.M != nullptr;
    km = V ? true : a.km != nullptr;
    clip = V ? true : (a.flags & EGT_F_CLIP) != 0;
    rmb = V ? false : a.rm != nullptr;
    rng = V == 2 ? true : (V == 1 ? false : a.rng_rm != 0);
    X = V ? true : a.d_h_ext != nullptr;
  }
};

// additive masks of one element, in the reference's order (egt_layers.py:91-108)
template <int V>
__device__ __forceinline__ float mask_add(const AttnMfmaArgs& a, const Feat<V>& f, float kadd, float mval, size_t gi) {
  float add = 0.f;
  if (f.km) add += kadd;
  if (f.M) add += (mval - 1.0f) * EGT_NEG;
  if (f.rmb || f.rng) {
    const bool hit = f.rmb ? (a.rm[gi] != 0) : ((egt_hash32((uint32_t)gi, a.s0, a.s1) >> 8) < a.rm_thr);
    add += hit ? -EGT_NEG : 0.0f;
  }
  return add;
}

// ---- cooperative pair-tile transfers: a [16 rows][16 cols x 8 heads] tile of a [B,N,N,8]
// tensor is 16 contiguous 512-byte runs; the workgroup's 512 threads move it with one 16-byte
// global access each (thread -> row tid>>5, column (tid&31)>>1, heads 4*(tid&1)..+3).  In LDS
// the tile is HEAD-MAJOR: eight [16][16] planes of stride PT_PL, so the wave that owns head h
// reads its four consecutive columns with one ds_read_b128 (forward / bwd_q lanes) or walks a
// column with ds_read_b32 (bwd_kv lanes).  Inside a plane rows 4..7 and 12..15 are pair-swapped
// and the 4-column chunks are XORed with (row>>1)&3: every access pattern of the three kernels
// and the 4 x b32 cooperative scatter is bank-conflict free (enumerated against the gfx950
// lane-group tables; the [row][col][head] layout it replaces was 4- to 8-way conflicted).
// Rows / columns past N read a valid address and are zeroed (component-wise selects).
#define PT_PL 260
#define PT_SZ (8 * PT_PL)

__device__ __forceinline__ int pt_off(int row, int m) {   // offset inside one head's plane
  return ((row ^ ((row >> 2) & 1)) << 4) + ((((m >> 2) ^ (row >> 1)) & 3) << 2) + (m & 3);
}
__device__ __forceinline__ float4 ptile_gload(const float* src, int b, int N, int row0, int col0, int tid) {
  const int row = tid >> 5, c4 = (tid & 31) * 4;
  const int rr = min(row0 + row, N - 1), cc = min(col0 + (c4 >> 3), N - 1);
  return *reinterpret_cast<const float4*>(src + (((size_t)b * N + rr) * N + cc) * AH + (c4 & 7));
}
__device__ __forceinline__ void ptile_lds_put(float* tl, float4 v, int N, int row0, int col0, int tid) {
  const int row = tid >> 5, m = (tid & 31) >> 1;
  const bool ok = row0 + row < N && col0 + m < N;
  float* p = tl + (tid & 1) * 4 * PT_PL + pt_off(row, m);
  p[0] = ok ? v.x : 0.f;
  p[PT_PL] = ok ? v.y : 0.f;
  p[2 * PT_PL] = ok ? v.z : 0.f;
  p[3 * PT_PL] = ok ? v.w : 0.f;
}
__device__ __forceinline__ void ptile_gstore(float* dst, const float* tl, int b, int N, int row0, int col0, int tid) {
  const int row = tid >> 5, m = (tid & 31) >> 1;
  if (row0 + row < N && col0 + m < N) {
    const float* p = tl + (tid & 1) * 4 * PT_PL + pt_off(row, m);
    *reinterpret_cast<float4*>(dst + (((size_t)b * N + row0 + row) * N + col0 + m) * AH + (tid & 1) * 4) =
        make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
  }
}

// Same transfers with the address split into a workgroup-uniform base (tensor + graph + first row,
// scalar registers) and a 32-bit lane offset: rowoff = ptile_rowoff() is fixed for the kernel, the
// column part is two VALU ops per tile, and one offset serves every [B,N,N,8] tensor.
__device__ __forceinline__ uint32_t ptile_rowoff(int N, int row0, int tid) {
  return (uint32_t)(min(row0 + (tid >> 5), N - 1) - row0) * N * AH + (tid & 1) * 4;
}
__device__ __forceinline__ uint32_t ptile_off(uint32_t rowoff, int N, int col0, int tid) {
  return rowoff + (uint32_t)min(col0 + ((tid & 31) >> 1), N - 1) * AH;
}
__device__ __forceinline__ const float* ptile_base(const float* src, int b, int N, int row0) {
  return src + ((size_t)b * N + row0) * N * AH;
}

// key-mask bytes of keys m .. m+3 (clamped).  Kept as four separate registers: packing them
// would consume the loads at once, and a wait on these (the youngest loads of the prefetch
// group) would drain the whole group.
struct Km4 { uint32_t v[4]; };
__device__ __forceinline__ Km4 km_load4(const uint8_t* km, int N, int m) {
  Km4 k;
#pragma unroll
  for (int r = 0; r < 4; ++r) k.v[r] = km[min(m + r, N - 1)];
  return k;
}

// ================================================================== forward =====
// workgroup = (graph b, 16 query rows), wave = head.  Lane (ll = lane&15, q = lane>>4) owns
// query row l0 + ll and, in every key tile, keys m0 + 4q + r.  K / V^T operands come straight
// from the packed arrays (register prefetch one tile ahead); the E / G / M tiles are fetched
// coalesced by the whole workgroup one tile ahead into double-buffered LDS tiles, H_hat leaves
// through one: ONE barrier per key tile.
template <int D, int V>
__global__ void __launch_bounds__(512, 4) k_attn_mfma_fwd(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* In = sm;                    // [2][E | G | M][PT_SZ]
  float* Hout = sm + 2 * 3 * PT_SZ;  // [2][PT_SZ]
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ll = lane & 15, q = lane >> 4;
  const int N = a.N, NP = a.NP;
  const int ltiles = NP / 16;
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);   // a graph's K / V^T stay within one XCD's L2
  const int b = wg / ltiles, l0 = (wg % ltiles) * 16;
  const int l = l0 + ll, lc = min(l, N - 1);
  const Feat<V> f(a);
  const bool gated = f.G, clip = f.clip;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const float* Kh = a.pk + PK_KH * arr + ((size_t)b * AH + h) * NP * D;   // [NP/16][D/16][16][16]
  const float* VT = a.pk + PK_VT * arr + ((size_t)b * AH + h) * D * NP;   // [NP/16][D][16]
  // uniform bases + 32-bit lane offsets: scalar address arithmetic, one VGPR per access stream
  const uint32_t koff = ll * 16 + 4 * q, voff = koff;
  const uint32_t prow = ptile_rowoff(N, l0, tid);
  const float* Eb = ptile_base(a.E, b, N, l0);
  const float* Gb = ptile_base(a.G, b, N, l0);
  const float* Mb = ptile_base(a.M, b, N, l0);
  float* Hb = const_cast<float*>(ptile_base(a.h_hat, b, N, l0));
  auto pload = [&](const float* base, int col0) __attribute__((always_inline)) {
    return *reinterpret_cast<const float4*>(base + ptile_off(prow, N, col0, tid));
  };

  // Q fragments (B operand of S^T = K.Q^T): Q[l][16T + 4q + u]
  float Qr[4 * KT];
  {
    const float* qrow = a.qkv + ((size_t)b * N + lc) * 3 * DH + h;
#pragma unroll
    for (int t = 0; t < 4 * KT; ++t) Qr[t] = qrow[(16 * (t >> 2) + 4 * q + (t & 3)) * AH];
  }
  v4f oacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) oacc[kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float mrun = -3.0e38f, lrun = 0.f;

  // K / V^T operands are single-buffered: the next tile's K is requested right after this tile's
  // S MFMAs have consumed the registers, V^T right after the P.V MFMAs -- a full tile ahead of its
  // use either way.  The pair tiles travel TWO key tiles ahead (HBM latency under load is about one
  // tile's worth of work) in two register sets (A, B) that alternate roles tile by tile; alternating
  // instead of copying matters: a register copy waits for the load it copies.  All prefetches are
  // unconditional (clamped to the last tile) so each tile is straight-line code with exact waits.
  float4 kc[KT], vc[KT];
  float4 eA = make_float4(0.f, 0.f, 0.f, 0.f), gA = eA, mA = eA, eB = eA, gB = eA, mB = eA;
  Km4 kmc{{1u, 1u, 1u, 1u}};
  const int mlast = NP - 16;
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    kc[T] = *reinterpret_cast<const float4*>(Kh + koff + 256 * T);
    vc[T] = *reinterpret_cast<const float4*>(VT + voff + 256 * T);
  }
  if (f.km) kmc = km_load4(a.km + (size_t)b * N, N, 4 * q);
  if (f.E) ptile_lds_put(In + 0 * PT_SZ, pload(Eb, 0), N, l0, 0, tid);
  if (f.G) ptile_lds_put(In + 1 * PT_SZ, pload(Gb, 0), N, l0, 0, tid);
  if (f.M) ptile_lds_put(In + 2 * PT_SZ, pload(Mb, 0), N, l0, 0, tid);
  {
    const int m1 = min(16, mlast);
    if (f.E) eA = pload(Eb, m1);
    if (f.G) gA = pload(Gb, m1);
    if (f.M) mA = pload(Mb, m1);
  }
  __builtin_amdgcn_s_waitcnt(0);   // nothing pending at the loop header: its waits then reflect the loop alone
  __syncthreads();

  // one key tile: (pe, pg, pm) hold the pair tiles of it+1 and go to LDS at the bottom,
  // (qe, qg, qm) receive those of it+2
  auto tile = [&](const int m0, const int it, float4& pe4, float4& pg4, float4& pm4, float4& qe4, float4& qg4,
                  float4& qm4) __attribute__((always_inline)) {
    const int m1 = min(m0 + 16, mlast), m2 = min(m0 + 32, mlast);
    if (ABL_ON(a, 4)) {
    if (f.E) qe4 = pload(Eb, m2);
    if (f.G) qg4 = pload(Gb, m2);
    if (f.M) qm4 = pload(Mb, m2);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetches up here: hipcc otherwise sinks them past the MFMAs
    const float* Et = In + (it & 1) * 3 * PT_SZ;
    const float* Gt = Et + PT_SZ;
    const float* Mt = Gt + PT_SZ;
    float* Ht = Hout + (it & 1) * PT_SZ;
    // ---- S^T[m][l] = sum_k K[m][k] Q[l][k] ----
    v4f s = {0.f, 0.f, 0.f, 0.f};
    if (ABL_ON(a, 8))
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      s = MFMA(kc[T].x, Qr[4 * T + 0], s);
      s = MFMA(kc[T].y, Qr[4 * T + 1], s);
      s = MFMA(kc[T].z, Qr[4 * T + 2], s);
      s = MFMA(kc[T].w, Qr[4 * T + 3], s);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL_ON(a, 1))
#pragma unroll
    for (int T = 0; T < KT; ++T) kc[T] = *reinterpret_cast<const float4*>(Kh + (size_t)m1 * D + koff + 256 * T);
    __builtin_amdgcn_sched_barrier(0);
    float x[4], pa[4];
    float tmax = -3.0e38f;
    const int po4 = h * PT_PL + pt_off(ll, 4 * q);   // this lane's keys 4q..4q+3 of row ll: one 16-byte access
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = e4, m4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (f.E) e4 = *reinterpret_cast<const float4*>(Et + po4);
    if (f.G) g4 = *reinterpret_cast<const float4*>(Gt + po4);
    if (f.M) m4 = *reinterpret_cast<const float4*>(Mt + po4);
    const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
    float hv4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * q + r;
      const bool valid = m < N;
      float ah = s[r] * a.scale;
      if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
      const size_t gi = (((size_t)b * N + lc) * N + min(m, N - 1)) * AH + h;
      const float hv = ah + ev[r];
      hv4[r] = hv;                                         // H_hat: post-clip, pre-mask (egt_layers.py:85-86)
      const float kadd = kmc.v[r] ? 0.0f : -EGT_NEG;
      const float add = mask_add(a, f, kadd, mv[r], gi);
      x[r] = valid ? hv + add : -3.0e38f;
      pa[r] = gated ? egt_sigmoid(gv[r] + add) : 1.0f;    // gate (multiplied into p below)
      tmax = fmaxf(tmax, x[r]);
    }
    *reinterpret_cast<float4*>(Ht + po4) = make_float4(hv4[0], hv4[1], hv4[2], hv4[3]);
    if (f.km) {   // key mask of the next tile, into the registers this tile just finished with
      __builtin_amdgcn_sched_barrier(0);
      kmc = km_load4(a.km + (size_t)b * N, N, m1 + 4 * q);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- online softmax over the key axis: in-lane over r, then across q ----
    tmax = pair_max_q(tmax);
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __expf(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pe = (x[r] > -2.9e38f) ? __expf(x[r] - mnew) : 0.f;
      psum += pe;
      pa[r] *= pe;
    }
    psum = pair_sum_q(psum);
    lrun = fmaf(lrun, alpha, psum);
    mrun = mnew;
    // ---- O^T[k][l] = alpha * O^T + sum_m V^T[k][m] P^T[m][l]  (contraction order m = 4q + t) ----
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      v4f o = oacc[kt];
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
      if (ABL_ON(a, 8)) {
      o = MFMA(vc[kt].x, pa[0], o);
      o = MFMA(vc[kt].y, pa[1], o);
      o = MFMA(vc[kt].z, pa[2], o);
      o = MFMA(vc[kt].w, pa[3], o);
      }
      oacc[kt] = o;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL_ON(a, 1))
#pragma unroll
    for (int T = 0; T < KT; ++T) vc[T] = *reinterpret_cast<const float4*>(VT + (size_t)m1 * D + voff + 256 * T);
    __builtin_amdgcn_sched_barrier(0);
    {   // tile it+1 into the other LDS buffer (its last readers passed the previous barrier)
      float* nx = In + ((it + 1) & 1) * 3 * PT_SZ;
      if (f.E) ptile_lds_put(nx, pe4, N, l0, m1, tid);
      if (f.G) ptile_lds_put(nx + PT_SZ, pg4, N, l0, m1, tid);
      if (f.M) ptile_lds_put(nx + 2 * PT_SZ, pm4, N, l0, m1, tid);
    }
    if (ABL_ON(a, 16)) __syncthreads();
    {   // H_hat tile out: whole 512-byte runs
      const int row = tid >> 5, m = (tid & 31) >> 1;
      if (l0 + row < N && m0 + m < N && ABL_ON(a, 2)) {
        const float* p = Ht + (tid & 1) * 4 * PT_PL + pt_off(row, m);
        *reinterpret_cast<float4*>(Hb + ptile_off(prow, N, m0, tid)) = make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
      }
    }
  };
  // pairs of tiles in the loop, an odd last tile outside it: a conditional second tile inside the
  // loop would put the first tile's pending loads on a (never taken) path to the loop header and
  // make hipcc drain every prefetch there
  int m0 = 0;
  for (; m0 + 16 < NP; m0 += 32) {
    tile(m0, 0, eA, gA, mA, eB, gB, mB);
    tile(m0 + 16, 1, eB, gB, mB, eA, gA, mA);
  }
  if (m0 < NP) tile(m0, 0, eA, gA, mA, eB, gB, mB);
  // ---- finalize: V_att[l][k*8+h] = O[l][k] / l_run ; row statistics for the backward ----
  if (l < N) {
    const float inv = 1.0f / lrun;
    float* vo = a.v_att + ((size_t)b * N + l) * DH + h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vo[(16 * kt + 4 * q + r) * AH] = oacc[kt][r] * inv;
    if (q == 0) {
      float* rs = a.rowstats + (((size_t)b * N + l) * AH + h) * 4;
      rs[0] = mrun; rs[1] = lrun; rs[2] = 0.f; rs[3] = 0.f;
    }
  }
}

// ---- forward, two key tiles per iteration ------------------------------------------------------
// Same decomposition as k_attn_mfma_fwd, but one iteration covers 32 keys: two independent
// S^T / softmax chains per wave (the scheduler interleaves them), one barrier and one online-softmax
// rescale per 32 keys, pair tiles one iteration (= two tiles) ahead in registers -- no role swap
// needed.  ~190 VGPRs, so ONE workgroup per CU: the variant for grids that cannot give a CU two
// workgroups anyway (B * N/16 <= 2 x 256 on MI355X).
template <int D, int V>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_fwd2(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* In = sm;                        // [2][2 sub-tiles][E | G | M][PT_SZ]
  float* Hout = sm + 2 * 2 * 3 * PT_SZ;  // [2][2][PT_SZ]
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ll = lane & 15, q = lane >> 4;
  const int N = a.N, NP = a.NP;
  const int ltiles = NP / 16;
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / ltiles, l0 = (wg % ltiles) * 16;
  const int l = l0 + ll, lc = min(l, N - 1);
  const Feat<V> f(a);
  const bool gated = f.G, clip = f.clip;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const float* Kh = a.pk + PK_KH * arr + ((size_t)b * AH + h) * NP * D;
  const float* VT = a.pk + PK_VT * arr + ((size_t)b * AH + h) * D * NP;
  const uint32_t koff = ll * 16 + 4 * q;
  const uint32_t prow = ptile_rowoff(N, l0, tid);
  const float* Eb = ptile_base(a.E, b, N, l0);
  const float* Gb = ptile_base(a.G, b, N, l0);
  const float* Mb = ptile_base(a.M, b, N, l0);
  float* Hb = const_cast<float*>(ptile_base(a.h_hat, b, N, l0));
  auto pload = [&](const float* base, int col0) __attribute__((always_inline)) {
    return *reinterpret_cast<const float4*>(base + ptile_off(prow, N, col0, tid));
  };
  float Qr[4 * KT];
  {
    const float* qrow = a.qkv + ((size_t)b * N + lc) * 3 * DH + h;
#pragma unroll
    for (int t = 0; t < 4 * KT; ++t) Qr[t] = qrow[(16 * (t >> 2) + 4 * q + (t & 3)) * AH];
  }
  v4f oacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) oacc[kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  float mrun = -3.0e38f, lrun = 0.f;
  const int mlast = NP - 16;
  auto clampm = [&](int m) { return min(m, mlast); };   // tiles past the end re-read the last one and are masked off
  float4 kc[2][KT], vc[2][KT];
  float4 pe[2], pg[2], pm[2];
  Km4 kmc[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int mk = clampm(16 * kb);
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      kc[kb][T] = *reinterpret_cast<const float4*>(Kh + (size_t)mk * D + koff + 256 * T);
      vc[kb][T] = *reinterpret_cast<const float4*>(VT + (size_t)mk * D + koff + 256 * T);
    }
    kmc[kb] = Km4{{1u, 1u, 1u, 1u}};
    if (f.km) kmc[kb] = km_load4(a.km + (size_t)b * N, N, mk + 4 * q);
    pe[kb] = pg[kb] = pm[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f.E) ptile_lds_put(In + (kb * 3 + 0) * PT_SZ, pload(Eb, mk), N, l0, mk, tid);
    if (f.G) ptile_lds_put(In + (kb * 3 + 1) * PT_SZ, pload(Gb, mk), N, l0, mk, tid);
    if (f.M) ptile_lds_put(In + (kb * 3 + 2) * PT_SZ, pload(Mb, mk), N, l0, mk, tid);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  for (int m0 = 0, it = 0; m0 < NP; m0 += 32, ++it) {
    int mn[2];   // the two key tiles of the NEXT iteration
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      mn[kb] = clampm(m0 + 32 + 16 * kb);
      if (f.E) pe[kb] = pload(Eb, mn[kb]);
      if (f.G) pg[kb] = pload(Gb, mn[kb]);
      if (f.M) pm[kb] = pload(Mb, mn[kb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* Inb = In + (it & 1) * 6 * PT_SZ;
    float* Hb2 = Hout + (it & 1) * 2 * PT_SZ;
    // ---- S^T of both key tiles ----
    v4f s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        s[kb] = MFMA(kc[kb][T].x, Qr[4 * T + 0], s[kb]);
        s[kb] = MFMA(kc[kb][T].y, Qr[4 * T + 1], s[kb]);
        s[kb] = MFMA(kc[kb][T].z, Qr[4 * T + 2], s[kb]);
        s[kb] = MFMA(kc[kb][T].w, Qr[4 * T + 3], s[kb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int T = 0; T < KT; ++T) kc[kb][T] = *reinterpret_cast<const float4*>(Kh + (size_t)mn[kb] * D + koff + 256 * T);
    __builtin_amdgcn_sched_barrier(0);
    float x[2][4], pa[2][4];
    float tmax = -3.0e38f;
    const int po4 = h * PT_PL + pt_off(ll, 4 * q);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int mt0 = m0 + 16 * kb;   // may lie past NP on the last iteration: every key then fails `valid`
      float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = e4, m4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (f.E) e4 = *reinterpret_cast<const float4*>(Inb + (kb * 3 + 0) * PT_SZ + po4);
      if (f.G) g4 = *reinterpret_cast<const float4*>(Inb + (kb * 3 + 1) * PT_SZ + po4);
      if (f.M) m4 = *reinterpret_cast<const float4*>(Inb + (kb * 3 + 2) * PT_SZ + po4);
      const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
      float hv4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt0 + 4 * q + r;
        const bool valid = m < N;
        float ah = s[kb][r] * a.scale;
        if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
        const size_t gi = (((size_t)b * N + lc) * N + min(m, N - 1)) * AH + h;
        const float hv = ah + ev[r];
        hv4[r] = hv;
        const float kadd = kmc[kb].v[r] ? 0.0f : -EGT_NEG;
        const float add = mask_add(a, f, kadd, mv[r], gi);
        x[kb][r] = valid ? hv + add : -3.0e38f;
        pa[kb][r] = gated ? egt_sigmoid(gv[r] + add) : 1.0f;
        tmax = fmaxf(tmax, x[kb][r]);
      }
      *reinterpret_cast<float4*>(Hb2 + kb * PT_SZ + po4) = make_float4(hv4[0], hv4[1], hv4[2], hv4[3]);
    }
    if (f.km) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kmc[kb] = km_load4(a.km + (size_t)b * N, N, mn[kb] + 4 * q);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- online softmax over the 32 keys ----
    tmax = pair_max_q(tmax);
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __expf(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pexp = (x[kb][r] > -2.9e38f) ? __expf(x[kb][r] - mnew) : 0.f;
        psum += pexp;
        pa[kb][r] *= pexp;
      }
    psum = pair_sum_q(psum);
    lrun = fmaf(lrun, alpha, psum);
    mrun = mnew;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      v4f o = oacc[kt];
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        o = MFMA(vc[kb][kt].x, pa[kb][0], o);
        o = MFMA(vc[kb][kt].y, pa[kb][1], o);
        o = MFMA(vc[kb][kt].z, pa[kb][2], o);
        o = MFMA(vc[kb][kt].w, pa[kb][3], o);
      }
      oacc[kt] = o;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int T = 0; T < KT; ++T) vc[kb][T] = *reinterpret_cast<const float4*>(VT + (size_t)mn[kb] * D + koff + 256 * T);
    __builtin_amdgcn_sched_barrier(0);
    {
      float* nx = In + ((it + 1) & 1) * 6 * PT_SZ;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (f.E) ptile_lds_put(nx + (kb * 3 + 0) * PT_SZ, pe[kb], N, l0, mn[kb], tid);
        if (f.G) ptile_lds_put(nx + (kb * 3 + 1) * PT_SZ, pg[kb], N, l0, mn[kb], tid);
        if (f.M) ptile_lds_put(nx + (kb * 3 + 2) * PT_SZ, pm[kb], N, l0, mn[kb], tid);
      }
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {   // H_hat tiles out: whole 512-byte runs
      const int row = tid >> 5, m = (tid & 31) >> 1, mt0 = m0 + 16 * kb;
      if (l0 + row < N && mt0 + m < N) {
        const float* p = Hb2 + kb * PT_SZ + (tid & 1) * 4 * PT_PL + pt_off(row, m);
        *reinterpret_cast<float4*>(Hb + ptile_off(prow, N, mt0, tid)) = make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
      }
    }
  }
  if (l < N) {
    const float inv = 1.0f / lrun;
    float* vo = a.v_att + ((size_t)b * N + l) * DH + h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vo[(16 * kt + 4 * q + r) * AH] = oacc[kt][r] * inv;
    if (q == 0) {
      float* rs = a.rowstats + (((size_t)b * N + l) * AH + h) * 4;
      rs[0] = mrun; rs[1] = lrun; rs[2] = 0.f; rs[3] = 0.f;
    }
  }
}

// ================================================================= backward =====
// Launches (flash-attention style, the [N,N,H] probabilities are recomputed):
//   k_attn_pack        : head-major operand arrays of Q, K, V, dV_att
//   (delta[l,h] = sum_k dO[l,k,h] * O[l,k,h] -> rowstats[...][3] is computed by k_attn_pack's dO section)
//   k_attn_mfma_bwd_kv : workgroup = (graph, 16-key tile), wave = head, walks the query tiles;
//                        K/V fragments of the key tile live in registers; per tile S = Q.K^T and
//                        dP = dO.V^T on MFMA, softmax/gate/clip backward on the VALU, then
//                        dV^T += dO^T.A and dK^T += Q^T.dA on MFMA with the probabilities as B
//                        operands in place; writes dE, dG and dA = dH*c*scale
//   k_attn_mfma_bwd_q  : workgroup = (graph, 16 query rows), wave = head, walks the key tiles:
//                        dQ^T += K^T.dA^T on MFMA from the dA tensor
// Lane (mm = lane&15, q): key m0 + mm; in every query tile rows l0 + 4q + r.
template <int D, int V>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_bwd_kv(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mm = lane & 15, q = lane >> 4;
  const int N = a.N, NP = a.NP;
  const int mtiles = NP / 16;
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / mtiles, m0 = (wg % mtiles) * 16;
  const int m = m0 + mm, mc = min(m, N - 1);
  const bool mvalid = m < N;
  const Feat<V> f(a);
  const bool gated = f.G, clip = f.clip;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const size_t hb = ((size_t)b * AH + h) * NP * D;
  const float* Kh = a.pk + PK_KH * arr + hb;
  const float* Vh = a.pk + PK_VH * arr + hb;
  const float* Qh = a.pk + PK_QH * arr + hb;
  const float* Oh = a.pk + PK_OH * arr + hb;
  float id4[4];   // identity slices for the in-register transposes
#pragma unroll
  for (int u = 0; u < 4; ++u) id4[u] = (mm == 4 * q + u) ? 1.0f : 0.0f;

  // K / V fragments of this lane's key (B operands of S = Q.K^T and dP = dO.V^T); padded keys are zero
  float4 Kr[KT], Vr[KT];
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    Kr[T] = *reinterpret_cast<const float4*>(Kh + (size_t)m0 * D + 256 * T + mm * 16 + 4 * q);
    Vr[T] = *reinterpret_cast<const float4*>(Vh + (size_t)m0 * D + 256 * T + mm * 16 + 4 * q);
  }
  const float kadd = (f.km && a.km[(size_t)b * N + mc] == 0) ? -EGT_NEG : 0.0f;
  v4f dKacc[KT], dVacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) { dKacc[kt] = (v4f){0.f, 0.f, 0.f, 0.f}; dVacc[kt] = (v4f){0.f, 0.f, 0.f, 0.f}; }

  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* In = sm;                     // [2][E | G | M | dH_ext][PT_SZ]
  float* Out = sm + 2 * 4 * PT_SZ;    // [2][dE | dG | dA][PT_SZ]
  const int tid = threadIdx.x;
  float4 pe4 = make_float4(0.f, 0.f, 0.f, 0.f), pg4 = pe4, pm4 = pe4, px4 = pe4;
  if (f.E) ptile_lds_put(In + 0 * PT_SZ, ptile_gload(a.E, b, N, 0, m0, tid), N, 0, m0, tid);
  if (f.G) ptile_lds_put(In + 1 * PT_SZ, ptile_gload(a.G, b, N, 0, m0, tid), N, 0, m0, tid);
  if (f.M) ptile_lds_put(In + 2 * PT_SZ, ptile_gload(a.M, b, N, 0, m0, tid), N, 0, m0, tid);
  if (f.X) ptile_lds_put(In + 3 * PT_SZ, ptile_gload(a.d_h_ext, b, N, 0, m0, tid), N, 0, m0, tid);
  float4 qan[KT], oan[KT], stn[4];   // row operands / statistics of the next query tile
#pragma unroll
  for (int T = 0; T < KT; ++T) {
    qan[T] = *reinterpret_cast<const float4*>(Qh + 256 * T + mm * 16 + 4 * q);
    oan[T] = *reinterpret_cast<const float4*>(Oh + 256 * T + mm * 16 + 4 * q);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    stn[r] = *reinterpret_cast<const float4*>(a.rowstats + (((size_t)b * N + min(4 * q + r, N - 1)) * AH + h) * 4);
  __syncthreads();

  for (int l0 = 0, it = 0; l0 < NP; l0 += 16, ++it) {
    const bool more = l0 + 16 < NP;
    // ---- loads, oldest first: transposed operands of THIS tile (consumed after the elementwise
    //      phase), then the row operands / statistics / pair tiles of the NEXT tile ----
    float4 qa[KT], oa[KT], st[4];
#pragma unroll
    for (int T = 0; T < KT; ++T) { qa[T] = qan[T]; oa[T] = oan[T]; }
    // The transposed operands (A rows are channels: Q^T, dO^T of this tile) come from the row
    // operands through the matrix core itself: D = X . I with the identity split over the four
    // contraction steps (step u, B[k = q][j] = [j == 4q + u]) leaves lane (mm, q) with
    // X[l0 + 4q + r][16T + mm] -- exact (one non-zero product per output), 32 extra MFMAs on a
    // pipe that is 30 % busy instead of a second 8 KB operand fetch per wave and tile.
    float4 qt[KT], ot[KT];
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      v4f tq = {0.f, 0.f, 0.f, 0.f}, to = {0.f, 0.f, 0.f, 0.f};
      tq = MFMA(qa[T].x, id4[0], tq);   to = MFMA(oa[T].x, id4[0], to);
      tq = MFMA(qa[T].y, id4[1], tq);   to = MFMA(oa[T].y, id4[1], to);
      tq = MFMA(qa[T].z, id4[2], tq);   to = MFMA(oa[T].z, id4[2], to);
      tq = MFMA(qa[T].w, id4[3], tq);   to = MFMA(oa[T].w, id4[3], to);
      qt[T] = make_float4(tq[0], tq[1], tq[2], tq[3]);
      ot[T] = make_float4(to[0], to[1], to[2], to[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) st[r] = stn[r];
    size_t gi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gi[r] = (((size_t)b * N + min(l0 + 4 * q + r, N - 1)) * N + mc) * AH + h;
    if (more && ABL_ON(a, 1)) {
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        qan[T] = *reinterpret_cast<const float4*>(Qh + (size_t)(l0 + 16) * D + 256 * T + mm * 16 + 4 * q);   // A rows are query rows
        oan[T] = *reinterpret_cast<const float4*>(Oh + (size_t)(l0 + 16) * D + 256 * T + mm * 16 + 4 * q);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stn[r] = *reinterpret_cast<const float4*>(a.rowstats + (((size_t)b * N + min(l0 + 16 + 4 * q + r, N - 1)) * AH + h) * 4);
    }
    if (more && ABL_ON(a, 4)) {
      if (f.E) pe4 = ptile_gload(a.E, b, N, l0 + 16, m0, tid);
      if (f.G) pg4 = ptile_gload(a.G, b, N, l0 + 16, m0, tid);
      if (f.M) pm4 = ptile_gload(a.M, b, N, l0 + 16, m0, tid);
      if (f.X) px4 = ptile_gload(a.d_h_ext, b, N, l0 + 16, m0, tid);
    }
    const float* Et = In + (it & 1) * 4 * PT_SZ;
    const float* Gt = Et + PT_SZ;
    const float* Mt = Gt + PT_SZ;
    const float* Xt = Mt + PT_SZ;
    float* dEt = Out + (it & 1) * 3 * PT_SZ;
    float* dGt = dEt + PT_SZ;
    float* dAt = dGt + PT_SZ;
    // ---- S[l][m] = sum_k Q[l][k] K[m][k] ; dP[l][m] = sum_k dO[l][k] V[m][k] ----
    v4f s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
    if (ABL_ON(a, 8))
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      s = MFMA(qa[T].x, Kr[T].x, s);   dp = MFMA(oa[T].x, Vr[T].x, dp);
      s = MFMA(qa[T].y, Kr[T].y, s);   dp = MFMA(oa[T].y, Vr[T].y, dp);
      s = MFMA(qa[T].z, Kr[T].z, s);   dp = MFMA(oa[T].z, Vr[T].z, dp);
      s = MFMA(qa[T].w, Kr[T].w, s);   dp = MFMA(oa[T].w, Vr[T].w, dp);
    }
    float at[4], da[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int l = l0 + 4 * q + r;
      const bool valid = mvalid && l < N;
      const float araw = s[r] * a.scale;
      float ah = araw, inr = 1.0f;
      if (clip) {
        inr = (araw >= a.clip_lo && araw <= a.clip_hi) ? 1.0f : 0.0f;
        ah = fminf(fmaxf(araw, a.clip_lo), a.clip_hi);
      }
      const int po = h * PT_PL + pt_off(4 * q + r, mm);
      const float add = mask_add(a, f, kadd, f.M ? Mt[po] : 1.f, gi[r]);
      const float xv = ah + (f.E ? Et[po] : 0.f) + add;
      const float S = valid ? __expf(xv - st[r].x) * __builtin_amdgcn_rcpf(st[r].y) : 0.f;
      const float g = gated ? egt_sigmoid(Gt[po] + add) : 1.0f;
      const float dAt_ = dp[r];
      float dH = S * (dAt_ * g - st[r].w) + (f.X ? Xt[po] : 0.f);
      if (!valid) dH = 0.f;
      da[r] = dH * inr * a.scale;
      at[r] = S * g;
      dEt[po] = dH;
      dGt[po] = gated ? dAt_ * S * g * (1.0f - g) : 0.f;
      dAt[po] = da[r];
    }
    // ---- dV^T[k][m] += sum_l dO[l][k] A[l][m] ; dK^T[k][m] += sum_l Q[l][k] dA[l][m] ----
    if (ABL_ON(a, 8))
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      v4f dv = dVacc[kt], dk = dKacc[kt];
      dv = MFMA(ot[kt].x, at[0], dv);   dk = MFMA(qt[kt].x, da[0], dk);
      dv = MFMA(ot[kt].y, at[1], dv);   dk = MFMA(qt[kt].y, da[1], dk);
      dv = MFMA(ot[kt].z, at[2], dv);   dk = MFMA(qt[kt].z, da[2], dk);
      dv = MFMA(ot[kt].w, at[3], dv);   dk = MFMA(qt[kt].w, da[3], dk);
      dVacc[kt] = dv; dKacc[kt] = dk;
    }
    if (more) {
      float* nx = In + ((it + 1) & 1) * 4 * PT_SZ;
      if (f.E) ptile_lds_put(nx, pe4, N, l0 + 16, m0, tid);
      if (f.G) ptile_lds_put(nx + PT_SZ, pg4, N, l0 + 16, m0, tid);
      if (f.M) ptile_lds_put(nx + 2 * PT_SZ, pm4, N, l0 + 16, m0, tid);
      if (f.X) ptile_lds_put(nx + 3 * PT_SZ, px4, N, l0 + 16, m0, tid);
    }
    if (ABL_ON(a, 16)) __syncthreads();
    if (ABL_ON(a, 2)) {
    if (f.E) ptile_gstore(a.d_E, dEt, b, N, l0, m0, tid);
    if (f.G) ptile_gstore(a.d_G, dGt, b, N, l0, m0, tid);
    ptile_gstore(a.ws_dA, dAt, b, N, l0, m0, tid);
    }
  }
  if (mvalid) {
    float* o = a.d_qkv + ((size_t)b * N + m) * 3 * DH + h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + 4 * q + r;
        o[DH + k * AH] = dKacc[kt][r];
        o[2 * DH + k * AH] = dVacc[kt][r];
      }
  }
}

// Lane (ll = lane&15, q): query row l0 + ll; keys m0 + 4q + t as the contraction index.
// TB key tiles per iteration (one barrier per 16*TB keys; K^T operands single-buffered: the next
// block's are requested right after the MFMAs have read the registers).  TB = 4 for grids that
// cannot give a CU two workgroups, TB = 1 (76 VGPRs, several workgroups per CU) otherwise.
template <int D, int TB>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_bwd_q(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ll = lane & 15, q = lane >> 4;
  const int N = a.N, NP = a.NP;
  const int ltiles = NP / 16;
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / ltiles, l0 = (wg % ltiles) * 16;
  const int l = l0 + ll;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const float* KTp = a.pk + PK_KT * arr + ((size_t)b * AH + h) * D * NP;
  const uint32_t koff = ll * 16 + 4 * q;
  v4f dQacc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) dQacc[kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dA tiles [2][TB][PT_SZ]
  const int tid = threadIdx.x;
  const int mlast = NP - 16;
  float4 kc[TB][KT], pa4[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    const int mk = min(16 * tb, mlast);   // tiles past the end: K^T of the last tile against dA columns that are zero
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) kc[tb][kt] = *reinterpret_cast<const float4*>(KTp + (size_t)mk * D + koff + 256 * kt);
    ptile_lds_put(sm + tb * PT_SZ, ptile_gload(a.ws_dA, b, N, l0, 16 * tb, tid), N, l0, 16 * tb, tid);
    pa4[tb] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int m0 = 0, it = 0; m0 < NP; m0 += 16 * TB, ++it) {
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) pa4[tb] = ptile_gload(a.ws_dA, b, N, l0, m0 + 16 * (TB + tb), tid);   // clamped, zeroed at the LDS store
    __builtin_amdgcn_sched_barrier(0);
    const float* At = sm + (it & 1) * TB * PT_SZ;
    float4 da4[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) da4[tb] = *reinterpret_cast<const float4*>(At + tb * PT_SZ + h * PT_PL + pt_off(ll, 4 * q));
    // dQ^T[k][l] += sum_m K^T[k][m] dA[l][m]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      v4f dq = dQacc[kt];
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        dq = MFMA(kc[tb][kt].x, da4[tb].x, dq);
        dq = MFMA(kc[tb][kt].y, da4[tb].y, dq);
        dq = MFMA(kc[tb][kt].z, da4[tb].z, dq);
        dq = MFMA(kc[tb][kt].w, da4[tb].w, dq);
      }
      dQacc[kt] = dq;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int mk = min(m0 + 16 * (TB + tb), mlast);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) kc[tb][kt] = *reinterpret_cast<const float4*>(KTp + (size_t)mk * D + koff + 256 * kt);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
      ptile_lds_put(sm + (((it + 1) & 1) * TB + tb) * PT_SZ, pa4[tb], N, l0, m0 + 16 * (TB + tb), tid);
    __syncthreads();
  }
  if (l < N) {
    float* o = a.d_qkv + ((size_t)b * N + l) * 3 * DH + h;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(16 * kt + 4 * q + r) * AH] = dQacc[kt][r];
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

static int np_of(int N) { return (N + 15) & ~15; }

// forward uses the first 2 packed arrays, backward all 8 followed by the dA tensor
extern "C" size_t egt_attn_mfma_workspace_bytes(const egt_attn_desc* d) {
  if (!egt_attn_mfma_supported(d, 0)) return 0;
  const size_t arr = (size_t)d->B * AH * np_of(d->N) * d->d;
  return (PK_COUNT * arr + (size_t)d->B * d->N * d->N * AH) * sizeof(float);
}

extern "C" size_t egt_attn_mfma_fwd_workspace_bytes(const egt_attn_desc* d) {
  if (!egt_attn_mfma_supported(d, 0)) return 0;
  return (size_t)2 * d->B * AH * np_of(d->N) * d->d * sizeof(float);
}

// experiment switches, read once per process (no getenv on the launch path)
struct EgtAttnEnv { int ablate, generic, fwd2 /* -1: auto */, bwdq4; };
static const EgtAttnEnv& attn_env() {
  static const EgtAttnEnv e = [] {
    EgtAttnEnv v{};
    const char* g = getenv("EGT_ATTN_ABLATE");
    v.ablate = g ? atoi(g) : 0;
#ifndef EGT_ATTN_ABLATION
    if (v.ablate) { fprintf(stderr, "[egt] EGT_ATTN_ABLATE ignored: build with EGT_ATTN_FLAGS=-DEGT_ATTN_ABLATION\n"); v.ablate = 0; }
#endif
    v.generic = getenv("EGT_ATTN_GENERIC") != nullptr;
    const char* f2 = getenv("EGT_ATTN_FWD2");
    v.fwd2 = f2 ? (atoi(f2) != 0) : -1;
    const char* e4 = getenv("EGT_ATTN_BWDQ4");
    v.bwdq4 = e4 ? (atoi(e4) != 0) : 0;
    return v;
  }();
  return e;
}

static int fill(const egt_attn_desc* desc, const void* qkv, const void* E, const void* G,
                const uint8_t* key_mask, const void* attn_mask, const uint8_t* rand_mask, void* workspace,
                AttnMfmaArgs& a) {
  if (!egt_attn_mfma_supported(desc, 0)) EGT_FAIL(EGT_E_SHAPE, "configuration not covered by the MFMA inner-op kernel");
  if (!qkv || !workspace) EGT_FAIL(EGT_E_NULL, "qkv/workspace is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  a = AttnMfmaArgs{};
  a.B = desc->B; a.N = desc->N; a.NP = np_of(desc->N); a.d = desc->d; a.flags = desc->flags;
  a.clip_lo = desc->clip_lo; a.clip_hi = desc->clip_hi;
  a.scale = 1.0f / sqrtf((float)desc->d);
  a.guard = attn_env().ablate;
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
  a.pk = (float*)workspace;
  return EGT_OK;
}

template <int D>
static void launch_pack(const AttnMfmaArgs& a, hipStream_t st) {
  const size_t lds = (size_t)16 * (D * AH + 4) * 4;
  EGT_MAX_LDS_ONCE(k_attn_pack<D>);
  EGT_LAUNCH("k_attn_pack", k_attn_pack<D>, dim3(a.B * (a.NP / 16) * (a.pack_bwd ? 4 : 2)), dim3(256), lds, st, a);
}

// 1 / 2: the straight-line instances (see Feat), 0: the run-time-switched one
static int variant_of(const AttnMfmaArgs& a, bool bwd) {
  const bool main_cfg = a.E && (a.flags & EGT_F_GATE_INPUT) && a.G && !a.M && a.km && (a.flags & EGT_F_CLIP) && !a.rm &&
                        (!bwd || (a.d_h_ext && a.d_E && a.d_G));
  if (!main_cfg || attn_env().generic) return 0;
  return a.rng_rm ? 2 : 1;
}

template <int D, int V>
static void launch_fwd_v(const AttnMfmaArgs& a, hipStream_t st) {
  // small grids (no CU would get two workgroups anyway): two key tiles per iteration
  const int f2 = attn_env().fwd2;
  const bool two = f2 >= 0 ? f2 != 0 : (a.B * (a.NP / 16) <= 512);
  if (two) {
    EGT_MAX_LDS_ONCE(k_attn_mfma_fwd2<D, V>);
    EGT_LAUNCH("k_attn_mfma_fwd", (k_attn_mfma_fwd2<D, V>), dim3(a.B * (a.NP / 16)), dim3(512), (size_t)16 * PT_SZ * 4, st, a);
    return;
  }
  EGT_MAX_LDS_ONCE(k_attn_mfma_fwd<D, V>);
  EGT_LAUNCH("k_attn_mfma_fwd", (k_attn_mfma_fwd<D, V>), dim3(a.B * (a.NP / 16)), dim3(512), (size_t)8 * PT_SZ * 4, st, a);
}
template <int D>
static void launch_fwd(const AttnMfmaArgs& a, hipStream_t st) {
  launch_pack<D>(a, st);
  switch (variant_of(a, false)) {
    case 1: launch_fwd_v<D, 1>(a, st); break;
    case 2: launch_fwd_v<D, 2>(a, st); break;
    default: launch_fwd_v<D, 0>(a, st); break;
  }
}

extern "C" int egt_attn_mfma_fwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                                 const uint8_t* rand_mask, void* v_att, void* h_hat, void* rowstats,
                                 void* workspace, void* stream) {
  AttnMfmaArgs a;
  int rc = fill(desc, qkv, E, G, key_mask, attn_mask, rand_mask, workspace, a);
  if (rc) return rc;
  if (!v_att || !h_hat || !rowstats) EGT_FAIL(EGT_E_NULL, "v_att/h_hat/rowstats is NULL");
  a.v_att = (float*)v_att; a.h_hat = (float*)h_hat; a.rowstats = (float*)rowstats;
  a.pack_bwd = 0;
  switch (desc->d) {
    case 16: launch_fwd<16>(a, (hipStream_t)stream); break;
    case 32: launch_fwd<32>(a, (hipStream_t)stream); break;
    default: launch_fwd<64>(a, (hipStream_t)stream); break;
  }
  EGT_HIP_LAUNCH_CHECK("egt_attn_mfma_fwd");
  return EGT_OK;
}

template <int D, int V>
static void launch_bwd_kv_v(const AttnMfmaArgs& a, hipStream_t st) {
  EGT_MAX_LDS_ONCE(k_attn_mfma_bwd_kv<D, V>);
  EGT_LAUNCH("k_attn_mfma_bwd_kv", (k_attn_mfma_bwd_kv<D, V>), dim3(a.B * (a.NP / 16)), dim3(512), (size_t)14 * PT_SZ * 4, st, a);
}
template <int D>
static void launch_bwd(const AttnMfmaArgs& a, hipStream_t st) {
  launch_pack<D>(a, st);   // (also writes delta = sum_k dO*O into rowstats[..][3])
  switch (variant_of(a, true)) {
    case 1: launch_bwd_kv_v<D, 1>(a, st); break;
    case 2: launch_bwd_kv_v<D, 2>(a, st); break;
    default: launch_bwd_kv_v<D, 0>(a, st); break;
  }
  {
    const bool four = attn_env().bwdq4 != 0;   // four key tiles per iteration: measured equal to one (49 vs 50 us at B = 8, slower at B = 32), kept as an experiment switch
    if (four) {
      EGT_MAX_LDS_ONCE(k_attn_mfma_bwd_q<D, 4>);
      EGT_LAUNCH("k_attn_mfma_bwd_q", (k_attn_mfma_bwd_q<D, 4>), dim3(a.B * (a.NP / 16)), dim3(512), (size_t)8 * PT_SZ * 4, st, a);
    } else {
      EGT_LAUNCH("k_attn_mfma_bwd_q", (k_attn_mfma_bwd_q<D, 1>), dim3(a.B * (a.NP / 16)), dim3(512), (size_t)2 * PT_SZ * 4, st, a);
    }
  }
}

// rowstats is read AND written (slot 3 receives delta); workspace: egt_attn_mfma_workspace_bytes()
extern "C" int egt_attn_mfma_bwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                                 const uint8_t* rand_mask, const void* v_att, void* rowstats,
                                 const void* d_v_att, const void* d_h_ext, void* d_qkv, void* d_E,
                                 void* d_G, void* workspace, void* stream) {
  AttnMfmaArgs a;
  int rc = fill(desc, qkv, E, G, key_mask, attn_mask, rand_mask, workspace, a);
  if (rc) return rc;
  if (!v_att || !rowstats || !d_v_att || !d_qkv) EGT_FAIL(EGT_E_NULL, "v_att/rowstats/d_v_att/d_qkv is NULL");
  if ((desc->flags & EGT_F_EDGE_INPUT) && !d_E) EGT_FAIL(EGT_E_NULL, "edge_input set but d_E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !d_G) EGT_FAIL(EGT_E_NULL, "gate_input set but d_G is NULL");
  a.v_att_in = (const float*)v_att; a.rowstats = (float*)rowstats;
  a.d_v_att = (const float*)d_v_att; a.d_h_ext = (const float*)d_h_ext;
  a.d_qkv = (float*)d_qkv;
  a.d_E = (desc->flags & EGT_F_EDGE_INPUT) ? (float*)d_E : nullptr;
  a.d_G = (desc->flags & EGT_F_GATE_INPUT) ? (float*)d_G : nullptr;
  a.ws_dA = a.pk + (size_t)PK_COUNT * a.B * AH * a.NP * a.d;
  a.pack_bwd = 1;
  switch (desc->d) {
    case 16: launch_bwd<16>(a, (hipStream_t)stream); break;
    case 32: launch_bwd<32>(a, (hipStream_t)stream); break;
    default: launch_bwd<64>(a, (hipStream_t)stream); break;
  }
  EGT_HIP_LAUNCH_CHECK("egt_attn_mfma_bwd");
  return EGT_OK;
}
