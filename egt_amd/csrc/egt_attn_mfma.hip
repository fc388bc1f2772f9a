// Inner op on MFMA tiles (flash-style) for the large-head geometry (d in {16,32,64}, H = 8):
//   (V_att, H_hat) = EGT([QKV, E?, G?, M?], mask)     lib/models/egt_layers.py:57-213
// and its backward (SURVEY.md appendix A / egt_layers.py semantics under autodiff).
// This is the shape where QK^T / A.V carry the arithmetic (BASELINE config 5: N = 512, d = 64: fp32
// arithmetic intensity at the ridge), so every contraction runs on v_mfma_f32_16x16x4_f32 (exact fp32)
// and the N x N probabilities never exist outside registers.
//
// Round-4 structure (what bounds an fp32 MFMA kernel on gfx950: DESIGN.md 4.0 -- MFMA and VALU issue
// ADD on a SIMD, everything else can hide):
//   * a pack kernel rewrites Q (pre-scaled by d^-1/2), K, V, dV_att head-major and tile-major, row form
//     [B,H,NP/16,d/16,16,16] and transposed form [B,H,NP/16,d,16]: EVERY MFMA operand that comes from
//     them is one aligned 16-byte global load per four contraction steps, straight into the lane that
//     feeds the matrix core (contraction order kappa(T,u,q) = 16T + 4q + u on both operands);
//   * workgroup = (32-row block, 4 heads): 4 COMPUTE waves (one per SIMD, one head each) whose instruction
//     stream is ds_read + MFMA + VALU only, and 4 LOADER waves that own every vector-memory instruction
//     (one global_load in an fp32 MFMA stream costs its wave ~80 cycles of issue, a ds_read_b128 ~7; LDS-DMA
//     loaders beside the compute waves cost them nothing: tools/micro/mfma_stream.hip).  A compute wave
//     works on two 16-row tiles at once, so every operand fragment read from LDS feeds two MFMAs.  The
//     [N,N,8] pair tensors cross the heads <-> pairs layout change through head-major LDS planes: a
//     loader thread moves 16 bytes (4 heads of one pair) per tensor and 16 x 16 sub-tile, two iterations ahead;
//   * ONE barrier per iteration and it does not drain the vector-memory queue (s_waitcnt lgkmcnt(0) +
//     s_barrier, not __syncthreads): operand prefetches and output stores stay in flight across it;
//   * the elementwise skeleton is cut to the arithmetic: additive masks from a per-key LDS table
//     (0 / -1e9 / -3e38 for keys past N: no selects), v_med3 clip, exp2 with folded constants,
//     v_rcp sigmoid, Q pre-scaled;
//   * backward: K / V fragments and the dK / dV accumulators of two key tiles stay in registers while the
//     workgroup walks the query tiles; the Q / dO tile of a query tile is staged ONCE by LDS-DMA and read
//     in BOTH operand forms (row form b128, transposed form b32: no transposing MFMAs, no transposed
//     arrays); planes are key-major there so a lane moves its four query rows with one ds_read_b128;
//     dA = dH*c leaves in the lane's own layout (one 16-byte store) for k_attn_mfma_bwd_q, which is a
//     pure operand-streaming MFMA kernel (no LDS, no barrier).
// Attributes outside this cover (dropout, degree scalers, A_tild output, H != 8, other d) use the
// general kernels of egt_attn.hip.
#include <stdlib.h>

#include "egt_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA_(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define AH 8
#define L2E 1.4426950408889634f
#define KEY_OFF (-3.0e38f)   // additive term of key slots past N (below every masked logit, above -inf)

// packed operand arrays, each B*H*NP*d floats, in workspace order: Q rows (pre-scaled), K rows, V^T (what the forward
// reads), V rows, K^T, dO rows (the backward's); then stats2 [B,H,NP,4] and dA [B,H,NP,NP].  A forward launched with
// EGT_ATTN_WS_SHARED packs the five q/k/v arrays once and the backward, given the same workspace, adds only dO.
enum { PK_QH = 0, PK_KH, PK_VT, PK_FWD_COUNT, PK_VH = PK_FWD_COUNT, PK_KT, PK_OH, PK_COUNT };
enum { PACK_Q = 1, PACK_KH = 2, PACK_KT = 4, PACK_VT = 8, PACK_VH = 16, PACK_O = 32 };

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
  float *d_qkv, *d_E, *d_G, *ws_dA, *stats2;
  int pack_what;   // PACK_* bits
};

// Phase stamps (-DEGT_ATTN_STAMPS via EGT_ATTN_FLAGS): s_memtime deltas of every wave of workgroup 0, summed per
// phase, read back with egt_attn_mfma_read_stamps().  Compiled out otherwise.
#ifdef EGT_ATTN_STAMPS
__device__ long long g_attn_stamps[3][8][16];
#define STAMP_DECL long long st_last = __builtin_readcyclecounter(), st_acc[16] = {}
#define STAMP(i) do { const long long t__ = __builtin_readcyclecounter(); st_acc[i] += t__ - st_last; st_last = t__; } while (0)
#define STAMP_OUT(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i__ = 0; i__ < 16; ++i__) g_attn_stamps[k][threadIdx.x >> 6][i__] = st_acc[i__]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_OUT(k)
#endif
// timing ablations (results are wrong): -DEGT_ATTN_ABL=<bits>: 1 no operand DMA, 2 no pair loads / scatter, 4 no output stores,
// 8 no MFMAs, 16 no elementwise phase (probabilities = logits)
#ifndef EGT_ATTN_ABL
#define EGT_ATTN_ABL 0
#endif
#define ABL(bit) ((EGT_ATTN_ABL & (bit)) == 0)
#if EGT_ATTN_ABL & 8
__device__ __forceinline__ v4f MFMA(float a, float b, v4f c) { c[0] += a * b; return c; }
#else
#define MFMA(a, b, c) MFMA_(a, b, c)
#endif

// LDS hand-off between the waves of a workgroup WITHOUT draining vector memory: __syncthreads() is
// s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier on gfx950; the operand prefetches and the output stores of
// these kernels must stay in flight across the barrier, and nothing here communicates through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float pair_max_q(float v) {   // max over lanes l, l+16, l+32, l+48
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float pair_sum_q(float v) { return sum_xor32(sum_xor16(v)); }
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------- pack --------
// workgroup = (graph, 16 node rows), 512 threads: every section of the launch ([q | k | v | dO]) is de-interleaved
// through LDS and leaves head-major.  Channel index of the source: c = s*d*H + k*H + h (egt_layers.py:70-76).
// A thread loads 16 bytes = 4 heads of one channel k of one row and scatters them into head planes [h][row][k]
// (row stride D + 16, plane stride = 16 rows + 4: the 4 x ds_write_b32 and the 16-byte row reads are bank-conflict
// free); the next section's loads are in flight while the current one is written out (one launch = one wave of
// workgroups and, per section, one overlapped HBM round trip: the one-section-per-workgroup kernel took 18 us).
template <int D>
__global__ void __launch_bounds__(512) k_attn_pack(AttnMfmaArgs a) {
  constexpr int DH = D * AH, RS = D == 16 ? 16 : 80, RP = 16 * RS + 4, NPC = 16 * DH / 4 / 512;   // pieces per thread and section
  static_assert((16 * DH / 4) % 512 == 0, "whole pieces per thread");
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [8 heads][RP]
  const int N = a.N, NP = a.NP, tid = threadIdx.x;
  const int tiles = NP / 16;
  const int b = blockIdx.x / tiles, n0 = (blockIdx.x % tiles) * 16;
  int secs[4], nsec = 0;   // sections of this launch, in order q, k, v, dO
  if (a.pack_what & PACK_Q) secs[nsec++] = 0;
  if (a.pack_what & (PACK_KH | PACK_KT)) secs[nsec++] = 1;
  if (a.pack_what & (PACK_VT | PACK_VH)) secs[nsec++] = 2;
  if (a.pack_what & PACK_O) secs[nsec++] = 3;
  const size_t arr = (size_t)a.B * AH * NP * D;
  auto load = [&](int sec, float4 (&v)[NPC]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + 512 * i, r = p / (DH / 4), c = (p % (DH / 4)) * 4, n = n0 + r;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N)
        v[i] = sec < 3 ? *reinterpret_cast<const float4*>(a.qkv + ((size_t)b * N + n) * 3 * DH + sec * DH + c)
                       : *reinterpret_cast<const float4*>(a.d_v_att + ((size_t)b * N + n) * DH + c);
    }
  };
  auto scatter = [&](const float4 (&v)[NPC], float mul) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + 512 * i, r = p / (DH / 4), c4 = p % (DH / 4), k = c4 >> 1, h0 = 4 * (c4 & 1);
      float* o = sm + h0 * RP + r * RS + k;
      o[0] = v[i].x * mul; o[RP] = v[i].y * mul; o[2 * RP] = v[i].z * mul; o[3 * RP] = v[i].w * mul;
    }
  };
  // planes -> [b,h,n/16,k/16,16 nodes,16 channels] array `which`: a wave's operand fetch (16 nodes x 16 channels of
  // one k-tile) is ONE contiguous 1 KB block; lanes walk the 16-byte chunks of a block in order
  auto put_rows = [&](int which) __attribute__((always_inline)) {
    float* dst = a.pk + (size_t)which * arr;
    for (int i = tid; i < AH * 16 * (D / 4); i += 512) {
      const int j = i & 3, r = (i >> 2) & 15, T = (i >> 6) % (D / 16), h = i / (4 * D);
      *reinterpret_cast<float4*>(dst + ((size_t)b * AH + h) * NP * D + (size_t)n0 * D + T * 256 + r * 16 + j * 4) =
          *reinterpret_cast<const float4*>(sm + h * RP + r * RS + 16 * T + 4 * j);
    }
  };
  // planes -> transposed, tile-major [b,h,n/16,k,16] array: the 16 nodes of a tile are contiguous per channel and a
  // tile is one 64*D-byte block
  auto put_cols = [&](int which) __attribute__((always_inline)) {
    float* dst = a.pk + (size_t)which * arr;
    for (int i = tid; i < AH * D * 4; i += 512) {
      const int r4 = i & 3, k = (i >> 2) % D, h = i / (4 * D);
      const float* src = sm + h * RP + (r4 * 4) * RS + k;
      *reinterpret_cast<float4*>(dst + ((size_t)b * AH + h) * D * NP + (size_t)n0 * D + k * 16 + r4 * 4) =
          make_float4(src[0], src[RS], src[2 * RS], src[3 * RS]);
    }
  };
  float4 cur[NPC], nxt[NPC];
  load(secs[0], cur);
  for (int si = 0; si < nsec; ++si) {
    const int sec = secs[si];
    if (si + 1 < nsec) load(secs[si + 1], nxt);
    if (si > 0) __syncthreads();          // the previous section's readers are done with the planes
    scatter(cur, sec == 0 ? a.scale : 1.0f);   // Q leaves pre-scaled: S = (d^-1/2 Q).K^T, dK = dA^T.(d^-1/2 Q)
    __syncthreads();
    if (sec == 0) put_rows(PK_QH);
    else if (sec == 1) {
      if (a.pack_what & PACK_KH) put_rows(PK_KH);
      if (a.pack_what & PACK_KT) put_cols(PK_KT);
    } else if (sec == 2) {
      if (a.pack_what & PACK_VT) put_cols(PK_VT);
      if (a.pack_what & PACK_VH) put_rows(PK_VH);
    } else {
      put_rows(PK_OH);
      // per-row constants of the backward, head-major [b,h,n,4] = (m, 1/l, delta, 0) with
      // delta[row,h] = sum_k dO[row,k,h] * O[row,k,h] (flash-style); rows past N get 1/l = 0 (their probabilities vanish)
      const int r = (tid >> 5) & 15, hh = (tid >> 2) & 7, part = tid & 3, n = n0 + r;   // 4 lanes per (row, head)
      float sdel = 0.f;
      if (n < N) {
        const float* vo = a.v_att_in + ((size_t)b * N + n) * DH + hh;
        const float* dr = sm + hh * RP + r * RS;
#pragma unroll 4
        for (int k = part * (D / 4); k < (part + 1) * (D / 4); ++k) sdel = fmaf(dr[k], vo[k * AH], sdel);
      }
      sdel += __shfl_xor(sdel, 1, 64);
      sdel += __shfl_xor(sdel, 2, 64);
      if (part == 0) {
        float4 s2 = make_float4(3.0e38f, 0.f, 0.f, 0.f);   // rows past N: exp2(-inf) * 0
        if (n < N) {
          float* rs = a.rowstats + (((size_t)b * N + n) * AH + hh) * 4;
          rs[3] = sdel;
          s2 = make_float4(rs[0], __builtin_amdgcn_rcpf(rs[1]), sdel, 0.f);
        }
        *reinterpret_cast<float4*>(a.stats2 + (((size_t)b * AH + hh) * NP + n) * 4) = s2;
      }
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) cur[i] = nxt[i];
  }
}

// Feature set of a kernel instance.  V = 0 reads every switch at run time (any combination);
// V = 1 / 2 are the straight-line instances of the main configuration (edge bias + gates + key
// padding + clip, no attention-mask tensor, no injected mask bytes; 2 = in-kernel random mask).
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

// additive mask of one element beyond its key term, in the reference's order (egt_layers.py:96-108)
template <int V>
__device__ __forceinline__ float mask_extra(const AttnMfmaArgs& a, const Feat<V>& f, float mval, uint32_t gi) {
  float add = 0.f;
  if (f.M) add += (mval - 1.0f) * EGT_NEG;
  if (f.rmb || f.rng) {
    const bool hit = f.rmb ? (a.rm[gi] != 0) : ((egt_hash32(gi, a.s0, a.s1) >> 8) < a.rm_thr);
    add += hit ? -EGT_NEG : 0.0f;
  }
  return add;
}

// ---- pair-tile planes: a [16 rows][16 cols x 8 heads] tile of a [B,N,N,8] tensor is 16 contiguous 512-byte
// runs; the workgroup's 512 threads move it with one 16-byte global access each (thread -> row tid>>5, column
// (tid&31)>>1, heads 4*(tid&1)..+3).  In LDS the tile is HEAD-MAJOR: eight planes of stride PT_PL.
//   forward planes  [query row][key]: the wave that owns head h reads the four consecutive keys of its lane with
//     one ds_read_b128; rows 4..7 / 12..15 pair-swapped, 4-column chunks XORed with (row>>1)&3;
//   backward planes [key][query row]: lane (key, q) moves query rows 4q..4q+3 with one b128; chunk XOR (0,2,3,1)[key>>2].
// tools/lds_bank_check.py enumerates every pattern against the gfx950 lane-group tables: all conflict free except the
// backward's 4 x b32 cooperative scatter / gather (4-way: 8 array cycles under a 4-cycle issue, 2 x).
#define PT_PL 260
#define PT_SZ (8 * PT_PL)

__device__ __forceinline__ int pt_off(int row, int m) {
  return ((row ^ ((row >> 2) & 1)) << 4) + ((((m >> 2) ^ (row >> 1)) & 3) << 2) + (m & 3);
}
__device__ __forceinline__ int ptT_off(int row, int col) {
  const int c2 = col >> 2;
  const int a = (0x78 >> (2 * c2)) & 3;   // (0, 2, 3, 1)[c2]
  return (col << 4) + ((((row >> 2) ^ a) & 3) << 2) + (row & 3);
}
// address of a thread's 16 bytes inside a tile: workgroup-uniform base (tensor + graph + first row, scalar
// registers) + a 32-bit lane offset: rowoff is fixed for the kernel, the column part is two VALU ops per tile
__device__ __forceinline__ uint32_t ptile_rowoff(int N, int row0, int tid) {
  return (uint32_t)(min(row0 + (tid >> 5), N - 1) - row0) * N * AH + (tid & 1) * 4;
}
__device__ __forceinline__ uint32_t ptile_off(uint32_t rowoff, int N, int col0, int tid) {
  return rowoff + (uint32_t)min(col0 + ((tid & 31) >> 1), N - 1) * AH;
}
__device__ __forceinline__ const float* ptile_base(const float* src, int b, int N, int row0) {
  return src + ((size_t)b * N + row0) * N * AH;
}
template <bool T>
__device__ __forceinline__ void plane_scatter(float* tl, float4 v, int tid) {
  const int row = tid >> 5, m = (tid & 31) >> 1;
  float* p = tl + (tid & 1) * 4 * PT_PL + (T ? ptT_off(row, m) : pt_off(row, m));
  p[0] = v.x; p[PT_PL] = v.y; p[2 * PT_PL] = v.z; p[3 * PT_PL] = v.w;
}
template <bool T>
__device__ __forceinline__ float4 plane_gather(const float* tl, int tid) {
  const int row = tid >> 5, m = (tid & 31) >> 1;
  const float* p = tl + (tid & 1) * 4 * PT_PL + (T ? ptT_off(row, m) : pt_off(row, m));
  return make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
}

// ---- producer / consumer split (tools/micro/mfma_stream.hip, profiles/r04_attn5_microbench.md) ----------------
// One global_load in an fp32 MFMA stream costs the issuing wave ~80 cycles (= 2.5 MFMAs), a ds_read_b128 ~7; LDS-DMA
// loader waves on the same SIMDs move 18 TB/s without slowing a compute wave's MFMA stream (33.7 vs 32 cycles per
// MFMA).  So a workgroup is 4 COMPUTE waves (one per SIMD, one head each: ds_read + MFMA + VALU only) and 4 LOADER
// waves: operand tiles by LDS-DMA (global_load_lds_dwordx4: no VGPR, no ds_write) one iteration ahead, pair tiles
// HBM -> registers -> head-major planes two iterations ahead, output planes -> HBM one iteration behind.  One
// s_barrier per iteration, which does not drain vector memory.
//
// Operand tile in LDS: the packed [T][16 rows][16 channels] block with the four 16-byte chunks of a row XORed by
// (0,2,3,1)[row >> 2] (done by the DMA's lane -> source mapping: LDS slot s of a 1 KB piece receives row s >> 2,
// chunk (s & 3) ^ a[s >> 4]): the row-form fragment reads (lane (row, q): ds_read_b128 of chunk q) are bank-conflict
// free and the transposed reads (lane (channel, q): ds_read_b32 of rows 4q + r) 2-way (tools/lds_bank_check.py).
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }   // LDS byte address of a __shared__ pointer
__device__ __forceinline__ int chunk_xor(int row) { return (0x78 >> (2 * (row >> 2))) & 3; }      // (0, 2, 3, 1)[row >> 2]
// one 1 KB piece HBM -> LDS (any LDS address: tools/micro/dma_hi.hip): lane i lands at lds + 16 i and fetches src + off
__device__ __forceinline__ void dma_piece(unsigned lds, const float* src, unsigned off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(src), "v"(off), "s"(lds) : "memory");
}
template <int N_>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N_) : "memory"); }
__device__ __forceinline__ unsigned dma_lane_off(int lane) {   // source byte offset (inside a 1 KB block) of the chunk that lands in slot `lane`
  return (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ chunk_xor(lane >> 2)) << 4));
}
#ifndef ATTN_BWD_SWAP
#define ATTN_BWD_SWAP 1   // k_attn_mfma_bwd_kv: operand-tile rows pair-swapped in LDS (A/B: 0)
#endif
#define OPS_STAGE 32768   // bytes of one operand stage: 4 heads x (two 4 KB tiles)
#define OPS_BYTES (2 * OPS_STAGE)

// ================================================================== forward =====
// workgroup = (graph b, 32 query rows, 4 heads); compute wave w = head 4 hg + w.  Lane (ll = lane&15, q = lane>>4)
// owns query rows l0 + 16 qt + ll (two query tiles: every K / V^T operand register feeds two MFMAs) and, in the key
// tile of an iteration, keys m0 + 4q + r.  S^T = K.Q^T puts the key index on the MFMA row axis: the softmax
// reductions over keys are in-lane ops + two permlane swaps, the online-softmax rescale is lane-uniform, and the
// gated probabilities feed the A.V MFMA as its B operand in place.
#define FQ 2   // query tiles per compute wave
template <int D, int V>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_fwd(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH, NIN = V ? 2 : 3, TS = FQ * 4 * PT_PL;   // TS: one tensor, one stage of planes
  static_assert(KT * 1024 * 2 * 4 <= OPS_STAGE, "operand stage");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 4;
  const int w = wv & 3;
  const int N = a.N, NP = a.NP;
  const int lgroups = (NP / 16 + FQ - 1) / FQ;
  const int KA = (NP + 16 + 3) & ~3;
  float* kaddL = sm + OPS_BYTES / 4;   // [NP + 16] additive key term of the logits
  float* kaddG = kaddL + KA;           // ... of the gate's exponent
  float* In = kaddG + KA;              // [2 stages][E | G | M][TS]
  float* Hout = In + 2 * NIN * TS;     // [2][TS]
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);   // the two head groups of a row block and a graph's K / V^T share an XCD's L2
  const int hg = wg & 1, grp = wg >> 1;
  const int b = grp / lgroups, l0 = (grp % lgroups) * 16 * FQ;
  const int h = 4 * hg + w;
  const Feat<V> f(a);
  const int mtiles = NP / 16;
  const size_t arr = (size_t)a.B * AH * NP * D;
  STAMP_DECL;

  if (loader) {
    // ------------------------------------------------------------------ loader waves ----
    const int lt = tid - 256;                   // 0..255
    const int crow = lt >> 4, ccol = lt & 15;   // pair-tile transfers: row crow, key ccol of every 16 x 16 sub-tile, 4 heads (16 bytes)
    for (int m = lt; m < NP + 16; m += 256) {
      float ka = 0.f;
      if (m >= N) ka = KEY_OFF;
      else if (f.km && a.km[(size_t)b * N + m] == 0) ka = -EGT_NEG;
      kaddL[m] = ka;
      kaddG[m] = m >= N ? 3.0e38f : -ka * L2E;
    }
    const float* Kh = a.pk + PK_KH * arr + ((size_t)b * AH + h) * NP * D;   // [NP/16][D/16][16][16]   (this loader wave feeds head h)
    const float* VT = a.pk + PK_VT * arr + ((size_t)b * AH + h) * D * NP;   // [NP/16][D][16]
    const unsigned doff = dma_lane_off(lane);
    const unsigned ops0 = lds_addr(sm) + w * (KT * 2048);   // head w: K tile, then V^T tile
    auto dma_ops = [&](int mt, int stage) __attribute__((always_inline)) {
      const float* ks = Kh + (size_t)mt * 16 * D;
      const float* vs = VT + (size_t)mt * 16 * D;
      const unsigned dst = ops0 + stage * OPS_STAGE;
#pragma unroll
      for (int T = 0; T < KT; ++T) if (ABL(1)) dma_piece(dst + T * 1024, ks + T * 256, doff);
#pragma unroll
      for (int T = 0; T < KT; ++T) if (ABL(1)) dma_piece(dst + KT * 1024 + T * 1024, vs + T * 256, doff);
    };
    const size_t gbase = ((size_t)b * N + l0) * N * AH;
    uint32_t rowoff[FQ];
#pragma unroll
    for (int qt = 0; qt < FQ; ++qt) rowoff[qt] = (uint32_t)(min(l0 + 16 * qt + crow, N - 1) - l0) * N * AH + 4 * hg;
    auto pload = [&](const float* src, int qt, int mcol) __attribute__((always_inline)) {
      return *reinterpret_cast<const float4*>(src + gbase + rowoff[qt] + (uint32_t)min(mcol + ccol, N - 1) * AH);
    };
    const int sc_off = pt_off(crow, ccol);
    auto scatter = [&](float* tl, int qt, float4 v) __attribute__((always_inline)) {
      float* p = tl + qt * 4 * PT_PL + sc_off;
      p[0] = v.x; p[PT_PL] = v.y; p[2 * PT_PL] = v.z; p[3 * PT_PL] = v.w;
    };
    auto hstore = [&](int itp) __attribute__((always_inline)) {   // H_hat tiles of iteration itp: planes -> whole 16-byte pieces of the [N,N,8] rows
      const float* Hp = Hout + (itp & 1) * TS;
      const int mp = 16 * itp;
#pragma unroll
      for (int qt = 0; qt < FQ; ++qt)
        if (l0 + 16 * qt + crow < N && mp + ccol < N) {
          const float* p = Hp + qt * 4 * PT_PL + sc_off;
          *reinterpret_cast<float4*>(a.h_hat + gbase + rowoff[qt] + (uint32_t)(mp + ccol) * AH) = make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
        }
    };
    dma_ops(0, 0);
    float4 pe[FQ], pg[FQ], pm[FQ];
    {
      float4 fe[FQ], fg[FQ], fm[FQ];
      const int m1 = min(16, NP - 16);
#pragma unroll
      for (int qt = 0; qt < FQ; ++qt) {   // both tiles' requests before the first use: one HBM round trip, not two
        if (f.E) fe[qt] = pload(a.E, qt, 0);
        if (f.G) fg[qt] = pload(a.G, qt, 0);
        if (f.M) fm[qt] = pload(a.M, qt, 0);
      }
#pragma unroll
      for (int qt = 0; qt < FQ; ++qt) {
        pe[qt] = pg[qt] = pm[qt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f.E) pe[qt] = pload(a.E, qt, m1);
        if (f.G) pg[qt] = pload(a.G, qt, m1);
        if (f.M) pm[qt] = pload(a.M, qt, m1);
      }
#pragma unroll
      for (int qt = 0; qt < FQ; ++qt) {
        if (f.E) scatter(In + 0 * TS, qt, fe[qt]);
        if (f.G) scatter(In + 1 * TS, qt, fg[qt]);
        if (f.M) scatter(In + 2 * TS, qt, fm[qt]);
      }
    }
    if (V) vm_wait<FQ * NIN>(); else vm_wait<0>();   // the DMA pieces are older than the register loads just issued
    lds_barrier();
    for (int it = 0; it < mtiles; ++it) {
      {   // pair tiles of iteration it+1 (requested one iteration ago) into the other stage: its last readers passed the previous barrier
        float* nx = In + ((it + 1) & 1) * NIN * TS;
#pragma unroll
        for (int qt = 0; qt < FQ; ++qt) if (ABL(2)) {
          if (f.E) scatter(nx + 0 * TS, qt, pe[qt]);
          if (f.G) scatter(nx + 1 * TS, qt, pg[qt]);
          if (f.M) scatter(nx + 2 * TS, qt, pm[qt]);
        }
      }
      if (it > 0 && ABL(4)) hstore(it - 1);
      dma_ops(min(it + 1, mtiles - 1), (it + 1) & 1);
      if (ABL(2)) {
        const int m2 = min(16 * (it + 2), NP - 16);
#pragma unroll
        for (int qt = 0; qt < FQ; ++qt) {
          if (f.E) pe[qt] = pload(a.E, qt, m2);
          if (f.G) pg[qt] = pload(a.G, qt, m2);
          if (f.M) pm[qt] = pload(a.M, qt, m2);
        }
      }
      // exactly the register loads above are younger than the DMA pieces (their count is a template constant for the
      // straight-line instances; the generic instance drains)
      if (V && EGT_ATTN_ABL == 0) vm_wait<FQ * NIN>(); else vm_wait<0>();
      lds_barrier();
    }
    hstore(mtiles - 1);
  } else {
  // -------------------------------------------------------------------- compute waves ----
  const int ll = lane & 15, q = lane >> 4;
  const bool gated = f.G, clip = f.clip;
  // Q fragments (B operand of S^T = K.Q^T), pre-scaled: d^-1/2 Q[l][16T + 4q + u]
  float Qr[FQ][4 * KT];
  {
    const float* Qh = a.pk + PK_QH * arr + ((size_t)b * AH + h) * NP * D;
#pragma unroll
    for (int qt = 0; qt < FQ; ++qt) {
      const int ltile = min((l0 >> 4) + qt, mtiles - 1);
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        const float4 v = *reinterpret_cast<const float4*>(Qh + (size_t)ltile * 16 * D + 256 * T + ll * 16 + 4 * q);
        Qr[qt][4 * T + 0] = v.x; Qr[qt][4 * T + 1] = v.y; Qr[qt][4 * T + 2] = v.z; Qr[qt][4 * T + 3] = v.w;
      }
    }
  }
  v4f oacc[FQ][KT];
  float mrun[FQ], lrun[FQ];
#pragma unroll
  for (int qt = 0; qt < FQ; ++qt) {
    mrun[qt] = KEY_OFF; lrun[qt] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) oacc[qt][kt] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
  const int po4 = w * PT_PL + pt_off(ll, 4 * q);   // this lane's keys 4q..4q+3 of row ll inside a sub-tile: one 16-byte access
  // operand fragment address: row ll, chunk q of the k-tile blocks of head w's tiles (+ stage, + 1024 T, + KT*1024 for V^T)
  const float* opl = sm + w * (KT * 512) + ll * 16 + ((q ^ chunk_xor(ll)) << 2);
  lds_barrier();
  STAMP(0);
  for (int it = 0; it < mtiles; ++it) {
    const int m0 = 16 * it;
    const float* ops = opl + (it & 1) * (OPS_STAGE / 4);
    const float* Inb = In + (it & 1) * NIN * TS;
    float* Hb2 = Hout + (it & 1) * TS;
    float4 kc[KT], vc[KT];
#pragma unroll
    for (int T = 0; T < KT; ++T) kc[T] = *reinterpret_cast<const float4*>(ops + T * 256);
    float4 e4[FQ], g4[FQ], m4[FQ], ka4, kg4 = make_float4(0.f, 0.f, 0.f, 0.f);
    ka4 = *reinterpret_cast<const float4*>(kaddL + m0 + 4 * q);
    if (V == 1) kg4 = *reinterpret_cast<const float4*>(kaddG + m0 + 4 * q);
#pragma unroll
    for (int qt = 0; qt < FQ; ++qt) {
      e4[qt] = g4[qt] = make_float4(0.f, 0.f, 0.f, 0.f);
      m4[qt] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (f.E) e4[qt] = *reinterpret_cast<const float4*>(Inb + 0 * TS + qt * 4 * PT_PL + po4);
      if (f.G) g4[qt] = *reinterpret_cast<const float4*>(Inb + 1 * TS + qt * 4 * PT_PL + po4);
      if (f.M) m4[qt] = *reinterpret_cast<const float4*>(Inb + 2 * TS + qt * 4 * PT_PL + po4);
    }
    __builtin_amdgcn_sched_barrier(0);   // every LDS request of the S phase is in flight before its first MFMA
    STAMP(1);
    // ---- S^T[m][l] = sum_k K[m][k] (d^-1/2 Q)[l][k]: every K register feeds the two query tiles ----
    v4f s[FQ];
#pragma unroll
    for (int qt = 0; qt < FQ; ++qt) s[qt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < KT; ++T)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float kk = u == 0 ? kc[T].x : u == 1 ? kc[T].y : u == 2 ? kc[T].z : kc[T].w;
#pragma unroll
        for (int qt = 0; qt < FQ; ++qt) s[qt] = MFMA(kk, Qr[qt][4 * T + u], s[qt]);
      }
#pragma unroll
    for (int T = 0; T < KT; ++T) vc[T] = *reinterpret_cast<const float4*>(ops + KT * 256 + T * 256);   // V^T fragments: requested behind the S MFMAs, landed long before P.V
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    const float kav[4] = {ka4.x, ka4.y, ka4.z, ka4.w}, kgv[4] = {kg4.x, kg4.y, kg4.z, kg4.w};
    float pa[FQ][4];
#pragma unroll
    for (int qt = 0; qt < FQ; ++qt) {
      float x[4];
      float tmax = KEY_OFF;
      const int lq = min(l0 + 16 * qt + ll, N - 1);
      const float ev[4] = {e4[qt].x, e4[qt].y, e4[qt].z, e4[qt].w}, gv[4] = {g4[qt].x, g4[qt].y, g4[qt].z, g4[qt].w};
      const float mv[4] = {m4[qt].x, m4[qt].y, m4[qt].z, m4[qt].w};
      float hv4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ah = s[qt][r];
        if (clip) ah = __builtin_amdgcn_fmed3f(ah, a.clip_lo, a.clip_hi);
        const float hv = ah + ev[r];
        hv4[r] = hv;                                            // H_hat: post-clip, pre-mask (egt_layers.py:85-86)
        float add = kav[r];
        if (V != 1) {
          const int m = min(m0 + 4 * q + r, N - 1);
          add += mask_extra(a, f, mv[r], (uint32_t)((((size_t)b * N + lq) * N + m) * AH + h));
        }
        x[r] = hv + add;
        tmax = fmaxf(tmax, x[r]);
        if (gated) {   // sigmoid(G + add); masked -> exp2(+1.4e9) = inf -> rcp = 0 exactly, as the reference's fp32 path
          const float tg = V == 1 ? exp2_fast(fmaf(gv[r], -L2E, kgv[r])) : exp2_fast((gv[r] + add) * -L2E);
          pa[qt][r] = __builtin_amdgcn_rcpf(1.0f + tg);
        } else {
          pa[qt][r] = 1.0f;
        }
      }
      *reinterpret_cast<float4*>(Hb2 + qt * 4 * PT_PL + po4) = make_float4(hv4[0], hv4[1], hv4[2], hv4[3]);
      // ---- online softmax over the 16 keys: in-lane over r, then across q ----
      tmax = pair_max_q(tmax);
      const float mnew = fmaxf(mrun[qt], tmax);
      const float alpha = exp2_fast((mrun[qt] - mnew) * L2E);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pexp = exp2_fast((x[r] - mnew) * L2E);   // keys past N: exp2(-inf) = 0
        psum += pexp;
        pa[qt][r] *= pexp;
      }
      psum = pair_sum_q(psum);
      lrun[qt] = fmaf(lrun[qt], alpha, psum);
      mrun[qt] = mnew;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        v4f o = oacc[qt][kt];
        o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
        oacc[qt][kt] = o;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    // ---- O^T[k][l] += sum_m V^T[k][m] P^T[m][l]  (contraction order m = 4q + t): every V^T register feeds two MFMAs ----
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float va = r == 0 ? vc[kt].x : r == 1 ? vc[kt].y : r == 2 ? vc[kt].z : vc[kt].w;
#pragma unroll
        for (int qt = 0; qt < FQ; ++qt) oacc[qt][kt] = MFMA(va, pa[qt][r], oacc[qt][kt]);
      }
    STAMP(4);
    lds_barrier();
    STAMP(6);
  }
  // ---- finalize: O[l][k] / l_run into the (now idle) operand stages as [row][k][4 heads]; row statistics for the backward ----
#pragma unroll
  for (int qt = 0; qt < FQ; ++qt) {
    const int l = l0 + 16 * qt + ll;
    const float inv = 1.0f / lrun[qt];
    float* vs = sm + ((16 * qt + ll) * D + 4 * q) * 4 + w;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vs[(16 * kt + r) * 4] = oacc[qt][kt][r] * inv;
    if (l < N && q == 0)
      *reinterpret_cast<float4*>(a.rowstats + (((size_t)b * N + l) * AH + h) * 4) = make_float4(mrun[qt], lrun[qt], 0.f, 0.f);
  }
  STAMP(7);
  STAMP_OUT(0);
  }
  // V_att[l][k*8 + h]: all eight waves store 16-byte pieces (the group's 4 heads of one channel), consecutive threads
  // consecutive channels (a dword store per lane and channel costs ~300 cycles each: 19 k cycles per workgroup before)
  lds_barrier();
  for (int p = tid; p < 16 * FQ * D; p += 512) {
    const int row = p / D, k = p % D;
    if (l0 + row < N)
      *reinterpret_cast<float4*>(a.v_att + ((size_t)b * N + l0 + row) * DH + k * AH + 4 * hg) = *reinterpret_cast<const float4*>(sm + (size_t)p * 4);
  }
}

// ================================================================= backward =====
// Launches (flash-attention style, the [N,N,H] probabilities are recomputed):
//   k_attn_pack        : head-major operand arrays of Q (pre-scaled), K, V, dV_att; the per-row constants
//                        (m, 1/l, delta = sum_k dO*O) head-major
//   k_attn_mfma_bwd_kv : workgroup = (graph, 32 keys, 4 heads): compute wave = head, K / V fragments and the dK / dV
//                        accumulators of TWO key tiles in registers while the workgroup walks the query tiles; the loader
//                        waves stage the Q / dO tiles of a query tile by LDS-DMA (the compute waves read BOTH operand forms
//                        from that one tile), the pair tiles (E, G, dH_ext) into key-major planes, and store dE / dG;
//                        per tile S = Q.K^T and dP = dO.V^T on MFMA, softmax / gate / clip backward on the VALU, then
//                        dV^T += dO^T.A and dK^T += Q^T.dA on MFMA with the probabilities as B operands in place;
//                        dA = dH*c leaves in the lane's own layout (one 16-byte store per key tile)
//   k_attn_mfma_bwd_q  : wave = (graph, head, 32 query rows), walks the key tiles: dQ^T += K^T.dA^T straight from the
//                        dA tiles and the packed K^T
// Compute lane (mm = lane&15, q): keys m0 + 16 kb + mm; in every query tile rows l0 + 4q + r.
#define BK 2   // key tiles per compute wave
template <int D, int V>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_bwd_kv(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH, NIN = 3, TS = BK * 4 * PT_PL;   // planes E | G | X (an attention-mask tensor is folded into E and G by the loaders)
  static_assert(KT * 1024 * 2 * 4 <= OPS_STAGE, "operand stage");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 4;
  const int w = wv & 3;
  const int N = a.N, NP = a.NP;
  const int mtiles = NP / 16, mgroups = (mtiles + BK - 1) / BK;
  float* statL = sm + OPS_BYTES / 4;   // [2 stages][4 heads][16 rows][4]
  float* In = statL + 2 * 4 * 64;      // [2][E | G | X][TS], key-major planes
  float* Out = In + 2 * NIN * TS;      // [2][dE | dG][TS]
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int hg = wg & 1, grp = wg >> 1;
  const int b = grp / mgroups, m0 = (grp % mgroups) * 16 * BK;
  const int h = 4 * hg + w;
  const Feat<V> f(a);
  const size_t arr = (size_t)a.B * AH * NP * D;
  const size_t hb = ((size_t)b * AH + h) * NP * D;
  const size_t gb = (size_t)b * N * N * AH;
  STAMP_DECL;

  if (loader) {
    // ------------------------------------------------------------------ loader waves ----
    const int lt = tid - 256;
    const int crow = lt >> 4, ccol = lt & 15;   // query row, key of each 16 x 16 sub-tile; the group's 4 heads (16 bytes)
    const float* Qh = a.pk + PK_QH * arr + hb;  // this loader wave feeds head h
    const float* Oh = a.pk + PK_OH * arr + hb;
    const unsigned doff = dma_lane_off(ATTN_BWD_SWAP ? lane ^ (((lane >> 4) & 1) << 2) : lane);   // (swap: LDS row sigma <- tile row sigma ^ ((sigma >> 2) & 1), see the compute waves)
    const unsigned ops0 = lds_addr(sm) + w * (KT * 2048);   // head w: Q tile, then dO tile
    auto dma_ops = [&](int ltile, int stage) __attribute__((always_inline)) {
      const float* qs = Qh + (size_t)ltile * 16 * D;
      const float* os = Oh + (size_t)ltile * 16 * D;
      const unsigned dst = ops0 + stage * OPS_STAGE;
#pragma unroll
      for (int T = 0; T < KT; ++T) if (ABL(1)) dma_piece(dst + T * 1024, qs + T * 256, doff);
#pragma unroll
      for (int T = 0; T < KT; ++T) if (ABL(1)) dma_piece(dst + KT * 1024 + T * 1024, os + T * 256, doff);
    };
    // per-row constants of the query tile: one float4 per (head, row) -- lanes 0..15 of each loader wave
    const float* st2 = a.stats2 + ((size_t)b * AH + h) * NP * 4;
    uint32_t coff[BK];
#pragma unroll
    for (int kb = 0; kb < BK; ++kb) coff[kb] = (uint32_t)min(m0 + 16 * kb + ccol, N - 1) * AH + 4 * hg;   // + row * N * 8
    auto cload = [&](const float* src, int l0, int kb) __attribute__((always_inline)) {
      return *reinterpret_cast<const float4*>(src + gb + (size_t)min(l0 + crow, N - 1) * N * AH + coff[kb]);
    };
    struct Pair { float4 e, g, x; };
    // tensors past their valid range read a valid address; only dH_ext must be ZERO there (it is added to dH).  An
    // attention-mask tensor enters the backward only through the additive term of logit and gate: folded into E and G here.
    auto pair_load = [&](int l0, int kb) __attribute__((always_inline)) {
      Pair p;
      p.e = p.g = p.x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f.E) p.e = cload(a.E, l0, kb);
      if (f.G) p.g = cload(a.G, l0, kb);
      if (f.X) {
        p.x = cload(a.d_h_ext, l0, kb);
        if (!(l0 + crow < N && m0 + 16 * kb + ccol < N)) p.x = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (f.M) {
        const float4 mk = cload(a.M, l0, kb);
        p.e.x += (mk.x - 1.0f) * EGT_NEG; p.e.y += (mk.y - 1.0f) * EGT_NEG; p.e.z += (mk.z - 1.0f) * EGT_NEG; p.e.w += (mk.w - 1.0f) * EGT_NEG;
        p.g.x += (mk.x - 1.0f) * EGT_NEG; p.g.y += (mk.y - 1.0f) * EGT_NEG; p.g.z += (mk.z - 1.0f) * EGT_NEG; p.g.w += (mk.w - 1.0f) * EGT_NEG;
      }
      return p;
    };
    const int sc_off = ptT_off(crow, ccol);
    auto scatter = [&](float* tl, int kb, float4 v) __attribute__((always_inline)) {
      float* p = tl + kb * 4 * PT_PL + sc_off;
      p[0] = v.x; p[PT_PL] = v.y; p[2 * PT_PL] = v.z; p[3 * PT_PL] = v.w;
    };
    auto pair_put = [&](float* st, int kb, const Pair& p) __attribute__((always_inline)) {
      scatter(st + 0 * TS, kb, p.e);
      if (f.G) scatter(st + 1 * TS, kb, p.g);
      scatter(st + 2 * TS, kb, p.x);
    };
    auto gstore = [&](int itp) __attribute__((always_inline)) {   // dE / dG tiles of iteration itp: planes -> 16-byte pieces of the [N,N,8] rows
      const int lp = 16 * itp;
      const float* o = Out + (itp & 1) * 2 * TS;
#pragma unroll
      for (int kb = 0; kb < BK; ++kb)
        if (lp + crow < N && m0 + 16 * kb + ccol < N) {
          const float* p = o + kb * 4 * PT_PL + sc_off;
          const size_t go = gb + (size_t)(lp + crow) * N * AH + coff[kb];
          if (f.E) *reinterpret_cast<float4*>(a.d_E + go) = make_float4(p[0], p[PT_PL], p[2 * PT_PL], p[3 * PT_PL]);
          if (f.G) *reinterpret_cast<float4*>(a.d_G + go) = make_float4(p[TS], p[TS + PT_PL], p[TS + 2 * PT_PL], p[TS + 3 * PT_PL]);
        }
    };
    auto stat_load = [&](int ltile) __attribute__((always_inline)) {   // (every lane loads: lanes >= 16 re-read rows 0..15 -- the count of register loads per trip stays fixed)
      return *reinterpret_cast<const float4*>(st2 + (size_t)(ltile * 16 + (lane & 15)) * 4);
    };
    auto stat_put = [&](float4 v, int stage) __attribute__((always_inline)) {
      if (lane < 16) *reinterpret_cast<float4*>(statL + ((stage * 4 + w) * 16 + lane) * 4) = v;
    };
    dma_ops(0, 0);
    Pair pr[BK];
    {
      Pair p0[BK];
#pragma unroll
      for (int kb = 0; kb < BK; ++kb) p0[kb] = pair_load(0, kb);   // both tiles' requests before the first use: one HBM round trip
#pragma unroll
      for (int kb = 0; kb < BK; ++kb) pr[kb] = pair_load(min(16, NP - 16), kb);
      stat_put(stat_load(0), 0);
#pragma unroll
      for (int kb = 0; kb < BK; ++kb) pair_put(In, kb, p0[kb]);
    }
    float4 stn = stat_load(min(1, mtiles - 1));
    vm_wait<0>();
    lds_barrier();
    for (int it = 0; it < mtiles; ++it) {
      {   // pair tiles of query tile it+1 (requested one iteration ago) into the other stage
        float* nx = In + ((it + 1) & 1) * NIN * TS;
#pragma unroll
        for (int kb = 0; kb < BK; ++kb) if (ABL(2)) pair_put(nx, kb, pr[kb]);
      }
      stat_put(stn, (it + 1) & 1);
      if (it > 0 && ABL(4)) gstore(it - 1);
      dma_ops(min(it + 1, mtiles - 1), (it + 1) & 1);
      if (ABL(2)) {
        const int l2 = min(16 * (it + 2), NP - 16);
#pragma unroll
        for (int kb = 0; kb < BK; ++kb) pr[kb] = pair_load(l2, kb);
        stn = stat_load(l2 >> 4);
      }
      if (V && EGT_ATTN_ABL == 0) vm_wait<BK * 3 + 1>(); else vm_wait<0>();   // exactly the BK * 3 + 1 register loads above are younger than the DMA pieces
      lds_barrier();
    }
    gstore(mtiles - 1);
  } else {
  // -------------------------------------------------------------------- compute waves ----
  const int mm = lane & 15, q = lane >> 4;
  const bool gated = f.G, clip = f.clip;
  const uint32_t ooff = mm * 16 + 4 * q;
  // K / V fragments of this lane's keys (B operands of S = Q.K^T and dP = dO.V^T); key rows past N are zero in the pack,
  // a key tile past NP re-reads the last one and is switched off by its additive term
  float4 Kr[BK][KT], Vr[BK][KT];
  float kadd[BK], kaddg[BK];
  bool tile_ok[BK];
  float* dAb[BK];
  {
    const float* Kh = a.pk + PK_KH * arr + hb;
    const float* Vh = a.pk + PK_VH * arr + hb;
#pragma unroll
    for (int kb = 0; kb < BK; ++kb) {
      const int mt = (m0 >> 4) + kb, mtc = min(mt, mtiles - 1);
      tile_ok[kb] = mt < mtiles;
      dAb[kb] = a.ws_dA + (((size_t)b * AH + h) * mtiles * mtiles + (size_t)mtc) * 256;   // + ltile * mtiles * 256
#pragma unroll
      for (int T = 0; T < KT; ++T) {
        Kr[kb][T] = *reinterpret_cast<const float4*>(Kh + (size_t)mtc * 16 * D + 256 * T + ooff);
        Vr[kb][T] = *reinterpret_cast<const float4*>(Vh + (size_t)mtc * 16 * D + 256 * T + ooff);
      }
      const int m = m0 + 16 * kb + mm;
      kadd[kb] = 0.f;
      if (m >= N) kadd[kb] = KEY_OFF;
      else if (f.km && a.km[(size_t)b * N + m] == 0) kadd[kb] = -EGT_NEG;
      kaddg[kb] = m < N ? -kadd[kb] * L2E : 3.0e38f;
    }
  }
  v4f dKacc[BK][KT], dVacc[BK][KT];
#pragma unroll
  for (int kb = 0; kb < BK; ++kb)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { dKacc[kb][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; dVacc[kb][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  const int po4 = w * PT_PL + ptT_off(4 * q, mm);   // this lane's query rows 4q..4q+3 of key mm inside a sub-tile: one 16-byte access
  // operand tiles of head w (Q, then dO): row form: lane (row mm, chunk q) b128; transposed form: lane (channel mm, q):
  // rows 4q + r, dword mm & 3 of chunk mm >> 2
  // (ATTN_BWD_SWAP: tile rows 4-7 / 12-15 pair-swapped in LDS, as in k_pair_bwd: the transposed b32 reads of a 32-lane group touch
  //  tile rows r and 4 + r -- the same 16 banks with 16-float rows, 2-way on the 32 reads of a tile; swapped, the group covers 32 banks)
  const float* oprow = sm + w * (KT * 512) + (ATTN_BWD_SWAP ? mm ^ ((mm >> 2) & 1) : mm) * 16 + ((q ^ chunk_xor(mm)) << 2);
  const float* optr = sm + w * (KT * 512) + (4 * q) * 16 + (((mm >> 2) ^ chunk_xor(4 * q)) << 2) + (mm & 3);
  const float* optr_e = optr + (ATTN_BWD_SWAP ? (q & 1) * 16 : 0);   // steps 0, 2: LDS row 4q + r + (q & 1)
  const float* optr_o = optr - (ATTN_BWD_SWAP ? (q & 1) * 16 : 0);   // steps 1, 3: LDS row 4q + r - (q & 1)
  lds_barrier();
  STAMP(0);
  for (int l0 = 0, it = 0; it < mtiles; l0 += 16, ++it) {
    const float* opr = oprow + (it & 1) * (OPS_STAGE / 4);
    const float* opt_e = optr_e + (it & 1) * (OPS_STAGE / 4);
    const float* opt_o = optr_o + (it & 1) * (OPS_STAGE / 4);
    const float* Inb = In + (it & 1) * NIN * TS;
    float* Ob = Out + (it & 1) * 2 * TS;
    float4 qa[KT], oa[KT];   // row operands first: the S / dP MFMAs start as soon as they land, everything else lands behind them
#pragma unroll
    for (int T = 0; T < KT; ++T) {
      qa[T] = *reinterpret_cast<const float4*>(opr + T * 256);
      oa[T] = *reinterpret_cast<const float4*>(opr + KT * 256 + T * 256);
    }
    float4 e4[BK], g4[BK], x4[BK], stc[4];
#pragma unroll
    for (int kb = 0; kb < BK; ++kb) {
      e4[kb] = *reinterpret_cast<const float4*>(Inb + 0 * TS + kb * 4 * PT_PL + po4);
      g4[kb] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f.G) g4[kb] = *reinterpret_cast<const float4*>(Inb + 1 * TS + kb * 4 * PT_PL + po4);
      x4[kb] = *reinterpret_cast<const float4*>(Inb + 2 * TS + kb * 4 * PT_PL + po4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) stc[r] = *reinterpret_cast<const float4*>(statL + (((it & 1) * 4 + w) * 16 + 4 * q + r) * 4);
    __builtin_amdgcn_sched_barrier(0);   // every LDS request of the first two phases is in flight before the first MFMA
    STAMP(1);
    // ---- S[l][m] = sum_k (d^-1/2 Q)[l][k] K[m][k] ; dP[l][m] = sum_k dO[l][k] V[m][k]: every Q / dO register feeds two MFMAs ----
    v4f s[BK], dp[BK];
#pragma unroll
    for (int kb = 0; kb < BK; ++kb) { s[kb] = (v4f){0.f, 0.f, 0.f, 0.f}; dp[kb] = (v4f){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int T = 0; T < KT; ++T) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float qq = u == 0 ? qa[T].x : u == 1 ? qa[T].y : u == 2 ? qa[T].z : qa[T].w;
        const float oo = u == 0 ? oa[T].x : u == 1 ? oa[T].y : u == 2 ? oa[T].z : oa[T].w;
#pragma unroll
        for (int kb = 0; kb < BK; ++kb) {
          const float kk = u == 0 ? Kr[kb][T].x : u == 1 ? Kr[kb][T].y : u == 2 ? Kr[kb][T].z : Kr[kb][T].w;
          const float vv = u == 0 ? Vr[kb][T].x : u == 1 ? Vr[kb][T].y : u == 2 ? Vr[kb][T].z : Vr[kb][T].w;
          s[kb] = MFMA(qq, kk, s[kb]);
          dp[kb] = MFMA(oo, vv, dp[kb]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    float at[BK][4], da[BK][4];
#pragma unroll
    for (int kb = 0; kb < BK; ++kb) {
      const float ev[4] = {e4[kb].x, e4[kb].y, e4[kb].z, e4[kb].w}, gv[4] = {g4[kb].x, g4[kb].y, g4[kb].z, g4[kb].w};
      const float xv4[4] = {x4[kb].x, x4[kb].y, x4[kb].z, x4[kb].w};
      float dE4[4], dG4[4];
      const int mc = min(m0 + 16 * kb + mm, N - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float araw = s[kb][r];
        float ah = araw;
        if (clip) ah = __builtin_amdgcn_fmed3f(araw, a.clip_lo, a.clip_hi);
        float add = kadd[kb];
        if (V != 1) {
          const int lq = min(l0 + 4 * q + r, N - 1);
          add += mask_extra(a, f, 1.0f, (uint32_t)((((size_t)b * N + lq) * N + mc) * AH + h));   // (the mask tensor arrives inside E / G)
        }
        const float xx = ah + ev[r] + add;
        const float S = exp2_fast((xx - stc[r].x) * L2E) * stc[r].y;   // rows past N: 1/l = 0; keys past N: exp2(-inf) = 0
        float g = 1.0f;
        if (gated) {
          const float tg = V == 1 ? exp2_fast(fmaf(gv[r], -L2E, kaddg[kb])) : exp2_fast((gv[r] + add) * -L2E);
          g = __builtin_amdgcn_rcpf(1.0f + tg);
        }
        const float dAt_ = dp[kb][r];
        const float Sg = S * g;
        const float dH = fmaf(S, fmaf(dAt_, g, -stc[r].z), xv4[r]);
        at[kb][r] = Sg;
        da[kb][r] = (clip && ah != araw) ? 0.f : dH;   // clip passes the gradient where lo <= x <= hi
        dE4[r] = dH;
        dG4[r] = gated ? dAt_ * Sg * (1.0f - g) : 0.f;
      }
      if (f.E) *reinterpret_cast<float4*>(Ob + kb * 4 * PT_PL + po4) = make_float4(dE4[0], dE4[1], dE4[2], dE4[3]);
      if (f.G) *reinterpret_cast<float4*>(Ob + TS + kb * 4 * PT_PL + po4) = make_float4(dG4[0], dG4[1], dG4[2], dG4[3]);
      // dA tile [key][query] of (head, query tile, key tile): the lane's own four rows, one 16-byte store
      if (tile_ok[kb] && ABL(4))
        *reinterpret_cast<float4*>(dAb[kb] + (size_t)it * mtiles * 256 + ooff) = make_float4(da[kb][0], da[kb][1], da[kb][2], da[kb][3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    // ---- dV^T[k][m] += sum_l dO[l][k] A[l][m] ; dK^T[k][m] += sum_l (d^-1/2 Q)[l][k] dA[l][m]: transposed operands from the same tiles ----
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* opt = (r & 1) ? opt_o : opt_e;
        const float qq = opt[kt * 256 + r * 16];
        const float oo = opt[KT * 256 + kt * 256 + r * 16];
#pragma unroll
        for (int kb = 0; kb < BK; ++kb) {
          dVacc[kb][kt] = MFMA(oo, at[kb][r], dVacc[kb][kt]);
          dKacc[kb][kt] = MFMA(qq, da[kb][r], dKacc[kb][kt]);
        }
      }
    STAMP(4);
    lds_barrier();
    STAMP(6);
  }
  // dK / dV into the (now idle) operand stages as [key][k][4 heads], dK in stage 0, dV in stage 1
#pragma unroll
  for (int kb = 0; kb < BK; ++kb) {
    float* ks = sm + ((16 * kb + mm) * D + 4 * q) * 4 + w;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ks[(16 * kt + r) * 4] = dKacc[kb][kt][r];
        ks[OPS_STAGE / 4 + (16 * kt + r) * 4] = dVacc[kb][kt][r];
      }
  }
  STAMP(7);
  STAMP_OUT(1);
  }
  // dK / dV rows of d_qkv: all eight waves store 16-byte pieces (the group's 4 heads of one channel)
  lds_barrier();
  for (int p = tid; p < 16 * BK * D; p += 512) {
    const int row = p / D, k = p % D;
    if (m0 + row < N) {
      float* o = a.d_qkv + ((size_t)b * N + m0 + row) * 3 * DH + k * AH + 4 * hg;
      *reinterpret_cast<float4*>(o + DH) = *reinterpret_cast<const float4*>(sm + (size_t)p * 4);
      *reinterpret_cast<float4*>(o + 2 * DH) = *reinterpret_cast<const float4*>(sm + OPS_STAGE / 4 + (size_t)p * 4);
    }
  }
}

// dQ: workgroup = (graph, head, 128 query rows): 4 compute waves (32 query rows = two tiles each) + 4 loader waves.  Per
// iteration the loaders stage two key tiles by LDS-DMA: their K^T tiles (shared by the four compute waves) and the sixteen
// dA tiles (query tile x key tile) -- the register-direct version issued 24 global loads per 64 MFMAs, i.e. as many issue
// cycles of loads as of MFMAs.  Compute lane (ll = lane&15, q): query rows 16 j + ll of its tiles; keys 4q + u as the
// contraction index: K^T fragments by ds_read_b128 (chunk-swizzled like every operand tile), dA[key][query] read
// transposed by ds_read_b32.
#define QW_TILES 8   // query tiles per workgroup
#define QW_STAGES 4  // LDS ring: the DMA of a stage is issued QW_STAGES - 1 trips ahead of its use (issue -> landed is ~1.1 us, a trip ~1 us)
template <int D>
__global__ void __launch_bounds__(512, 2) k_attn_mfma_bwd_q(AttnMfmaArgs a) {
  constexpr int KT = D / 16, DH = D * AH, STG = (2 * KT + 2 * QW_TILES) * 256;   // floats per stage
  constexpr int PPW = (2 * KT + 2 * QW_TILES + 3) / 4;                           // DMA pieces per loader wave and stage (upper bound)

  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, NP = a.NP;
  const int mtiles = NP / 16, qgroups = (mtiles + QW_TILES - 1) / QW_TILES;
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / (AH * qgroups), h = (wg / qgroups) % AH, lt0 = (wg % qgroups) * QW_TILES;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const float* KTp = a.pk + PK_KT * arr + ((size_t)b * AH + h) * D * NP;           // [mtile][D][16 keys]
  const float* dAh = a.ws_dA + ((size_t)b * AH + h) * mtiles * mtiles * 256;       // [ltile][mtile][16 keys][16 queries]
  const int nit = (mtiles + 1) / 2;
  if (wv >= 4) {
    // ---- loaders: piece p of a stage: p < 2 KT: K^T block (kb = p / KT, T = p % KT); else dA tile (ltile = (p - 2 KT) >> 1, kb = (p - 2 KT) & 1)
    const int j = wv - 4;
    const unsigned doff = dma_lane_off(lane), lin = lane * 16;
    // piece i of this wave: p = j + 4 i; its source address advances by a constant per trip (two key tiles): K^T blocks by
    // 32 D floats, dA tiles by 512 -- bases and strides are hoisted (the address arithmetic of a piece was ~45 scalar
    // instructions, six pieces per trip: the loaders were SALU-bound)
    const float* base[PPW];
    int stride[PPW];
    unsigned voff[PPW];
    bool second[PPW];   // piece belongs to the second key tile of a trip (absent in the last trip of an odd tile count)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = j + 4 * i;
      if (p < 2 * KT) {
        base[i] = KTp + (size_t)(p / KT) * 16 * D + (p % KT) * 256; stride[i] = 32 * D; voff[i] = doff; second[i] = p / KT != 0;
      } else {
        const int t = p - 2 * KT, ltile = min(lt0 + (t >> 1), mtiles - 1);
        base[i] = dAh + ((size_t)ltile * mtiles + (t & 1)) * 256; stride[i] = 512; voff[i] = lin; second[i] = (t & 1) != 0;
      }
    }
    auto stage_in = [&](int it, int stage) __attribute__((always_inline)) {
      const unsigned dst = lds_addr(sm) + stage * (STG * 4);
      const bool odd_last = 2 * it + 1 >= mtiles;   // the second key tile of this trip does not exist: re-read the first (its MFMAs are skipped)
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int p = j + 4 * i;
        if (p < 2 * KT + 2 * QW_TILES) {
          const float* src = base[i] + (size_t)it * stride[i];
          if (odd_last && second[i]) src -= (p < 2 * KT) ? 16 * D : 256;
          if (p < 2 * KT ? ABL(1) : ABL(32)) dma_piece(dst + p * 1024, src, voff[i]);
        }
      }
    };
    const bool full = j + 4 * (PPW - 1) < 2 * KT + 2 * QW_TILES;   // this wave issues PPW (else PPW - 1) pieces per stage
    auto wait_ring = [&]() __attribute__((always_inline)) {
      if (full) vm_wait<(QW_STAGES - 2) * PPW>(); else vm_wait<(QW_STAGES - 2) * (PPW - 1)>();
    };
#pragma unroll
    for (int t = 0; t < QW_STAGES - 1; ++t) if (t < nit) stage_in(t, t);
    if (nit >= QW_STAGES - 1) wait_ring(); else vm_wait<0>();
    lds_barrier();
    for (int it = 0; it < nit; ++it) {
      const int tn = it + QW_STAGES - 1;
      if (tn < nit) { stage_in(tn, tn % QW_STAGES); wait_ring(); }
      else vm_wait<0>();   // the ring drains: nothing is in flight when the stages are reused by the epilogue
      lds_barrier();
    }
  } else {
    const int ll = lane & 15, q = lane >> 4;
    v4f dQ[2][KT];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) dQ[jj][kt] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float* kfrag = sm + ll * 16 + ((q ^ chunk_xor(ll)) << 2);                 // + stage, + (kb KT + kt) * 256
    const float* dfrag = sm + 2 * KT * 256 + (2 * wv) * 512 + (4 * q) * 16 + ll;   // + stage, + jj * 512 + kb * 256 + u * 16
    struct Frag { float4 kc[2][KT]; float da[2][2][4]; };
    auto fetch = [&](int stage, Frag& fr) __attribute__((always_inline)) {
      const float* kf = kfrag + stage * STG;
      const float* df = dfrag + stage * STG;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) fr.kc[kb][kt] = *reinterpret_cast<const float4*>(kf + (kb * KT + kt) * 256);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int u = 0; u < 4; ++u) fr.da[kb][jj][u] = df[jj * 512 + kb * 256 + u * 16];
    };
    // dQ^T[k][l] += sum_m K^T[k][m] dA[l][m]  (a key tile past the end -- odd tile counts -- is skipped)
    auto mma = [&](int it, const Frag& fr) __attribute__((always_inline)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        if (2 * it + kb < mtiles) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const float kk = u == 0 ? fr.kc[kb][kt].x : u == 1 ? fr.kc[kb][kt].y : u == 2 ? fr.kc[kb][kt].z : fr.kc[kb][kt].w;
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) dQ[jj][kt] = MFMA(kk, fr.da[kb][jj][u], dQ[jj][kt]);
            }
        }
    };
    // The MFMAs lag one trip behind the LDS requests: behind barrier it the fragments of stage it+1 are requested, then the
    // MFMAs of trip it run from the registers filled a trip ago -- the LDS latency never faces an idle matrix pipe.  Two
    // named register sets swap roles.
    Frag fa, fb;
    lds_barrier();
    fetch(0, fa);
    int it = 0;
    for (; it + 1 < nit; it += 2) {
      lds_barrier(); fetch((it + 1) % QW_STAGES, fb); mma(it, fa);
      lds_barrier(); fetch((it + 2) % QW_STAGES, fa); mma(it + 1, fb);
    }
    if (it < nit) { lds_barrier(); mma(it, fa); }
    // d^-1/2 dQ into the (now idle) stages as [query row 128][channel]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float* o = sm + ((2 * wv + jj) * 16 + ll) * D + 4 * q;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        *reinterpret_cast<float4*>(o + 16 * kt) = make_float4(dQ[jj][kt][0] * a.scale, dQ[jj][kt][1] * a.scale, dQ[jj][kt][2] * a.scale, dQ[jj][kt][3] * a.scale);
    }
  }
  // dQ rows of d_qkv (channel k*8 + h): consecutive lanes consecutive channels of one row
  lds_barrier();
  for (int p = tid; p < 16 * QW_TILES * D; p += 512) {
    const int row = p / D, k = p % D, l = lt0 * 16 + row;
    if (l < N && ABL(4)) a.d_qkv[((size_t)b * N + l) * 3 * DH + k * AH + h] = sm[p];
  }
}

#include "egt_pair.h"   // the fused pair operator of the same geometry (k_pair_fwd / k_pair_bwd)

// ------------------------------------------------------------------ host glue --
extern "C" int egt_attn_mfma_supported(const egt_attn_desc* d, int need_a_tild) {
  if (!d || d->dtype != EGT_F32 || d->H != AH) return 0;
  if (d->d != 16 && d->d != 32 && d->d != 64) return 0;
  if (d->flags & EGT_F_SCALE_DEGREE) return 0;
  if ((d->flags & EGT_F_TRAINING) && d->attn_dropout > 0.0f) return 0;
  if (need_a_tild) return 0;
  if ((size_t)d->B * d->N * d->N * d->H > 0xFFFFFFFFull) return 0;
  if (d->N > 2048) return 0;   // the forward keeps two per-key tables of N floats in LDS beside its stages
  return 1;
}

static int np_of(int N) { return (N + 15) & ~15; }

// both directions: six packed arrays, the per-row constants and the dA tiles; forward alone: its three arrays
extern "C" size_t egt_attn_mfma_workspace_bytes(const egt_attn_desc* d) {
  if (!egt_attn_mfma_supported(d, 0)) return 0;
  const size_t NP = np_of(d->N), arr = (size_t)d->B * AH * NP * d->d;
  return (PK_COUNT * arr + (size_t)d->B * AH * NP * 4 + (size_t)d->B * AH * NP * NP) * sizeof(float);
}

extern "C" size_t egt_attn_mfma_fwd_workspace_bytes(const egt_attn_desc* d) {
  if (!egt_attn_mfma_supported(d, 0)) return 0;
  if (d->reserved & EGT_ATTN_WS_SHARED) return egt_attn_mfma_workspace_bytes(d);   // the forward then packs the backward's arrays too
  return (size_t)PK_FWD_COUNT * d->B * AH * np_of(d->N) * d->d * sizeof(float);
}

static int fill(const egt_attn_desc* desc, const void* qkv, const void* E, const void* G,
                const uint8_t* key_mask, const void* attn_mask, const uint8_t* rand_mask, void* workspace,
                AttnMfmaArgs& a) {
  if (!egt_attn_mfma_supported(desc, 0)) EGT_FAIL(EGT_E_SHAPE, "configuration not covered by the MFMA inner-op kernel");
  if (!qkv || !workspace) EGT_FAIL(EGT_E_NULL, "qkv/workspace is NULL");
  if (desc->reserved & ~EGT_ATTN_WS_SHARED) EGT_FAIL(EGT_E_FLAGS, "egt_attn_desc.reserved: unknown bits 0x%x (EGT_ATTN_WS_SHARED is the only one)", desc->reserved);
  if ((desc->flags & EGT_F_EDGE_INPUT) && !E) EGT_FAIL(EGT_E_NULL, "edge_input set but E is NULL");
  if ((desc->flags & EGT_F_GATE_INPUT) && !G) EGT_FAIL(EGT_E_NULL, "gate_input set but G is NULL");
  if ((desc->flags & EGT_F_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "attn_mask set but M is NULL");
  a = AttnMfmaArgs{};
  a.B = desc->B; a.N = desc->N; a.NP = np_of(desc->N); a.d = desc->d; a.flags = desc->flags;
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
  a.pk = (float*)workspace;
  return EGT_OK;
}

template <int D>
static void launch_pack(const AttnMfmaArgs& a, hipStream_t st) {
  const size_t lds = (size_t)AH * (16 * (D == 16 ? 16 : 80) + 4) * 4;
  EGT_MAX_LDS_ONCE(k_attn_pack<D>);
  EGT_LAUNCH("k_attn_pack", k_attn_pack<D>, dim3(a.B * (a.NP / 16)), dim3(512), lds, st, a);
}

// 1 / 2: the straight-line instances (see Feat), 0: the run-time-switched one (EGT_ATTN_GENERIC forces it: tests)
static int variant_of(const AttnMfmaArgs& a, bool bwd) {
  static const bool generic = getenv("EGT_ATTN_GENERIC") != nullptr;
  const bool main_cfg = a.E && (a.flags & EGT_F_GATE_INPUT) && a.G && !a.M && a.km && (a.flags & EGT_F_CLIP) && !a.rm &&
                        (!bwd || (a.d_h_ext && a.d_E && a.d_G));
  if (!main_cfg || generic) return 0;
  return a.rng_rm ? 2 : 1;
}

template <int D, int V>
static void launch_fwd_v(const AttnMfmaArgs& a, hipStream_t st) {
  const size_t ts = (size_t)FQ * 4 * PT_PL;
  const size_t lds = OPS_BYTES + ((size_t)2 * ((a.NP + 16 + 3) & ~3) + (size_t)(2 * (V ? 2 : 3) + 2) * ts) * 4;
  const int lgroups = (a.NP / 16 + FQ - 1) / FQ;
  EGT_MAX_LDS_ONCE(k_attn_mfma_fwd<D, V>);
  EGT_LAUNCH("k_attn_mfma_fwd", (k_attn_mfma_fwd<D, V>), dim3(a.B * lgroups * 2), dim3(512), lds, st, a);
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
  // EGT_ATTN_WS_SHARED: the workspace is egt_attn_mfma_workspace_bytes() large and will be handed to the backward:
  // pack the backward's q / k / v arrays too, once
  a.pack_what = PACK_Q | PACK_KH | PACK_VT | ((desc->reserved & EGT_ATTN_WS_SHARED) ? (PACK_KT | PACK_VH) : 0);
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
  const size_t lds = OPS_BYTES + ((size_t)2 * 4 * 64 + (size_t)(2 * 3 + 4) * BK * 4 * PT_PL) * 4;
  const int mgroups = (a.NP / 16 + BK - 1) / BK;
  EGT_MAX_LDS_ONCE(k_attn_mfma_bwd_kv<D, V>);
  EGT_LAUNCH("k_attn_mfma_bwd_kv", (k_attn_mfma_bwd_kv<D, V>), dim3(a.B * mgroups * 2), dim3(512), lds, st, a);
}
template <int D>
static void launch_bwd(const AttnMfmaArgs& a, hipStream_t st) {
  launch_pack<D>(a, st);   // (also the per-row constants, delta = sum_k dO*O among them)
  switch (variant_of(a, true)) {
    case 1: launch_bwd_kv_v<D, 1>(a, st); break;
    case 2: launch_bwd_kv_v<D, 2>(a, st); break;
    default: launch_bwd_kv_v<D, 0>(a, st); break;
  }
  {
    const int qgroups = (a.NP / 16 + QW_TILES - 1) / QW_TILES;
    const size_t lds = (size_t)QW_STAGES * (2 * (D / 16) + 2 * QW_TILES) * 1024;
    EGT_MAX_LDS_ONCE(k_attn_mfma_bwd_q<D>);
    EGT_LAUNCH("k_attn_mfma_bwd_q", (k_attn_mfma_bwd_q<D>), dim3(a.B * AH * qgroups), dim3(512), lds, st, a);
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
  const size_t arr = (size_t)a.B * AH * a.NP * a.d;
  a.stats2 = a.pk + (size_t)PK_COUNT * arr;
  a.ws_dA = a.stats2 + (size_t)a.B * AH * a.NP * 4;
  a.pack_what = PACK_O | ((desc->reserved & EGT_ATTN_WS_SHARED) ? 0 : (PACK_Q | PACK_KH | PACK_KT | PACK_VH));
  switch (desc->d) {
    case 16: launch_bwd<16>(a, (hipStream_t)stream); break;
    case 32: launch_bwd<32>(a, (hipStream_t)stream); break;
    default: launch_bwd<64>(a, (hipStream_t)stream); break;
  }
  EGT_HIP_LAUNCH_CHECK("egt_attn_mfma_bwd");
  return EGT_OK;
}

// ---- fused pair operator (egt_pair.h): (V_att, e') = pair(QKV, e, mask) and its backward -------------------------------
extern "C" int egt_pair_supported(const egt_block_desc* d) {
  if (!d || d->dtype != EGT_F32 || d->H != AH) return 0;
  if (d->d != 64 || d->De != 32) return 0;                                   // the instantiated geometry (BASELINE config 5)
  if (!(d->flags & EGT_BF_GATE)) return 0;                                    // gated attention with edge bias ('residual' edge channels)
  if (d->flags & (EGT_BF_ATTN_MASK | EGT_BF_NO_EDGE_LN | EGT_BF_SEED_DEVICE)) return 0;
  if (d->B < 1 || d->N < 1 || d->N > 2048) return 0;
  if ((size_t)d->B * d->N * d->N * AH > 0xFFFFFFFFull) return 0;              // 32-bit element index of the mask hash
  return 1;
}

static size_t pair_partial_floats(const egt_block_desc* d) {   // per-workgroup partials + one reduced image each
  const size_t nwg = (size_t)d->B * (np_of(d->N) / 16);
  return (nwg + 1) * ((size_t)d->De * 16 + 16) + (nwg + 1) * ((size_t)AH * d->De + d->De) + 256   // + the 1 KB store dump (PairArgs::dump)
         + (3 * ((size_t)d->De / 16) + 1) * 64 * 4;                                                      // + the prepared weight table (PairArgs::wprep)
}

// packed operand arrays (all six), row constants, dA tiles, parameter-gradient partials.  The forward writes the q / k / v arrays;
// with desc->reserved & EGT_ATTN_WS_SHARED the caller hands the SAME untouched workspace to egt_pair_bwd, which then adds only dO
extern "C" size_t egt_pair_workspace_bytes(const egt_block_desc* d) {
  if (!egt_pair_supported(d)) return 0;
  const size_t NP = np_of(d->N), arr = (size_t)d->B * AH * NP * d->d;
  return (PK_COUNT * arr + (size_t)d->B * AH * NP * 4 + (size_t)d->B * AH * NP * NP + pair_partial_floats(d)) * sizeof(float);
}

static int pair_fill(const egt_block_desc* d, const egt_block_params* P, const void* qkv, const void* e, const uint8_t* key_mask,
                     void* workspace, AttnMfmaArgs& a, PairArgs& pa) {
  if (!egt_pair_supported(d)) EGT_FAIL(EGT_E_SHAPE, "configuration not covered by the fused pair operator (d = 64, De = 32, H = 8, gated, fp32)");
  if (!P || !qkv || !e || !workspace) EGT_FAIL(EGT_E_NULL, "params/qkv/e/workspace is NULL");
  if (!P->norm_edge_gamma || !P->norm_edge_beta || !P->attention_gates_kernel || !P->attention_gates_bias || !P->dense_edge_b_kernel ||
      !P->dense_edge_b_bias || !P->dense_edge_r_kernel || !P->dense_edge_r_bias)
    EGT_FAIL(EGT_E_NULL, "an edge-side parameter pointer is NULL");
  if (d->reserved & ~EGT_ATTN_WS_SHARED) EGT_FAIL(EGT_E_FLAGS, "egt_block_desc.reserved: unknown bits 0x%x", d->reserved);
  a = AttnMfmaArgs{};
  a.B = d->B; a.N = d->N; a.NP = np_of(d->N); a.d = d->d;
  a.flags = EGT_F_EDGE_INPUT | EGT_F_GATE_INPUT | ((d->flags & EGT_BF_CLIP) ? EGT_F_CLIP : 0);
  a.clip_lo = d->clip_lo; a.clip_hi = d->clip_hi;
  a.scale = 1.0f / sqrtf((float)d->d);
  a.rm_thr = egt_threshold24(d->random_mask_prob);
  a.s0 = (uint32_t)(d->seed & 0xFFFFFFFFull); a.s1 = (uint32_t)(d->seed >> 32);
  a.rng_rm = ((d->flags & EGT_BF_TRAINING) && d->random_mask_prob > 0.0f) ? 1 : 0;
  a.qkv = (const float*)qkv; a.km = key_mask;
  a.pk = (float*)workspace;
  pa = PairArgs{};
  pa.De = d->De; pa.ln_eps = d->ln_eps;
  pa.e = (const float*)e;
  pa.gamma = (const float*)P->norm_edge_gamma; pa.beta = (const float*)P->norm_edge_beta;
  pa.Wg = (const float*)P->attention_gates_kernel; pa.bg = (const float*)P->attention_gates_bias;
  pa.We = (const float*)P->dense_edge_b_kernel; pa.be = (const float*)P->dense_edge_b_bias;
  pa.Wr = (const float*)P->dense_edge_r_kernel; pa.br = (const float*)P->dense_edge_r_bias;
  {   // the last 1 KB of the workspace
    const size_t NP = np_of(d->N), arr = (size_t)d->B * AH * NP * d->d;
    const size_t wt = (3 * ((size_t)d->De / 16) + 1) * 64 * 4;
    pa.wprep = (float*)workspace + PK_COUNT * arr + (size_t)d->B * AH * NP * 4 + (size_t)d->B * AH * NP * NP + pair_partial_floats(d) - wt;
    pa.dump = pa.wprep - 256;
  }
  return EGT_OK;
}

extern "C" int egt_pair_fwd(const egt_block_desc* desc, const egt_block_params* params, const void* qkv, const void* e,
                            const uint8_t* key_mask, void* v_att, void* e_out, void* rowstats, void* workspace, void* stream) {
  AttnMfmaArgs a; PairArgs pa;
  int rc = pair_fill(desc, params, qkv, e, key_mask, workspace, a, pa);
  if (rc) return rc;
  if (!v_att || !e_out || !rowstats) EGT_FAIL(EGT_E_NULL, "v_att/e_out/rowstats is NULL");
  a.v_att = (float*)v_att; a.rowstats = (float*)rowstats;
  pa.e_out = (float*)e_out;
  a.pack_what = PACK_Q | PACK_KH | PACK_VT | ((desc->reserved & EGT_ATTN_WS_SHARED) ? (PACK_KT | PACK_VH) : 0);
  hipStream_t st = (hipStream_t)stream;
  launch_pack<64>(a, st);
  constexpr int D = 64, DE = 32, HS = (D / 16) * 256;
  EGT_LAUNCH("k_pair_prep", k_pair_prep<DE>, dim3(1), dim3(64), 0, st, pa);
  const size_t lds = ((size_t)2 * AH * HS + 2 * (size_t)((a.NP + 16 + 3) & ~3) + (size_t)6 * AH * PT_PL + 16 + DE) * sizeof(float);
  const int grid = a.B * (a.NP / 16);
#define PAIR_FWD(V_, F_) do { EGT_MAX_LDS_ONCE(k_pair_fwd<D, DE, V_, F_>); \
    EGT_LAUNCH("k_pair_fwd", (k_pair_fwd<D, DE, V_, F_>), dim3(grid), dim3(64 * PR_WAVES), lds, st, a, pa); } while (0)
  const bool full = a.N % 16 == 0;
  if (a.rng_rm) { if (full) PAIR_FWD(2, true); else PAIR_FWD(2, false); }
  else { if (full) PAIR_FWD(1, true); else PAIR_FWD(1, false); }
#undef PAIR_FWD
  EGT_HIP_LAUNCH_CHECK("egt_pair_fwd");
  return EGT_OK;
}

// rowstats is read AND written (slot 3 receives delta).  Every edge-side pointer of `grads` is written; d_e may alias d_e_out.
extern "C" int egt_pair_bwd(const egt_block_desc* desc, const egt_block_params* params, const void* qkv, const void* e,
                            const uint8_t* key_mask, const void* v_att, void* rowstats, const void* d_v_att, const void* d_e_out,
                            void* d_qkv, void* d_e, const egt_block_params* grads, void* workspace, void* stream) {
  AttnMfmaArgs a; PairArgs pa;
  int rc = pair_fill(desc, params, qkv, e, key_mask, workspace, a, pa);
  if (rc) return rc;
  if (!v_att || !rowstats || !d_v_att || !d_e_out || !d_qkv || !d_e || !grads) EGT_FAIL(EGT_E_NULL, "v_att/rowstats/d_v_att/d_e_out/d_qkv/d_e/grads is NULL");
  if (!grads->norm_edge_gamma || !grads->norm_edge_beta || !grads->attention_gates_kernel || !grads->attention_gates_bias ||
      !grads->dense_edge_b_kernel || !grads->dense_edge_b_bias || !grads->dense_edge_r_kernel || !grads->dense_edge_r_bias)
    EGT_FAIL(EGT_E_NULL, "an edge-side gradient pointer is NULL");
  a.v_att_in = (const float*)v_att; a.rowstats = (float*)rowstats;
  a.d_v_att = (const float*)d_v_att; a.d_qkv = (float*)d_qkv;
  pa.d_e_out = (const float*)d_e_out; pa.d_e = (float*)d_e;
  const size_t arr = (size_t)a.B * AH * a.NP * a.d;
  a.stats2 = a.pk + (size_t)PK_COUNT * arr;
  a.ws_dA = a.stats2 + (size_t)a.B * AH * a.NP * 4;
  constexpr int D = 64, DE = 32, HS = (D / 16) * 256;
  constexpr int PSZ1 = DE * 16 + 16, PSZ2 = AH * DE + DE;
  const int nwg = a.B * (a.NP / 16);
  pa.part_proj = a.ws_dA + (size_t)a.B * AH * a.NP * a.NP;
  pa.part_upd = pa.part_proj + (size_t)(nwg + 1) * PSZ1;
  a.pack_what = PACK_O | ((desc->reserved & EGT_ATTN_WS_SHARED) ? 0 : (PACK_Q | PACK_KH | PACK_KT | PACK_VH));
  hipStream_t st = (hipStream_t)stream;
  launch_pack<64>(a, st);   // (also the per-row constants, delta = sum_k dO*O among them)
  if (!(desc->reserved & EGT_ATTN_WS_SHARED))   // (a shared workspace still holds the forward's table: same parameter values by contract)
    EGT_LAUNCH("k_pair_prep", k_pair_prep<DE>, dim3(1), dim3(64), 0, st, pa);
  const size_t lds = ((size_t)2 * 4 * 2 * HS + (size_t)2 * 3 * AH * PT_PL + (size_t)2 * AH * 64 + (size_t)4 * 2 * 16 * DE + (size_t)3 * (DE / 16) * 64 * 4 + 64 * 4 + 4 * 2 * 4 * 16) * sizeof(float);
#define PAIR_BWD(V_, F_) do { EGT_MAX_LDS_ONCE(k_pair_bwd<D, DE, V_, F_>); \
    EGT_LAUNCH("k_pair_bwd", (k_pair_bwd<D, DE, V_, F_>), dim3(nwg), dim3(64 * PR_WAVES), lds, st, a, pa); } while (0)
  const bool full = a.N % 16 == 0;
  if (a.rng_rm) { if (full) PAIR_BWD(2, true); else PAIR_BWD(2, false); }
  else { if (full) PAIR_BWD(1, true); else PAIR_BWD(1, false); }
#undef PAIR_BWD
  {
    const int qgroups = (a.NP / 16 + QW_TILES - 1) / QW_TILES;
    const size_t ldsq = (size_t)QW_STAGES * (2 * (D / 16) + 2 * QW_TILES) * 1024;
    EGT_MAX_LDS_ONCE(k_attn_mfma_bwd_q<D>);
    EGT_LAUNCH("k_attn_mfma_bwd_q", (k_attn_mfma_bwd_q<D>), dim3(a.B * AH * qgroups), dim3(512), ldsq, st, a);
  }
  egt_edge_finish_param_grads(DE, pa.gamma, pa.beta, pa.Wg, pa.We, pa.part_proj, pa.part_upd, nwg,
                              pa.part_proj + (size_t)nwg * PSZ1, pa.part_upd + (size_t)nwg * PSZ2,
                              (float*)grads->norm_edge_gamma, (float*)grads->norm_edge_beta, (float*)grads->attention_gates_kernel,
                              (float*)grads->attention_gates_bias, (float*)grads->dense_edge_b_kernel, (float*)grads->dense_edge_b_bias,
                              (float*)grads->dense_edge_r_kernel, (float*)grads->dense_edge_r_bias, st);
  EGT_HIP_LAUNCH_CHECK("egt_pair_bwd");
  return EGT_OK;
}

#ifdef EGT_ATTN_STAMPS
extern "C" int egt_attn_mfma_read_stamps(long long* host, int n) {
  long long tmp[3 * 8 * 16];
  if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_attn_stamps), sizeof(tmp)) != hipSuccess) return -1;
  for (int i = 0; i < n && i < 3 * 8 * 16; ++i) host[i] = tmp[i];
  return 0;
}
#endif
