// Fused pair operator for the large-head geometry (d = 64, H = 8; BASELINE config 5: N = 512, Dh = 512, De = 32):
//   (V_att, e') = pair(QKV, e, mask)   =   norm_edge -> attention_gates / dense_edge_b -> EGT -> dense_edge_r + res_edge
//   lib/models/graph_xformer_model_base.py:195-218 around lib/models/egt_layers.py:57-143
// with E, G, H_hat (and dE, dG, dH_ext in the backward) living only in LDS planes: the [B,N,N,8] tensors of the composed
// path (k_edge_proj_* -> k_attn_mfma_* -> k_edge_update_*) never reach HBM.  The node-side Dense layers around it
// (norm_mha, dense_qkv, dense_mha: [B N, 512] x [512, 1536] GEMMs) stay library GEMMs.
//
// Included by egt_attn_mfma.hip (same translation unit: packed operand arrays, plane layouts, LDS-DMA helpers, k_attn_pack
// and k_attn_mfma_bwd_q are shared with the inner-op kernels).
//
// Workgroup = (graph, 16 query rows [forward] / 16 keys [backward], ALL 8 heads), 8 waves, one workgroup per CU:
//   * 4 ATTENTION waves (one per SIMD), two heads each: QK^T / A.V (backward: S, dP, dV, dK) on v_mfma_f32_16x16x4_f32 with
//     their Q / dO-side operands in registers, the streamed operand tiles in LDS, the [16 x 16] pair values of their heads
//     read from / written to head-major LDS planes;
//   * 4 EDGE waves (one per SIMD), four rows of the 16 x 16 pair tile each: the e tile HBM -> registers (lane (p, q) = pair p,
//     channels 16 t + 4 q + {0..3}: the B-operand layout of the channel contractions), LayerNorm in registers, the
//     [gamma Wg | gamma We] projection on the matrix core -> G / E planes; H_hat plane -> dense_edge_r on the matrix core ->
//     e' = e + ... stored from the registers that loaded e; and every vector-memory instruction of the workgroup: the
//     attention waves' operand tiles by LDS-DMA, e two tiles ahead.
// Both roles issue MFMAs, so the matrix pipe of a SIMD always has a second wave to draw from while the other is in its
// VALU / LDS phase (the inner-op kernels pair a compute wave with a loader wave that issues none).
// Two barriers per tile, neither drains vector memory.
#pragma once

struct PairArgs {
  int De;
  float ln_eps;
  const float *e, *gamma, *beta, *Wg, *bg, *We, *be, *Wr, *br;
  float* e_out;
  // backward
  const float* d_e_out;
  float *d_e, *part_proj, *part_upd;   // per-workgroup partials: [That[De][16] | s[16]] and [dWr[8][De] | dbr[De]]
  float* wprep;  // [wA | wB | wR][T][64 lanes][4] | bias [64][4]: the edge waves' lane-constant MFMA operands, LN-folded once per launch by k_pair_prep
  float* dump;   // 1 KB of workspace: where the lanes of pairs outside the graph send their stores.  Every store of the edge waves is
                 // issued unconditionally (no branch around it), so the compiler's vmcnt bookkeeping counts it: behind a store it cannot
                 // count (exec-masked block) the wait for an older prefetch would also wait for that store's completion
};

#define PR_WAVES 8
// LDS hand-off inside ONE wave (its DS operations complete in order): drain lgkmcnt, never vmcnt
__device__ __forceinline__ void lds_sync_() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
// The same hand-off WITHOUT the drain: the LDS executes one wave's DS instructions in program order, so a read issued behind a write of
// the same wave sees it and a write issued behind a read cannot overtake it -- only the compiler has to keep the order (the waits for
// the reads' results are its own, at their first use).  PAIR_LDS_NOWAIT=0 restores the drains (A/B).
#ifndef PAIR_LDS_NOWAIT
#define PAIR_LDS_NOWAIT 1
#endif
__device__ __forceinline__ void lds_order_() {
  if (PAIR_LDS_NOWAIT) { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
  else lds_sync_();
}
// the workgroup barrier of the pair kernels: nothing is scheduled across it (a plain asm barrier orders memory only: hipcc hoisted the
// next trip's LayerNorm arithmetic above it, i.e. in front of the wait for a tile that had only just been requested)
__device__ __forceinline__ void pair_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
#define PR_CLAMP(x) (FULL ? (x) : min((x), N - 1))
#ifndef PAIR_ABL
#define PAIR_ABL 0   // timing ablations of k_pair_fwd (results are wrong): 1 attention waves idle, 2 no dense_edge_r / stores, 4 no LN / projections, 8 no e requests, 16 no e' stores, 32 / 64 e loads / e' stores on the tile of trip 0 (cache hits)
#endif
#ifndef PAIR_BWD_NT
#define PAIR_BWD_NT 4   // k_pair_bwd cache-policy hints: 1 e tiles non-temporal, 2 de' first read, 4 de' second read (POST).  Measured (same box, us per launch):
                        // 0: 263.3-263.4, 1: 265.6-266.2, 4: 246.8-255.5, 5: 249.4-253.9 -> the re-read of de' (a tile nobody reads again) is non-temporal
#endif
#ifndef PAIR_ST_SC
#define PAIR_ST_SC 0   // k_pair_fwd: e' stores with scope bits (1: sc1, 2: sc0 sc1 -- write-through forms that drop the line from L2); A/B
#endif
#ifndef PAIR_NT_ST
#define PAIR_NT_ST 0   // k_pair_fwd: e' stores non-temporal (A/B)
#endif
#ifndef PAIR_STAGGER
#define PAIR_STAGGER 0   // k_pair_fwd: 1 = a workgroup starts on the key tile of its own row block instead of every workgroup on tile `it` in lock step
                        // (the 16 rows of a workgroup are 64 KB apart: a test for HBM channel camping) -- measured: no difference (145.1-146.2 vs 145.4-146.7 us)
#endif
#ifndef PAIR_BWD_ABL
#define PAIR_BWD_ABL 0   // timing ablations of k_pair_bwd (results are wrong): 1 first reads of e / de', 2 de stores, 4 the POST re-read of de' -- on the tile of trip 0 (cache hits)
#endif
#ifndef PAIR_EDGE_PRIO
#define PAIR_EDGE_PRIO 2
#endif
#ifndef PAIR_NT_E
#define PAIR_NT_E 0   // 1: the forward's e tiles by non-temporal loads (read once; measured before adopting)
#endif
// a wave-uniform pointer the compiler cannot prove uniform -> scalar registers (LDS-DMA takes its base address from an SGPR pair)
__device__ __forceinline__ const float* uni_ptr(const float* p) {
  const uint64_t v = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const float*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// four consecutive 1 KB pieces (4 KB of HBM -> 4 KB of LDS) in one asm block: M0 saved / restored once, advanced by 1 KB per piece; the
// lane offsets of pieces 1..3 are doff + 1024 i (doff < 1024), kept in registers by the caller.  (Per piece the single-piece form costs
// 5 scalar instructions + the 64-bit source add: 112 scalar instructions per key tile in an attention wave.)
__device__ __forceinline__ void dma_pieces4(unsigned lds, const float* src, unsigned o0, unsigned o1, unsigned o2, unsigned o3) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
               "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(src), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(lds) : "memory", "scc");
}
typedef float pr_nt_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 pr_ld4(const char* p) {
  if (NT) { const pr_nt_v4f t = __builtin_nontemporal_load(reinterpret_cast<const pr_nt_v4f*>(p)); return make_float4(t[0], t[1], t[2], t[3]); }
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ float4 egt_ld4_nt_(const float* p) {
  if (PAIR_NT_E) { const pr_nt_v4f t = __builtin_nontemporal_load(reinterpret_cast<const pr_nt_v4f*>(p)); return make_float4(t[0], t[1], t[2], t[3]); }
  return *reinterpret_cast<const float4*>(p);
}
template <int DE> struct PairGeo { static constexpr int T = DE / 16; };

// LayerNorm of a lane's fragments (two-pass moments like tf.nn.moments, as egt_tile.h: ln_frags); returns rstd
template <int T_>
__device__ __forceinline__ float pair_ln(float4 (&x)[T_], float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < T_; ++t) s += (x[t].x + x[t].y) + (x[t].z + x[t].w);
  const float mu = pair_sum_q(s) * (1.0f / (16 * T_));
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < T_; ++t) {
    x[t].x -= mu; x[t].y -= mu; x[t].z -= mu; x[t].w -= mu;
    v = fmaf(x[t].x, x[t].x, v); v = fmaf(x[t].y, x[t].y, v);
    v = fmaf(x[t].z, x[t].z, v); v = fmaf(x[t].w, x[t].w, v);
  }
  const float rstd = rsqrtf(pair_sum_q(v) * (1.0f / (16 * T_)) + eps);
#pragma unroll
  for (int t = 0; t < T_; ++t) { x[t].x *= rstd; x[t].y *= rstd; x[t].z *= rstd; x[t].w *= rstd; }
  return rstd;
}
__device__ __forceinline__ float f4get(const float4& v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; }

// LN-folded projection weights of an edge wave (A operands, lane (i = lane&15, q)): wA[t][r] = gamma_c Wcat[c][i], c = 16 t + 4 q + r;
// bias[r] = b_i' + sum_c beta_c Wcat[c][i'] for the lane's OUTPUT channels i' = 4 q + r.  Wcat = [attention_gates | dense_edge_b].
template <int DE>
__device__ __forceinline__ void pair_fold_weights(const PairArgs& pa, int i, int q, float (&wA)[PairGeo<DE>::T][4], float (&bias)[4]) {
  constexpr int T = PairGeo<DE>::T;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * q + r;
      wA[t][r] = (i < 8 ? pa.Wg[c * 8 + i] : pa.We[c * 8 + i - 8]) * pa.gamma[c];
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = 4 * q + r;
    const float* W = o < 8 ? pa.Wg + o : pa.We + (o - 8);
    float b = o < 8 ? pa.bg[o] : pa.be[o - 8];
#pragma unroll 8
    for (int c = 0; c < DE; ++c) b = fmaf(pa.beta[c], W[c * 8], b);
    bias[r] = b;
  }
}

// One wave, once per launch: the LN-folded projection weights in the edge waves' A-operand layout (lane (i = lane&15, q)) -> pa.wprep.
// (Inside the pair kernels this cost every workgroup ~10 us of dependent global round trips at its head: the bias sums over the channels.)
template <int DE>
__global__ void __launch_bounds__(64) k_pair_prep(PairArgs pa) {
  constexpr int T = PairGeo<DE>::T;
  const int lane = threadIdx.x, p = lane & 15, q = lane >> 4;
  float wA[T][4], bias[4];
  pair_fold_weights<DE>(pa, p, q, wA, bias);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float wB[4], wR[4];   // d ehat = Wp.dGE: A[c = 16 t + i][o = 4 q + s] = gamma_c Wcat[c][o];  dH_ext = Wr.de': A[h = i (< 8)][c = 16 t + 4 q + r] = Wr[h][c]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 16 * t + p, o = 4 * q + s;
      wB[s] = (o < 8 ? pa.Wg[c * 8 + o] : pa.We[c * 8 + o - 8]) * pa.gamma[c];
      const int c2 = 16 * t + 4 * q + s;
      wR[s] = p < 8 ? pa.Wr[p * DE + c2] : 0.f;
    }
    *reinterpret_cast<float4*>(pa.wprep + ((0 * T + t) * 64 + lane) * 4) = make_float4(wA[t][0], wA[t][1], wA[t][2], wA[t][3]);
    *reinterpret_cast<float4*>(pa.wprep + ((1 * T + t) * 64 + lane) * 4) = make_float4(wB[0], wB[1], wB[2], wB[3]);
    *reinterpret_cast<float4*>(pa.wprep + ((2 * T + t) * 64 + lane) * 4) = make_float4(wR[0], wR[1], wR[2], wR[3]);
  }
  *reinterpret_cast<float4*>(pa.wprep + (3 * T * 64 + lane) * 4) = make_float4(bias[0], bias[1], bias[2], bias[3]);
}

// ================================================================== forward =====
// One barrier per key tile.  An attention wave stages the K / V^T tiles of ITS OWN two heads by LDS-DMA (it is the only reader of those
// LDS regions: issue after its own last read, counted vmcnt before its own next read -- no cross-wave hand-off, one stage each);
// the planes cross the roles through two stages:
//   barrier(it): E / G planes of tile it written (edge), H_hat planes of tile it-1 written (attention)
//   tile it:  attention: S = K.Q^T -> DMA K(it+1) -> softmax x gates, H_hat -> stage it&1 -> A.V -> DMA V(it+1)
//             edge:      dense_edge_r + residual of tile it-1 (H_hat stage (it-1)&1) -> e(it+2) requested -> LN + projections of tile it+1 -> stage (it+1)&1
// (stamps of the first version -- DMA issued by the edge waves, two barriers -- profiles/r06_pair_stamps_v1.txt: the attention waves
//  waited at barriers for half of the launch while the edge waves issued the workgroup's vector memory)
// FULL: N is a multiple of 16 (no ragged tile: every row / key of every tile exists -- the clamps and the store-address selects fold away)
template <int D, int DE, int V, bool FULL>
__global__ void __launch_bounds__(64 * PR_WAVES, 2) k_pair_fwd(AttnMfmaArgs a, PairArgs pa) {
  constexpr int KT = D / 16, DH = D * AH, T = PairGeo<DE>::T, HS = KT * 256;   // HS: floats of one head's operand tile
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, NP = a.NP, mtiles = NP / 16;
  const int KA = (NP + 16 + 3) & ~3;
  float* Kb = sm;                      // [8][HS]
  float* Vb = Kb + AH * HS;            // [8][HS]
  float* kaddL = Vb + AH * HS;         // [KA]
  float* kaddG = kaddL + KA;
  float* In = kaddG + KA;              // [2 stages][E | G][8][PT_PL]
  float* Hp = In + 4 * AH * PT_PL;     // [2 stages][8][PT_PL]
  float* ctab = Hp + 2 * AH * PT_PL;   // [16] projection bias (output 4q + r) | [DE] dense_edge_r bias: lane-dependent through q only, 12 registers otherwise
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);   // the row blocks of a graph share an XCD's L2 (K / V^T of the graph: 2 MB)
  const int b = __builtin_amdgcn_readfirstlane(wg / mtiles), l0 = __builtin_amdgcn_readfirstlane((wg % mtiles) * 16);   // (the division runs on the VALU: back to scalar registers)
  const size_t arr = (size_t)a.B * AH * NP * D;
  const bool clip = (a.flags & EGT_F_CLIP) != 0;
  // Key-tile order: trip `it` works on key tile (it + row block) mod mtiles.  The online softmax does not care about the order of the keys,
  // HBM does: the 16 rows of a workgroup are N De 4 bytes apart (64 KB at config 5), so with every workgroup on key tile `it` in lock
  // step the whole chip reads and writes ONE 2 KB window of every 64 KB period of the e tensor at a time (a fraction of the channels);
  // staggered, the 32 row blocks of a graph cover the period.
  const int tstart = PAIR_STAGGER ? (l0 >> 4) % mtiles : 0;
  auto mt_of = [&](int it) __attribute__((always_inline)) { const int t = it + tstart; return t < mtiles ? t : t - mtiles; };
  STAMP_DECL;

  if (wv >= 4) {
    // --------------------------------------------------------------------- edge waves ----
    const int j = wv - 4, p = lane & 15, q = lane >> 4;
#if PAIR_EDGE_PRIO
    __builtin_amdgcn_s_setprio(PAIR_EDGE_PRIO);   // the edge waves are the longer role: the SIMD's issue arbitration (priority, then age) favours them
#endif
    for (int m = tid - 256; m < NP + 16; m += 256) {
      float ka = 0.f;
      if (m >= N) ka = KEY_OFF;
      else if (a.km && a.km[(size_t)b * N + m] == 0) ka = -EGT_NEG;
      kaddL[m] = ka;
      kaddG[m] = m >= N ? 3.0e38f : -ka * L2E;
    }
    float wA[T][4];
    {   // LN-folded [gamma Wg | gamma We] and its bias, prepared once per launch (k_pair_prep)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(pa.wprep + (t * 64 + lane) * 4);
        wA[t][0] = v.x; wA[t][1] = v.y; wA[t][2] = v.z; wA[t][3] = v.w;
      }
      // (every edge wave writes the same values: a benign race; read back only behind the first barrier)
      if (p == 0) *reinterpret_cast<float4*>(ctab + 4 * q) = *reinterpret_cast<const float4*>(pa.wprep + (3 * T * 64 + lane) * 4);
      if (lane < DE / 4) *reinterpret_cast<float4*>(ctab + 16 + 4 * lane) = *reinterpret_cast<const float4*>(pa.br + 4 * lane);
      lds_sync_();
    }
    // dense_edge_r: A[c = 16 t + i][h = q + 4 s] = Wr[h][c]; bias br[16 t + 4 q + r] on the lane's output channels (ctab)
    float wU[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      wU[t][0] = pa.Wr[q * DE + 16 * t + p];
      wU[t][1] = pa.Wr[(q + 4) * DE + 16 * t + p];
    }
    // row r of the wave: a wave-uniform row pointer (scalar registers) + one 32-bit lane offset (bytes) per tile
    const char* erow[4];
    char* orow[4];
    bool rowok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int l = l0 + 4 * j + r;
      rowok[r] = FULL || l < N;
      const size_t ro = (((size_t)b * N + PR_CLAMP(l)) * N * DE) * sizeof(float);
      erow[r] = reinterpret_cast<const char*>(pa.e) + ro;
      orow[r] = reinterpret_cast<char*>(pa.e_out) + ro;
    }
    char* dumpl = reinterpret_cast<char*>(pa.dump) + lane * 16;
    struct ESet { float4 x[4][T]; };
    auto eload = [&](ESet& s, int mt) __attribute__((always_inline)) {
      if (PAIR_ABL & 32) mt = 0;   // (timing ablation: every trip reads the tile of trip 0 -- cache-resident)
      const uint32_t mo = ((uint32_t)PR_CLAMP(16 * mt + p) * DE + 4 * q) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) s.x[r][t] = egt_ld4_nt_(reinterpret_cast<const float*>(erow[r] + mo + 64 * t));
    };
    // LN + projections of a tile from set s -> E / G planes of stage st
    auto project = [&](const ESet& s, int st) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float4 x[T];
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = s.x[r][t];
        pair_ln<T>(x, pa.ln_eps);
        const float4 b4 = *reinterpret_cast<const float4*>(ctab + 4 * q);
        v4f acc = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int t = 0; t < T; ++t) {
          acc = MFMA(wA[t][0], x[t].x, acc);
          acc = MFMA(wA[t][1], x[t].y, acc);
          acc = MFMA(wA[t][2], x[t].z, acc);
          acc = MFMA(wA[t][3], x[t].w, acc);
        }
        // lane holds outputs 4q..4q+3 of pair (row 4j + r, key p): q < 2 gates of heads 4q + i (plane set 1), q >= 2 edge bias of heads 4(q-2) + i (set 0)
        float* pl = In + st * (2 * AH * PT_PL) + (q < 2 ? AH * PT_PL : 0) + (4 * (q & 1)) * PT_PL + pt_off(4 * j + r, p);
        pl[0] = acc[0]; pl[PT_PL] = acc[1]; pl[2 * PT_PL] = acc[2]; pl[3 * PT_PL] = acc[3];
      }
    };
    // e' = e + H_hat.Wr + br of tile mt from set s (the registers that loaded e) and the H_hat planes of stage mt & 1
    auto update_compute = [&](const ESet& s, int itx, ESet& ov) __attribute__((always_inline)) {   // itx: the trip that produced the planes
      // all eight plane reads, then the sixteen MFMAs as eight independent chains issued crosswise (no dependent back-to-back pair), then
      // the eight stores: written row by row the compiler keeps row r's read -> MFMA -> MFMA -> add -> store chain together and the wave
      // sits out every latency of it in turn (ablation: 36 us of the 142 us launch)
      float h0[4], h1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* hp = Hp + (itx & 1) * (AH * PT_PL) + q * PT_PL + pt_off(4 * j + r, p);
        h0[r] = hp[0]; h1[r] = hp[4 * PT_PL];
      }
      v4f acc[4][T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 b4 = *reinterpret_cast<const float4*>(ctab + 16 + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r][t] = MFMA(wU[t][0], h0[r], ((v4f){b4.x, b4.y, b4.z, b4.w}));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[r][t] = MFMA(wU[t][1], h1[r], acc[r][t]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t)
          ov.x[r][t] = make_float4(s.x[r][t].x + acc[r][t][0], s.x[r][t].y + acc[r][t][1], s.x[r][t].z + acc[r][t][2], s.x[r][t].w + acc[r][t][3]);
    };
    // ... and its stores (timing ablation PAIR_ABL & 16: without them the launch is 38 us shorter -- 113 of 151 us -- although the write
    // bandwidth they need is a fraction of the chip's; non-temporal stores: +14 us)
    auto update_store = [&](const ESet& ov, int itx) __attribute__((always_inline)) {
      if (PAIR_ABL & 16) return;   // (timing ablation: no stores)
      const int m = 16 * ((PAIR_ABL & 64) ? 0 : mt_of(itx)) + p;   // (64: timing ablation, every trip stores to the tile of trip 0)
      const uint32_t mo = ((uint32_t)PR_CLAMP(m) * DE + 4 * q) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t) {
          char* dst = (FULL || (rowok[r] && m < N)) ? orow[r] + mo + 64 * t : dumpl;
          const pr_nt_v4f o4 = {ov.x[r][t].x, ov.x[r][t].y, ov.x[r][t].z, ov.x[r][t].w};
          if (PAIR_ST_SC == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(o4) : "memory");
          else if (PAIR_ST_SC == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(o4) : "memory");
          else *reinterpret_cast<pr_nt_v4f*>(dst) = o4;
        }
    };
    auto update = [&](const ESet& s, int itx) __attribute__((always_inline)) { ESet ov; update_compute(s, itx, ov); update_store(ov, itx); };
    ESet S0, S1, S2;
    eload(S0, mt_of(0));
    eload(S1, mt_of(min(1, mtiles - 1)));
    project(S0, 0);
    pair_barrier();             // barrier(0)
    // The set of e(k) is S[k % 3]; trip it updates tile it-1 from S[(it+2) % 3] (then refilled with e(it+2)) and projects e(it+1) from
    // S[(it+1) % 3]; e(it) waits in the third set.  The first and the last trip are peeled: the steady-state trip is branch-free
    // straight-line code whose vector-memory order is pinned (update's stores, then the request of e(it+2) a whole trip ahead of its
    // first use, then the projection) -- at a control-flow merge the compiler's vmcnt bookkeeping takes the conservative side and
    // drains the prefetch queue, and a load it moves behind the projection loses the trip of latency cover.
    // (pin: the values of e(it+1) are "produced" here, behind the barrier -- pure arithmetic on them cannot be hoisted above this
    //  point, i.e. into the previous trip in front of the wait for requests that were only just issued; the requests are a trip old here)
    auto pin = [&](ESet& s_) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < T; ++t)
          asm volatile("" : "+v"(s_.x[r][t].x), "+v"(s_.x[r][t].y), "+v"(s_.x[r][t].z), "+v"(s_.x[r][t].w));
    };
    auto steady = [&](int it, ESet& done, ESet& nxt) __attribute__((always_inline)) {
      STAMP(7);
      pin(nxt);
      if (!(PAIR_ABL & 2)) update(done, it - 1);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0);
      // (requesting e(it+2) BEFORE the stores -- vmcnt retires in order, so that the next trip's wait for it would not cover them -- was
      //  built and measured: 159-160 us against 146-147; the order below stays)
      if (!(PAIR_ABL & 8)) eload(done, mt_of(min(it + 2, mtiles - 1)));
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      if (!(PAIR_ABL & 4)) project(nxt, (it + 1) & 1);
      STAMP(2);
      pair_barrier();           // barrier(it+1)
      STAMP(3);
    };
    {   // trip 0: nothing to update yet
      eload(S2, mt_of(min(2, mtiles - 1)));
      __builtin_amdgcn_sched_barrier(0);
      if (mtiles > 1) project(S1, 1);
      pair_barrier();
    }
    int it = 1;
    for (; it + 3 <= mtiles - 1; it += 3) {
      steady(it, S0, S2);
      steady(it + 1, S1, S0);
      steady(it + 2, S2, S1);
    }
    if (it <= mtiles - 2) { steady(it, S0, S2); ++it; }
    if (it <= mtiles - 2) { steady(it, S1, S0); ++it; }
    if (mtiles > 1) {          // trip mtiles-1: no tile left to project
      const int k = it % 3;
      if (k == 1) update(S0, it - 1); else if (k == 2) update(S1, it - 1); else update(S2, it - 1);
      pair_barrier();
    }
    {   // H_hat of the last tile: its set is S[(mtiles-1) % 3]
      const int k = (mtiles - 1) % 3;
      if (k == 0) update(S0, mtiles - 1); else if (k == 1) update(S1, mtiles - 1); else update(S2, mtiles - 1);
    }
    STAMP(8);
    STAMP_OUT(0);
  } else {
    // ---------------------------------------------------------------- attention waves ----
    const int w = wv, ll = lane & 15, q = lane >> 4;
    // the wave's own operand tiles: K rows / V^T of heads 2w, 2w + 1, four 1 KB pieces per tile
    const unsigned doff = dma_lane_off(lane);
    const float* Kh = uni_ptr(a.pk + PK_KH * arr + ((size_t)b * AH + 2 * w) * NP * D);
    const float* VT = uni_ptr(a.pk + PK_VT * arr + ((size_t)b * AH + 2 * w) * D * NP);
    const unsigned kdst = lds_addr(Kb) + (2 * w) * HS * 4, vdst = lds_addr(Vb) + (2 * w) * HS * 4;
    const unsigned doff1 = doff + 1024, doff2 = doff + 2048, doff3 = doff + 3072;
    auto dma_tile = [&](unsigned dst, const float* src) __attribute__((always_inline)) {   // src: [2 heads] x (KT pieces), head stride NP * D floats
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if constexpr (KT == 4) dma_pieces4(dst + hh * KT * 1024, src + (size_t)hh * NP * D, doff, doff1, doff2, doff3);
        else {
#pragma unroll
          for (int Tt = 0; Tt < KT; ++Tt) dma_piece(dst + (hh * KT + Tt) * 1024, src + (size_t)hh * NP * D + Tt * 256, doff);
        }
      }
    };
    dma_tile(kdst, Kh + (size_t)mt_of(0) * 16 * D);
    dma_tile(vdst, VT + (size_t)mt_of(0) * 16 * D);
    float Qr[2][4 * KT];
    {
      const int ltile = l0 >> 4;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float* Qh = a.pk + PK_QH * arr + ((size_t)b * AH + 2 * w + hh) * NP * D + (size_t)ltile * 16 * D;
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt) {
          const float4 v = *reinterpret_cast<const float4*>(Qh + 256 * Tt + ll * 16 + 4 * q);
          Qr[hh][4 * Tt + 0] = v.x; Qr[hh][4 * Tt + 1] = v.y; Qr[hh][4 * Tt + 2] = v.z; Qr[hh][4 * Tt + 3] = v.w;
        }
      }
    }
    v4f oacc[2][KT];
    float mrun[2], lrun[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      mrun[hh] = KEY_OFF; lrun[hh] = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) oacc[hh][kt] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    const int po4 = (2 * w) * PT_PL + pt_off(ll, 4 * q);                       // + hh * PT_PL: the lane's keys 4q..4q+3 of row ll
    const float* opl = Kb + (2 * w) * HS + ll * 16 + ((q ^ chunk_xor(ll)) << 2);   // + hh * HS + 256 T
    const int lq = PR_CLAMP(l0 + ll);
    vm_wait<0>();              // (Q fragments in registers, tiles of key tile 0 landed)
    pair_barrier();             // barrier(0)
    STAMP(7);
    for (int it = 0; it < mtiles; ++it) {
      const int m0 = 16 * mt_of(it);
      const bool more = it + 1 < mtiles;
      const float* Inb = In + (it & 1) * (2 * AH * PT_PL);
      if (PAIR_ABL & 1) { pair_barrier(); continue; }
      if (it > 0) vm_wait<2 * KT>();   // K(it) landed (requested a tile ago; only the V^T(it) pieces are younger)
      float4 kc[2][KT], e4[2], g4[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt) kc[hh][Tt] = *reinterpret_cast<const float4*>(opl + hh * HS + Tt * 256);
      const float4 ka4 = *reinterpret_cast<const float4*>(kaddL + m0 + 4 * q);
      float4 kg4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (V == 1) kg4 = *reinterpret_cast<const float4*>(kaddG + m0 + 4 * q);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        e4[hh] = *reinterpret_cast<const float4*>(Inb + hh * PT_PL + po4);
        g4[hh] = *reinterpret_cast<const float4*>(Inb + AH * PT_PL + hh * PT_PL + po4);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- S^T[m][l] = sum_k K[m][k] (d^-1/2 Q)[l][k], two heads ----
      v4f s[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) s[hh] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int Tt = 0; Tt < KT; ++Tt)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) s[hh] = MFMA(f4get(kc[hh][Tt], u), Qr[hh][4 * Tt + u], s[hh]);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0);
      // the K tiles of the next key tile: this wave was their only reader and its reads have returned (the MFMAs above consumed them)
      if (more) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dma_tile(kdst, Kh + (size_t)mt_of(it + 1) * 16 * D); }
      const float kav[4] = {ka4.x, ka4.y, ka4.z, ka4.w}, kgv[4] = {kg4.x, kg4.y, kg4.z, kg4.w};
      float pa_[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * w + hh;
        float x[4], hv4[4];
        float tmax = KEY_OFF;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float ah = s[hh][r];
          if (clip) ah = __builtin_amdgcn_fmed3f(ah, a.clip_lo, a.clip_hi);
          const float hv = ah + f4get(e4[hh], r);
          hv4[r] = hv;                                          // H_hat: post-clip, pre-mask (egt_layers.py:85-86)
          float add = kav[r];
          if (V == 2) {
            const int m = PR_CLAMP(m0 + 4 * q + r);
            const uint32_t gi = (uint32_t)((((size_t)b * N + lq) * N + m) * AH + h);
            add += ((egt_hash32(gi, a.s0, a.s1) >> 8) < a.rm_thr) ? -EGT_NEG : 0.0f;
          }
          x[r] = hv + add;
          tmax = fmaxf(tmax, x[r]);
          const float tg = V == 1 ? exp2_fast(fmaf(f4get(g4[hh], r), -L2E, kgv[r])) : exp2_fast((f4get(g4[hh], r) + add) * -L2E);
          pa_[hh][r] = __builtin_amdgcn_rcpf(1.0f + tg);
        }
        *reinterpret_cast<float4*>(Hp + (it & 1) * (AH * PT_PL) + hh * PT_PL + po4) = make_float4(hv4[0], hv4[1], hv4[2], hv4[3]);
        tmax = pair_max_q(tmax);
        const float mnew = fmaxf(mrun[hh], tmax);
        const float alpha = exp2_fast((mrun[hh] - mnew) * L2E);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pexp = exp2_fast((x[r] - mnew) * L2E);
          psum += pexp;
          pa_[hh][r] *= pexp;
        }
        psum = pair_sum_q(psum);
        lrun[hh] = fmaf(lrun[hh], alpha, psum);
        mrun[hh] = mnew;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          v4f o = oacc[hh][kt];
          o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
          oacc[hh][kt] = o;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      if (more) vm_wait<2 * KT>(); else vm_wait<0>();   // V^T(it) landed (only the K(it+1) pieces just issued are younger)
      float4 vc[2][KT];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt) vc[hh][Tt] = *reinterpret_cast<const float4*>(opl + AH * HS + hh * HS + Tt * 256);
      // ---- O^T[k][l] += sum_m V^T[k][m] P^T[m][l] ----
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) oacc[hh][kt] = MFMA(f4get(vc[hh][kt], r), pa_[hh][r], oacc[hh][kt]);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      if (more) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dma_tile(vdst, VT + (size_t)mt_of(it + 1) * 16 * D); }
      pair_barrier();           // barrier(it+1)
      STAMP(3);
    }
    // ---- finalize: O[l][k] / l_run into the K stage as [row][k][8 heads]; row statistics for the backward ----
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * w + hh, l = l0 + ll;
      const float inv = 1.0f / lrun[hh];
      float* vs = sm + (ll * D + 4 * q) * AH + h;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vs[(16 * kt + r) * AH] = oacc[hh][kt][r] * inv;
      if (l < N && q == 0)
        *reinterpret_cast<float4*>(a.rowstats + (((size_t)b * N + l) * AH + h) * 4) = make_float4(mrun[hh], lrun[hh], 0.f, 0.f);
    }
    STAMP_OUT(0);
  }
  // V_att[l][k * 8 + h]: 16 rows x 512 channels, consecutive threads consecutive 16-byte pieces
  pair_barrier();
  for (int p = tid; p < 16 * DH / 4; p += 64 * PR_WAVES) {
    const int row = p / (DH / 4);
    if (l0 + row < N)
      *reinterpret_cast<float4*>(a.v_att + ((size_t)b * N + l0) * DH + (size_t)p * 4) = *reinterpret_cast<const float4*>(sm + (size_t)p * 4);
  }
}

// ================================================================= backward =====
// Launches: k_attn_pack (operand arrays of dV_att, row constants m, 1/l, delta), k_pair_bwd (below), k_attn_mfma_bwd_q (dQ from
// the dA tiles), then the deterministic reduction of the per-workgroup parameter-gradient partials (k_edge_reduce-style).
//
// k_pair_bwd: workgroup = (graph, 16 keys, all 8 heads) walking the query tiles (dK / dV accumulate in registers).
//   attention wave w owns heads w and w + 4 and works on ONE of them per half-trip: the Q / dO operand stage of a half-trip holds
//   four heads (32 KB), two stages -> the tiles of the next half-trip land while this one computes, and a wave keeps only one
//   head's transient state beside the K / V fragments and dK / dV accumulators of its two heads (the transposed operand
//   forms are read from the same tiles just in time);
//   edge wave j owns query rows j, j + 4, j + 8, j + 12 of the tile (the row enters the plane offsets as a compile-time chunk XOR + the dword j): PRE(it+1) = e, de' tile -> LayerNorm, G / E projection, dH_ext = de'.Wr^T
//   -> planes; POST(it-1) = dG / dE / H_hat planes (written by the attention waves IN PLACE of G / E / dH_ext) ->
//   d ehat = Wp.dGE, LayerNorm backward, de = de' + .. stored; weight gradients T += ehat^T.dGE, R += de'^T.[H_hat | 1] on the
//   matrix core (accumulators live for the whole workgroup -> deterministic partials).  ehat and de' of a tile stay in the
//   registers that loaded them from PRE to POST (two register sets, one per trip parity), refilled right after POST.
//   ONE barrier per trip: an attention wave stages the Q / dO tiles of its own heads itself (it is their only reader: DMA of the next
//   trip's tile right after its last read of a stage, counted vmcnt before its next read; no cross-wave hand-off), and inside an edge
//   wave POST(it-1) precedes PRE(it+1) in program order (they share a plane set, and a wave touches only its own rows of it)
//   trip it:  attention: half A (head w), half B (head w + 4)       edge: POST(it-1) + e, de' requests of it+1; PRE(it+1); row constants of it+1
template <int D, int DE, int V, bool FULL>
__global__ void __launch_bounds__(64 * PR_WAVES, 2) k_pair_bwd(AttnMfmaArgs a, PairArgs pa) {
  constexpr int KT = D / 16, DH = D * AH, T = PairGeo<DE>::T, HS = KT * 256, TSZ = AH * PT_PL;
  constexpr int PSZ1 = DE * 16 + 16, PSZ2 = AH * DE + DE;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, NP = a.NP, mtiles = NP / 16;
  float* Ops = sm;                         // [2 stages][4 head slots][Q | dO][HS]
  float* Pl = Ops + 2 * 4 * 2 * HS;        // [2 sets][E | G | X][8][PT_PL]
  float* statL = Pl + 2 * 3 * TSZ;         // [2 stages][8 heads][16 rows][4]
  float* scr = statL + 2 * AH * 64;        // [4 edge waves][ehat tile | de' tile][16][DE]
  float* wtab = scr + 4 * 2 * 16 * DE;     // [wA | wB | wR][T][64 lanes][4] | bias [64][4]: the edge waves' lane-constant MFMA A operands (28 registers
                                           // otherwise); then [4 waves][2 sets][4 rows][16] 1 / sigma of the held tiles
  const int wg = egt_xcd_remap(blockIdx.x, gridDim.x);
  const int b = __builtin_amdgcn_readfirstlane(wg / mtiles), mt0 = __builtin_amdgcn_readfirstlane(wg % mtiles), m0 = mt0 * 16;
  const size_t arr = (size_t)a.B * AH * NP * D;
  const bool clip = (a.flags & EGT_F_CLIP) != 0;
  STAMP_DECL;

  if (wv >= 4) {
    // --------------------------------------------------------------------- edge waves ----
    const int j = wv - 4, p = lane & 15, q = lane >> 4;
#if PAIR_EDGE_PRIO
    __builtin_amdgcn_s_setprio(PAIR_EDGE_PRIO);
#endif
    {   // the prepared table (k_pair_prep) -> LDS: every edge wave writes the same values (a benign race) and reads only after its own writes
#pragma unroll
      for (int i = 0; i < 3 * T + 1; ++i)
        *reinterpret_cast<float4*>(wtab + (i * 64 + lane) * 4) = *reinterpret_cast<const float4*>(pa.wprep + (i * 64 + lane) * 4);
      lds_sync_();
    }
    auto wget = [&](int which, int t) __attribute__((always_inline)) { return *reinterpret_cast<const float4*>(wtab + ((which * T + t) * 64 + lane) * 4); };
    v4f accT[T], accW[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accW[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    float4 accS = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* st2 = a.stats2 + ((size_t)b * AH + j + 4 * ((lane >> 4) & 1)) * NP * 4;   // lanes 0..15: head j, 16..31: head j + 4 (upper lanes repeat)
    auto stat_load = [&](int ltile) __attribute__((always_inline)) {
      return *reinterpret_cast<const float4*>(st2 + (size_t)(ltile * 16 + (lane & 15)) * 4);
    };
    auto stat_put = [&](float4 v, int stage) __attribute__((always_inline)) {
      if (lane < 32) *reinterpret_cast<float4*>(statL + ((stage * AH + j + 4 * (lane >> 4)) * 16 + (lane & 15)) * 4) = v;
    };
    const int mkey = m0 + p;
    const bool keyok = FULL || mkey < N;
    // Rows of the wave: in row group r (rows 4 r .. 4 r + 3 of the tile) the lanes of key p work on row 4 r + sk, sk = (j + p) & 3 -- the
    // four waves cover the four rows of every key (a Latin square).  With ONE row per wave-instruction (the first version: row 4 r + j)
    // every plane access of the wave has the same (row & 3): its 32 lanes reach 8 of the 32 banks, a 4-way conflict on the 20 plane
    // instructions per row (SQ_LDS_BANK_CONFLICT was 47 % of SQ_LDS_IDX_ACTIVE); skewed, a lane group's 32 addresses fall on 32 banks.
    const int sk = (j + p) & 3;
    const uint32_t rowB = (uint32_t)N * DE * 4;                          // bytes of one query row of e
    const uint32_t keyb = ((uint32_t)PR_CLAMP(mkey) * DE + 4 * q) * 4;   // lane offset (bytes, 32 bit) inside a row
    const uint32_t keybs = keyb + sk * rowB;                             // ... plus the lane's row inside a row group: the ONE lane offset of every e / de' / de access (FULL)
    const size_t gb = (size_t)b * N * N * DE;
    // address of the lane's pair in row group r of query tile ltile: wave-uniform base (scalar registers) + 32-bit lane offset
    auto rowaddr = [&](const float* base, int ltile, int r, uint32_t& lo, int kind = 1) __attribute__((always_inline)) {
      if (PAIR_BWD_ABL & kind) ltile = 0;   // (timing ablation: every trip on the tile of trip 0 -- cache-resident)
      if (FULL) { lo = keybs; return reinterpret_cast<const char*>(base) + (gb + (size_t)(16 * ltile + 4 * r) * N * DE) * sizeof(float); }
      lo = keyb + (uint32_t)min(16 * ltile + 4 * r + sk, N - 1) * rowB;   // (N <= 2048: < 2^30)
      return reinterpret_cast<const char*>(base) + gb * sizeof(float);
    };
    struct HSet { float4 x[4][T], df[4][T]; };
    float* rsd = wtab + 3 * T * 64 * 4 + 64 * 4 + j * (2 * 4 * 16);   // 1 / sigma of the held tiles: [set][row][pair] (8 registers otherwise)
    auto raw_load = [&](HSet& s, int r, int ltile) __attribute__((always_inline)) {
      uint32_t lo;
      const char* er = rowaddr(pa.e, ltile, r, lo);
      const char* dr = rowaddr(pa.d_e_out, ltile, r, lo);
#pragma unroll
      for (int t = 0; t < T; ++t) s.x[r][t] = pr_ld4<(PAIR_BWD_NT & 1) != 0>(er + lo + 64 * t);
#pragma unroll
      for (int t = 0; t < T; ++t) s.df[r][t] = pr_ld4<(PAIR_BWD_NT & 2) != 0>(dr + lo + 64 * t);
    };
    // de' of a held tile is NOT kept in registers from PRE to POST (two sets x four rows x 8 registers that the allocator does not have:
    // spill reloads wait on vmcnt(0), i.e. drain the wave's whole prefetch queue): POST reads it again -- a tile this CU streamed one
    // trip ago, served by L2 / the memory-side cache
    char* dumpl = reinterpret_cast<char*>(pa.dump) + lane * 16;
    auto df_load = [&](HSet& s, int r, int ltile) __attribute__((always_inline)) {
      uint32_t lo;
      const char* dr = rowaddr(pa.d_e_out, ltile, r, lo, 4);
#pragma unroll
      for (int t = 0; t < T; ++t) s.df[r][t] = pr_ld4<(PAIR_BWD_NT & 4) != 0>(dr + lo + 64 * t);
    };
    float* xs = scr + j * (2 * 16 * DE);   // [ehat tile | de' tile]: layout below
    // plane offsets (ptT_off) of the lane's rows 4 r + sk: the row enters through r (the chunk XOR, compile time) and sk (the dword):
    //   ptT_off(4 r + sk, col) = (col << 4) + (((r ^ A[col >> 2]) & 3) << 2) + sk,  A = (0, 2, 3, 1)
    // own pair (col = p): (p << 4) + sk + 4 ((r ^ A[p >> 2]) & 3); pairs on the contraction axis (col = 4 s + q, row 4 r + ((j + q) & 3)):
    // (q << 4) + ((j + q) & 3) + a compile-time constant -> ONE address register for the sixteen transposed reads of a row
    const int offp0 = (p << 4) + sk, ap4 = chunk_xor(p) << 2;
    const int q16 = (q << 4) + ((j + q) & 3);   // pair 4 s + q on the contraction axis: its row is 4 r + ((j + q) & 3)
    // scratch tiles: [channel block t][pair][16 channels], the 16-byte chunk index XORed by (pair >> 1) & 3
    //   own row (pair p), chunk q of block t:     256 t + 16 p + 4 (q ^ ((p >> 1) & 3))                   (b128 stores: 8-lane groups on 32 banks)
    //   element (pair 4 s + q, channel 16 t + p): 256 t + 64 s + [16 q + 4 ((p >> 2) ^ (q >> 1)) + (p & 3)] ^ 8 (s & 1)
    //     (b32 reads: the pairs q, q + 1 of a 32-lane group are 16 floats apart -> the two halves of the banks.  The first layout,
    //      [pair][DE] with the chunk XORed by pair & 7, put both on the same 16 banks: 2-way on the 16 reads of a row)
    const int xw0 = 16 * p + ((q ^ ((p >> 1) & 3)) << 2);
    const int xrd0 = 16 * q + (((p >> 2) ^ (q >> 1)) << 2) + (p & 3), xrd1 = xrd0 ^ 8;   // s even / odd
    // PRE: tile row r of query tile `ltile` from the raw values in set s -> planes of set `ps`; ehat / de' stay in s
    auto pre = [&](HSet& s, int set, int r, int ltile, float* ps) __attribute__((always_inline)) {
      const bool valid = FULL || (keyok && (16 * ltile + 4 * r + sk) < N);
      const float rstd_ = pair_ln<T>(s.x[r], pa.ln_eps);
      if (q == 0) rsd[(set * 4 + r) * 16 + p] = rstd_;   // (the exec-masked block also keeps hipcc from clustering the four rows' LayerNorm sums in front of the wait for the LAST row's loads)
      const float4 b4 = wget(3, 0);
      v4f acc = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 wa = wget(0, t);
        acc = MFMA(wa.x, s.x[r][t].x, acc);
        acc = MFMA(wa.y, s.x[r][t].y, acc);
        acc = MFMA(wa.z, s.x[r][t].z, acc);
        acc = MFMA(wa.w, s.x[r][t].w, acc);
      }
      const int off = offp0 + (ap4 ^ (4 * r));
      float* pl = ps + (q < 2 ? TSZ : 0) + (4 * (q & 1)) * PT_PL + off;   // q < 2: gates (slot 1), q >= 2: edge bias (slot 0)
      pl[0] = acc[0]; pl[PT_PL] = acc[1]; pl[2 * PT_PL] = acc[2]; pl[3 * PT_PL] = acc[3];
      if (!valid) {   // de' of a pair outside the graph must not reach dH (it is added to it) nor the weight gradients
#pragma unroll
        for (int t = 0; t < T; ++t) s.df[r][t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      v4f ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 wr = wget(2, t);
        ax = MFMA(wr.x, s.df[r][t].x, ax);
        ax = MFMA(wr.y, s.df[r][t].y, ax);
        ax = MFMA(wr.z, s.df[r][t].z, ax);
        ax = MFMA(wr.w, s.df[r][t].w, ax);
      }
      if (q < 2) {   // rows 4q + i of the product = heads 4q + i
        float* px = ps + 2 * TSZ + (4 * q) * PT_PL + off;
        px[0] = ax[0]; px[PT_PL] = ax[1]; px[2 * PT_PL] = ax[2]; px[3 * PT_PL] = ax[3];
      }
    };
    // POST: tile row r of query tile `ltile`: the planes of set `ps` now hold dE | dG | H_hat
    auto post = [&](HSet& s, int set, int r, int ltile, const float* ps) __attribute__((always_inline)) {
      const int l = 16 * ltile + 4 * r + sk;
      const bool valid = FULL || (keyok && l < N);
      if (!valid) {
#pragma unroll
        for (int t = 0; t < T; ++t) s.df[r][t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const int off = offp0 + (ap4 ^ (4 * r));
      const float* pd = ps + (q < 2 ? TSZ : 0) + (4 * (q & 1)) * PT_PL + off;
      const float dp0 = pd[0], dp1 = pd[PT_PL], dp2 = pd[2 * PT_PL], dp3 = pd[3 * PT_PL];   // d(pre-activation) of outputs 4q..4q+3 of pair p
      float4 dx[T];
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const float4 wb = wget(1, t);
        acc = MFMA(wb.x, dp0, acc);
        acc = MFMA(wb.y, dp1, acc);
        acc = MFMA(wb.z, dp2, acc);
        acc = MFMA(wb.w, dp3, acc);
        dx[t] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        m1 += (acc[0] + acc[1]) + (acc[2] + acc[3]);
        m2 = fmaf(acc[0], s.x[r][t].x, m2); m2 = fmaf(acc[1], s.x[r][t].y, m2);
        m2 = fmaf(acc[2], s.x[r][t].z, m2); m2 = fmaf(acc[3], s.x[r][t].w, m2);
      }
      m1 = pair_sum_q(m1) * (1.0f / DE);
      m2 = pair_sum_q(m2) * (1.0f / DE);
      const float rstd = rsd[(set * 4 + r) * 16 + p];
      uint32_t lo;
      char* orow_ = const_cast<char*>(rowaddr(pa.d_e, ltile, r, lo, 2));
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 xh = s.x[r][t], d0 = s.df[r][t];
        const float4 o = make_float4(d0.x + rstd * (dx[t].x - m1 - xh.x * m2), d0.y + rstd * (dx[t].y - m1 - xh.y * m2),
                                     d0.z + rstd * (dx[t].z - m1 - xh.z * m2), d0.w + rstd * (dx[t].w - m1 - xh.w * m2));
        *reinterpret_cast<float4*>(valid ? orow_ + lo + 64 * t : dumpl) = o;
      }
      // weight gradients: both operands need the pairs on the contraction axis: ehat / de' through the wave's scratch tiles
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int ch = xw0 + 256 * t;
        *reinterpret_cast<float4*>(xs + ch) = s.x[r][t];
        *reinterpret_cast<float4*>(xs + 16 * DE + ch) = s.df[r][t];
      }
      accS.x += dp0; accS.y += dp1; accS.z += dp2; accS.w += dp3;
      float aD[4], hA[4];
      {
        const int i = p;   // A row: output channel i of [gates | edge bias] / head i of H_hat
        const float* pi = ps + (i < 8 ? TSZ : 0) + (i & 7) * PT_PL + q16;
        const float* ph = ps + 2 * TSZ + (i & 7) * PT_PL + q16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int o2 = 64 * s4 + (((r ^ ((0x78 >> (2 * s4)) & 3)) & 3) << 2);   // compile-time: ptT_off(4 r + sk', 4 s + q) - q16
          aD[s4] = pi[o2];
          float hv = ph[o2];
          asm volatile("" : "+v"(hv));   // (an unconditional read: as an operand of the select alone hipcc sinks it into an exec-masked block)
          hA[s4] = i < 8 ? hv : (i == 8 ? 1.0f : 0.f);   // row 8 = ones: its product row is the bias gradient sum of de'
        }
      }
      lds_order_();
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int idx = ((s4 & 1) ? xrd1 : xrd0) + 64 * s4 + 256 * t;       // [pair 4 s + q][channel 16 t + p]
          accT[t] = MFMA(aD[s4], xs[idx], accT[t]);
          accW[t] = MFMA(hA[s4], xs[16 * DE + idx], accW[t]);
        }
      lds_order_();
    };

    HSet H0, H1;
#pragma unroll
    for (int r = 0; r < 4; ++r) raw_load(H0, r, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) raw_load(H1, r, min(1, mtiles - 1));
    stat_put(stat_load(0), 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) pre(H0, 0, r, 0, Pl);
    pair_barrier();             // barrier(0)
    // trip it: `prev` = set of tile it-1 (POST, then refilled with tile it+1), which PRE then turns into the held set of it+1
    // (no LDS-DMA in these waves: every vector-memory operation is visible to the compiler, whose own vmcnt waits before the first use of
    //  a loaded register are then exact.  First and last trip peeled: the steady-state trip is branch-free, its request order pinned)
    // trip it works on the set of tile it-1 (`prev` = H[(it + 1) & 1], planes Pl + set * 3 TSZ): POST(it-1), refilled row by row with the raw
    // values of tile it+1, which PRE then turns into the held set of it+1
    auto steady = [&](int it, HSet& prev, int set) __attribute__((always_inline)) {
      float* pprev = Pl + set * 3 * TSZ;
      STAMP(8);
#pragma unroll
      for (int r = 0; r < 4; ++r) df_load(prev, r, it - 1);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        post(prev, set, r, it - 1, pprev);
        raw_load(prev, r, it + 1);
      }
      STAMP(1);
      const float4 stn = stat_load(it + 1);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
#pragma unroll
      for (int r = 0; r < 4; ++r) pre(prev, set, r, it + 1, pprev);
      STAMP(5);
      stat_put(stn, (it + 1) & 1);
      STAMP(6);
      pair_barrier();           // barrier(it+1)
      STAMP(7);
    };
    {   // trip 0: nothing to POST yet; tile 1's raw values were requested in the prologue
      if (mtiles > 1) {
        const float4 stn = stat_load(1);
#pragma unroll
        for (int r = 0; r < 4; ++r) pre(H1, 1, r, 1, Pl + 3 * TSZ);
        stat_put(stn, 1);
      }
      pair_barrier();
    }
    int it = 1;
    for (; it + 1 <= mtiles - 2; it += 2) { steady(it, H0, 0); steady(it + 1, H1, 1); }
    if (it <= mtiles - 2) { steady(it, H0, 0); ++it; }
    if (mtiles > 1) {          // trip mtiles-1: POST(mtiles-2), nothing left to prepare
      if (it & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) df_load(H0, r, it - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) post(H0, 0, r, it - 1, Pl);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) df_load(H1, r, it - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) post(H1, 1, r, it - 1, Pl + 3 * TSZ);
      }
      pair_barrier();
    }
    // POST of the last tile
    {
      const float* pl = Pl + ((mtiles - 1) & 1) * 3 * TSZ;
      if ((mtiles - 1) & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) df_load(H1, r, mtiles - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) post(H1, 1, r, mtiles - 1, pl);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) df_load(H0, r, mtiles - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) post(H0, 0, r, mtiles - 1, pl);
      }
    }
    STAMP(9);
    STAMP_OUT(1);
    // partials of this wave -> the (now idle) other plane set: [That[DE][16] | s[16] | dWr[8][DE] | dbr[DE]]
    float* red = Pl + (mtiles & 1) * 3 * TSZ + j * (PSZ1 + PSZ2);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[(16 * t + p) * 16 + 4 * q + r] = accT[t][r];                 // accT[t][r] = That[c = 16 t + p][o = 4 q + r]
        const int hrow = 4 * q + r;                                       // accW[t][r] = R[row 4 q + r][c = 16 t + p]
        if (hrow < 8) red[PSZ1 + hrow * DE + 16 * t + p] = accW[t][r];
        else if (hrow == 8) red[PSZ1 + AH * DE + 16 * t + p] = accW[t][r];
      }
    {
      const float s0 = row_sum16(accS.x), s1 = row_sum16(accS.y), s2 = row_sum16(accS.z), s3 = row_sum16(accS.w);
      if (p == 0) { red[DE * 16 + 4 * q] = s0; red[DE * 16 + 4 * q + 1] = s1; red[DE * 16 + 4 * q + 2] = s2; red[DE * 16 + 4 * q + 3] = s3; }
    }
  } else {
    // ---------------------------------------------------------------- attention waves ----
    const int w = wv, mm = lane & 15, q = lane >> 4;
    const uint32_t ooff = mm * 16 + 4 * q;
    float4 Kr[2][KT], Vr[2][KT];
    float* dAb[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int h = w + 4 * hf;
      const size_t hb = ((size_t)b * AH + h) * NP * D;
      const float* Kh = a.pk + PK_KH * arr + hb;
      const float* Vh = a.pk + PK_VH * arr + hb;
      dAb[hf] = a.ws_dA + (((size_t)b * AH + h) * mtiles * mtiles + (size_t)mt0) * 256;   // + ltile * mtiles * 256
#pragma unroll
      for (int Tt = 0; Tt < KT; ++Tt) {
        Kr[hf][Tt] = *reinterpret_cast<const float4*>(Kh + (size_t)mt0 * 16 * D + 256 * Tt + ooff);
        Vr[hf][Tt] = *reinterpret_cast<const float4*>(Vh + (size_t)mt0 * 16 * D + 256 * Tt + ooff);
      }
    }
    float kadd = 0.f;
    {
      const int m = m0 + mm;
      if (m >= N) kadd = KEY_OFF;
      else if (a.km && a.km[(size_t)b * N + m] == 0) kadd = -EGT_NEG;
    }
    const float kaddg = (m0 + mm) < N ? -kadd * L2E : 3.0e38f;
    const int mc = PR_CLAMP(m0 + mm);
    v4f dKacc[2][KT], dVacc[2][KT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { dKacc[hf][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; dVacc[hf][kt] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    const int po4 = ptT_off(4 * q, mm);   // the lane's query rows 4q..4q+3 of key mm: one 16-byte access per plane
    // operand tiles of head slot w (Q, then dO): row form: lane (row mm, chunk q) b128; transposed: lane (channel mm, q): rows 4q + r.
    // Tile row rho lives in LDS row rho ^ ((rho >> 2) & 1) (rows 4-7 and 12-15 swapped in pairs -- the DMA's lane -> source mapping is
    // free): the transposed b32 reads of a 32-lane group touch rows r and 4 + r, which 16-float rows put on the same 16 banks (2-way on
    // the 32 reads of a half-trip); swapped, row 4 + r sits in an odd-even exchanged line and the group covers 32 banks.  Step r of a lane
    // still reads tile row 4q + r: from LDS row 4q + (r ^ (q & 1)) = 4q + r +- (q & 1) (two base addresses, no arithmetic in the loop).
    const float* oprow = Ops + w * (2 * HS) + (mm ^ ((mm >> 2) & 1)) * 16 + ((q ^ chunk_xor(mm)) << 2);
    const float* optr = Ops + w * (2 * HS) + (4 * q) * 16 + (((mm >> 2) ^ chunk_xor(4 * q)) << 2) + (mm & 3);
    const float* optr_e = optr + (q & 1) * 16;   // steps 0, 2: LDS row 4q + r + (q & 1)
    const float* optr_o = optr - (q & 1) * 16;   // steps 1, 3: LDS row 4q + r - (q & 1)
    // the wave's own operand stages: stage `half` = (Q, dO) tiles of head w + 4 half; four 1 KB pieces per tile
    const unsigned doff = dma_lane_off(lane ^ (((lane >> 4) & 1) << 2));   // LDS slot of row sigma <- source row sigma ^ ((sigma >> 2) & 1) (same chunk XOR: it depends on sigma >> 2 only)
    const float* Qh = uni_ptr(a.pk + PK_QH * arr + ((size_t)b * AH + w) * NP * D);
    const float* Oh = uni_ptr(a.pk + PK_OH * arr + ((size_t)b * AH + w) * NP * D);
    const unsigned odst = lds_addr(Ops) + w * (2 * HS * 4);
    const unsigned doff1 = doff + 1024, doff2 = doff + 2048, doff3 = doff + 3072;
    auto dma_half = [&](int ltile, int half) __attribute__((always_inline)) {
      const size_t ho = (size_t)half * 4 * NP * D + (size_t)ltile * 16 * D;
      const unsigned dst = odst + half * (4 * 2 * HS * 4);
      if constexpr (KT == 4) {
        dma_pieces4(dst, Qh + ho, doff, doff1, doff2, doff3);
        dma_pieces4(dst + KT * 1024, Oh + ho, doff, doff1, doff2, doff3);
      } else {
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt) {
          dma_piece(dst + Tt * 1024, Qh + ho + Tt * 256, doff);
          dma_piece(dst + (KT + Tt) * 1024, Oh + ho + Tt * 256, doff);
        }
      }
    };
    dma_half(0, 0);
    dma_half(0, 1);
    vm_wait<0>();              // (K / V fragments in registers, both stages of query tile 0 landed)
    pair_barrier();             // barrier(0)
    STAMP(8);
    for (int l0 = 0, it = 0; it < mtiles; l0 += 16, ++it) {
      float* ps = Pl + (it & 1) * 3 * TSZ;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int h = w + 4 * hf;
        const float* opr = oprow + hf * (4 * 2 * HS);
        const float* opt_e = optr_e + hf * (4 * 2 * HS);
        const float* opt_o = optr_o + hf * (4 * 2 * HS);
        // this stage landed: requested one half-trip ago; only the other stage's pieces (and the dA store) are younger -- except in the
        // second half of the last trip, where nothing was requested after it
        if (it + 1 < mtiles || hf == 0) { if (it > 0 || hf > 0) vm_wait<2 * KT>(); } else vm_wait<0>();
        float4 qa[KT], oa[KT];
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt) {
          qa[Tt] = *reinterpret_cast<const float4*>(opr + Tt * 256);
          oa[Tt] = *reinterpret_cast<const float4*>(opr + HS + Tt * 256);
        }
        const float4 e4 = *reinterpret_cast<const float4*>(ps + h * PT_PL + po4);
        const float4 g4 = *reinterpret_cast<const float4*>(ps + TSZ + h * PT_PL + po4);
        const float4 x4 = *reinterpret_cast<const float4*>(ps + 2 * TSZ + h * PT_PL + po4);
        float4 stc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) stc[r] = *reinterpret_cast<const float4*>(statL + (((it & 1) * AH + h) * 16 + 4 * q + r) * 4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- S[l][m] = sum_k (d^-1/2 Q)[l][k] K[m][k] ; dP[l][m] = sum_k dO[l][k] V[m][k] ----
        v4f s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int Tt = 0; Tt < KT; ++Tt)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s = MFMA(f4get(qa[Tt], u), f4get(Kr[hf][Tt], u), s);
            dp = MFMA(f4get(oa[Tt], u), f4get(Vr[hf][Tt], u), dp);
          }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(0);
        float at[4], da[4], dE4[4], dG4[4], hh4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float araw = s[r];
          float ah = araw;
          if (clip) ah = __builtin_amdgcn_fmed3f(araw, a.clip_lo, a.clip_hi);
          float add = kadd;
          if (V == 2) {
            const int lq = PR_CLAMP(l0 + 4 * q + r);
            const uint32_t gi = (uint32_t)((((size_t)b * N + lq) * N + mc) * AH + h);
            add += ((egt_hash32(gi, a.s0, a.s1) >> 8) < a.rm_thr) ? -EGT_NEG : 0.0f;
          }
          const float hv = ah + f4get(e4, r);
          const float xx = hv + add;
          const float S = exp2_fast((xx - stc[r].x) * L2E) * stc[r].y;   // rows past N: 1/l = 0; keys past N: exp2(-inf) = 0
          const float tg = V == 1 ? exp2_fast(fmaf(f4get(g4, r), -L2E, kaddg)) : exp2_fast((f4get(g4, r) + add) * -L2E);
          const float g = __builtin_amdgcn_rcpf(1.0f + tg);
          const float dAt_ = dp[r];
          const float Sg = S * g;
          const float dH = fmaf(S, fmaf(dAt_, g, -stc[r].z), f4get(x4, r));
          at[r] = Sg;
          da[r] = (clip && ah != araw) ? 0.f : dH;
          dE4[r] = dH;
          dG4[r] = dAt_ * Sg * (1.0f - g);
          hh4[r] = hv;
        }
        *reinterpret_cast<float4*>(ps + h * PT_PL + po4) = make_float4(dE4[0], dE4[1], dE4[2], dE4[3]);
        *reinterpret_cast<float4*>(ps + TSZ + h * PT_PL + po4) = make_float4(dG4[0], dG4[1], dG4[2], dG4[3]);
        *reinterpret_cast<float4*>(ps + 2 * TSZ + h * PT_PL + po4) = make_float4(hh4[0], hh4[1], hh4[2], hh4[3]);
        *reinterpret_cast<float4*>(dAb[hf] + (size_t)it * mtiles * 256 + ooff) = make_float4(da[0], da[1], da[2], da[3]);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        // ---- dV^T[k][m] += sum_l dO[l][k] A[l][m] ; dK^T[k][m] += sum_l (d^-1/2 Q)[l][k] dA[l][m] ----
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* opt = (r & 1) ? opt_o : opt_e;
            const float qq = opt[kt * 256 + r * 16];
            const float oo = opt[HS + kt * 256 + r * 16];
            dVacc[hf][kt] = MFMA(oo, at[r], dVacc[hf][kt]);
            dKacc[hf][kt] = MFMA(qq, da[r], dKacc[hf][kt]);
          }
        STAMP(2);
        // the stage is free (this wave was its only reader): the same head's tiles of the next query tile
        if (it + 1 < mtiles) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dma_half(it + 1, hf); }
        if (hf == 1) pair_barrier();   // barrier(it+1)
        STAMP(3);
      }
    }
    STAMP_OUT(1);
    // dK / dV into the (now idle) operand stages as [key][k][8 heads]: dK in the first 32 KB, dV in the second
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float* ks = sm + (mm * D + 4 * q) * AH + w + 4 * hf;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ks[(16 * kt + r) * AH] = dKacc[hf][kt][r];
          ks[16 * DH + (16 * kt + r) * AH] = dVacc[hf][kt][r];
        }
    }
  }
  pair_barrier();
  // dK / dV rows of d_qkv: 16 keys x 512 channels each, 16-byte pieces
  for (int pz = tid; pz < 16 * DH / 4; pz += 64 * PR_WAVES) {
    const int row = pz / (DH / 4), c = (pz % (DH / 4)) * 4;
    if (m0 + row < N) {
      float* o = a.d_qkv + ((size_t)b * N + m0 + row) * 3 * DH + c;
      *reinterpret_cast<float4*>(o + DH) = *reinterpret_cast<const float4*>(sm + (size_t)pz * 4);
      *reinterpret_cast<float4*>(o + 2 * DH) = *reinterpret_cast<const float4*>(sm + 16 * DH + (size_t)pz * 4);
    }
  }
  // parameter-gradient partials of the workgroup: the four edge waves' images summed in a fixed order
  {
    const float* red = Pl + (mtiles & 1) * 3 * TSZ;
    for (int i = tid; i < PSZ1 + PSZ2; i += 64 * PR_WAVES) {
      const float v = (red[i] + red[(PSZ1 + PSZ2) + i]) + (red[2 * (PSZ1 + PSZ2) + i] + red[3 * (PSZ1 + PSZ2) + i]);
      if (i < PSZ1) pa.part_proj[(size_t)wg * PSZ1 + i] = v;
      else pa.part_upd[(size_t)wg * PSZ2 + (i - PSZ1)] = v;
    }
  }
}
