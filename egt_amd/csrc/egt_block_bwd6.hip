// k_block_bwd_v6<DE> -- the fused backward of the attention block with every 16-pair tile worked on by a PAIR of waves.
//
// Why: k_block_bwd_v5 is bound by the latency of ONE wave's dependent chain per tile (LayerNorm -> MFMA chain -> exp /
// sigmoid -> LDS transposition -> MFMA chains -> LayerNorm backward) at two waves per SIMD (222 VGPRs): the matrix pipe is
// 44 % busy, the waves issue 26 % of their cycles.  Splitting the chain of a tile over two waves that run
// SIDE BY SIDE on the same LDS tiles halves the registers each needs (<= 128: four waves per SIMD, still two workgroups per
// CU because both waves of a pair share the tile buffers) and shortens the critical path of a tile from
// P1+P2+P3+dQ/dK/dV+P4+P5 to max(P1, P2+dots) + P3 + max(P4, dQ/dK/dV+P5).
//
// Workgroup = (graph, 16 query rows), 8 waves = 4 key-tile lanes x 2 roles; per query row three stages separated by
// workgroup barriers (raw s_barrier + lgkmcnt(0): the LDS-DMA of the next e tile stays in flight across them):
//   role A (waves 0-3)                                        role B (waves 4-7)
//   S1  e(l) landed (LDS-DMA, issued a row ahead) -> P1:      de'(l) (registers, loaded a row ahead) -> LDS tile -> P2:
//       norm_edge, [G|E] projections (16 MFMA); xhat          dH_ext = de'.Wr^T (16 MFMA); the QK^T and dV_att.V dots of
//       written back into the tile                            the lane's pair (K, V live in B's registers) -> LDS
//   S2  P3: logits, masks, softmax / gate backward ->         request de'(l+1)
//       dGE, H_hat, (dA, A~) -> LDS
//   S3  P4: T += xhat^T.dGE, R += de'^T.[H_hat|1] (32 MFMA)   dK, dV += ..., dQ partial -> HBM; P5: d xhat = Wp.dGE (16 MFMA),
//                                                             LayerNorm backward, de = de' + ... -> HBM
// Same arithmetic, operand order and partial-buffer contract as v5 (dqp / dkvp / epart, node-side prologue), so the parity
// suite applies unchanged.  fp32 edge tensors, no mask tensors, N and De multiples of 16.
// LDS per pair: 2 e buffers + 1 de' buffer + hand-off area (dGE [16][16] | [H_hat|1|rstd] [16][12] | (dA, A~) per lane; the
// B -> A hand-off of stage 1 aliases the first two); the dH_ext weight slab is stored without its zero rows.  80 960 B at
// De = 64: two workgroups per CU.
#include "egt_common.h"

#include <stdlib.h>
#include <vector>

#include "egt_block.h"
#include "egt_tile.h"
#include "egt_block_dev.h"
#include "egt_dma.h"

__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// the lane index, recomputed where it is needed (v_mbcnt) and opaque to loop-invariant code motion: at 128 registers per wave
// hipcc otherwise keeps a dozen lane-derived LDS addresses alive across the row loop and spills some of them -- and a scratch
// reload inside role A's loop carries an s_waitcnt vmcnt(0), which waits for the LDS-DMA of the next tile
__device__ __forceinline__ int lane_now() {
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}

#ifdef EGT_BWD_TIMING
#define T6STAMP(i) do { const unsigned tn__ = (unsigned)__builtin_amdgcn_s_memtime(); tacc[i] += tn__ - tlast; tlast = tn__; } while (0)
#else
#define T6STAMP(i) do {} while (0)
#endif

template <int DE>
struct Geo6 {
  using G = Geo<DE>;
  static constexpr int XCH = 704;                          // sc1 [256] | sc2 [192] | (dA, A~) [256]
  static constexpr int PW = 2 * G::TILE_FLOATS + XCH;      // per key-tile lane: e buffer (odd rows) | de' | hand-off
  static constexpr int E0 = 4 * PW > BWD_PRO_WS ? 4 * PW : BWD_PRO_WS;   // the four e buffers of the even rows sit BEHIND the node-side
  static constexpr int AREA = E0 + 4 * G::TILE_FLOATS;                   // prologue's scratch: their first DMA is issued before it runs
  static constexpr int WSB = G::TILES * 33 * 4;            // [t][32 entries + 1 zero entry] float4
  static constexpr size_t lds_floats(int TL) { return (size_t)AREA + (size_t)TL * QD_LD + 2 * G::TILES * 256 + WSB + 64; }   // + the prefetch dump (256 B)
};

template <int DE>
__global__ void __launch_bounds__(512, 4) k_block_bwd_v6(BlockArgs a) {
  seed_from_device(a);
#ifdef EGT_BWD_TIMING
  const unsigned tentry = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  using G = Geo<DE>;
  using G6 = Geo6<DE>;
  constexpr int NI = G::NF4 / 64;
  static_assert(G::NF4 % 64 == 0, "whole 1 KiB DMA chunks");
  static_assert(4 * G::EP <= G6::AREA, "edge partial staging must fit the tile area");
  static_assert((G6::AREA + 64) * 4 <= 65536, "LDS-DMA destinations (tile buffers, prefetch dump) must lie below 64 KiB");
  const float* e_in = a.e;
  const float* dey_in = a.de_out;
  float* dex_o = a.de;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kt = wave & 3;
  const bool roleB = wave >= 4;
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / a.NLR, lr = wg % a.NLR;
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL), nl = l_end - l_begin;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  float* tb = sm + kt * G6::PW;
  float* etE = sm + G6::E0 + kt * G::TILE_FLOATS;   // e / xhat tile of the even rows (li & 1 == 0)
  float* etO = tb;                                  // ... of the odd rows
  float* dt = tb + G::TILE_FLOATS;                  // de' tile
  float* sc1 = dt + G::TILE_FLOATS;           // A -> B, A: dGE [16 pairs][16]
  float* sc2 = sc1 + 256;                     // A -> A, B: [H_hat(8) | 1 | rstd | - | -] [16 pairs][12]
  float* xab = sc2 + 192;                     // A -> B: (dA0, dA1, at0, at1) of the lane
  float* xba = sc1;                           // B -> A (stage 1 -> 2): (dV_att.V dots (2), dH_ext (2)) of the lane [256]
  float* dump = sm + G6::AREA;                // 256 B nobody reads: destination of the L2 prefetches (tile_prefetch); LDS-DMA
                                              // destinations must lie below 64 KiB (M0 carries a 16-bit LDS address)
  float* qd = dump + 64;                      // [TL][QD_LD]
  float* wsA = qd + TL * QD_LD;               // projection weights   [t][lane] float4
  float* wsD = wsA + G::TILES * 256;          // d(xhat) weights      [t][lane] float4
  float* wsB = wsD + G::TILES * 256;          // dH_ext weights, rows with a head only: [t][33] float4 (entry 32 = zeros)
  const unsigned etE_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)etE);
  const unsigned etO_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)etO);
  const unsigned off0 = dma_lane_offset<DE>(lane);

  // ---- everything the first round needs that does not depend on the node-side prologue is requested FIRST: the first e tile
  //      (LDS-DMA into the even-row buffer, which lies behind the prologue's scratch), K / V fragments, the first de' tile; the
  //      staged query-side rows and the weight slabs follow in the same memory round trip.  The two roles are separate
  //      control-flow paths from here to the end of the kernel (what one role keeps in registers is not live in the other) ----
  const int ntile = N / 16;
  const size_t rowm0 = (size_t)b * N + min(kt, ntile - 1) * 16 + p;
  const size_t tile0 = (((size_t)b * N + l_begin) * N + min(kt, ntile - 1) * 16) * DE;
  auto stage_rows_and_slabs = [&]() {
    for (int i = threadIdx.x; i < nl * 40; i += 512) {
      const int r = i / 40, f = i % 40;
      const size_t rowl = (size_t)b * N + l_begin + r;
      if (a.pro && f >= 16 && f < 32) continue;   // dV_att comes from the prologue
      const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                       : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                                : a.stats + rowl * 32 + (f - 32) * 4;
      float4 v = *reinterpret_cast<const float4*>(src);
      if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
      *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
    }
    for (int i = threadIdx.x; i < G::TILES * 256; i += 512) {
      const int t = i >> 8, ln = (i >> 2) & 63, u = i & 3, pp = ln & 15, qq = ln >> 4;
      const int c = 16 * t + 4 * qq + u;
      wsA[i] = a.pw[c * 16 + pp];
      wsD[i] = a.pw[(16 * t + pp) * 16 + 4 * qq + u];
    }
    for (int i = threadIdx.x; i < G6::WSB; i += 512) {
      const int t = i / 132, r = i % 132, ent = r >> 2, u = r & 3;
      const int qq = ent >> 3, hd = ent & 7, c = 16 * t + 4 * qq + u;
      wsB[i] = (ent < 32 && c < DE) ? a.Wr[hd * DE + c] : 0.f;
    }
  };
#ifdef EGT_BWD_TIMING
  unsigned tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned tlast = 0, tstart = 0, rstart = 0;
#define T6START() do { tlast = (unsigned)__builtin_amdgcn_s_memtime(); tstart = tlast; rstart = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T6START() do {} while (0)
#endif
  if (!roleB) {
    // ===================================================================== role A
    tile_dma<DE>(etE_lds, e_in + tile0, off0);
    stage_rows_and_slabs();
    if (a.pro) {
      __syncthreads();
      bwd_node_prologue<DE, true>(a, sm, qd, b, l_begin, wg);   // written for the 256 threads of waves 0-3
    }
    float Kf[16];   // first needed behind the first row's projections: requested here, not across the prologue (registers)
    {
      const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm0 * QKVP + 64 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
      }
    }
    __syncthreads();
    T6START();
    v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
    for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f};
    float c2r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];
    for (int mt0 = 0; mt0 < ntile; mt0 += 4) {
      const int mt = mt0 + kt;
      if (mt >= ntile) {   // a pair without a key tile in this round still meets the barriers
        for (int l = l_begin; l < l_end; ++l) { wg_sync(); wg_sync(); wg_sync(); }
        continue;
      }
      const int m0 = mt * 16, m = m0 + p;
      const size_t rowm = (size_t)b * N + m;
      const float kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;
      if (mt0 > 0) {
        const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 kv = kp[i];
          Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
        }
        tile_dma<DE>(etE_lds, e_in + (((size_t)b * N + l_begin) * N + m0) * DE, off0);
      }
      vm_wait<0>();   // the first e tile of the key tile
      for (int l = l_begin; l < l_end; ++l) {
        const int li = l - l_begin;
        const size_t rowl = (size_t)b * N + l;
        const size_t pair0 = rowl * N + m0;
        float* et = (li & 1) ? etO : etE;
        const int lane = lane_now(), p = lane & 15, q = lane >> 4;   // (shadow the kernel-scope values: see lane_now)
        MaskRegs mr{make_float2(1.f, 1.f), 0};
        // ---- S1: e(l+1) -> the other e buffer (its readers, row l-1's P4 / P5, are behind the last barrier); it has the
        //      whole row to land and is retired at the end of S3 ----
        if (l + 1 < l_end)
          tile_dma<DE>(((li + 1) & 1) ? etO_lds : etE_lds, e_in + (pair0 + (size_t)N) * DE, off0);
        SCHED_FENCE();
        T6STAMP(0);
        float rstd;
        v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
        {
          float4 x[G::TILES];
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(et, p, q, t);
          rstd = ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) {
            frag_write<DE>(et, p, q, t, x[t]);     // xhat stays in the tile for P4 (A) and P5 (B)
            const float4 w = *reinterpret_cast<const float4*>(wsA + (t * 64 + lane) * 4);
            acc = MFMA(w.x, x[t].x, acc);
            acc = MFMA(w.y, x[t].y, acc);
            acc = MFMA(w.z, x[t].z, acc);
            acc = MFMA(w.w, x[t].w, acc);
          }
        }
        float dots[2];
        {   // QK^T of the lane's pair, heads 2q, 2q+1 (VALU work under the projection's MFMA chain)
          const float4* qp = reinterpret_cast<const float4*>(qd + li * QD_LD + q * 16);
          float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 u = qp[i];
            d0 = fmaf(u.x, Kf[4*i], d0);   d1 = fmaf(u.y, Kf[4*i+1], d1);
            d0 = fmaf(u.z, Kf[4*i+2], d0); d1 = fmaf(u.w, Kf[4*i+3], d1);
          }
          dots[0] = d0; dots[1] = d1;
        }
        SCHED_FENCE();
        T6STAMP(1);
        wg_sync();   // ---- barrier 1: B's dV_att.V dots / dH_ext are in xba, the de' tile is in dt
        T6STAMP(2);
        // ---- S2: P3 logits, softmax / gate backward ----
        float dge[4], hh[2], dA[2], at[2];
        {
          const float4 dd = *reinterpret_cast<const float4*>(xba + 4 * lane);   // (dAd0, dAd1, dH_ext0, dH_ext1)
          const float* qr = qd + li * QD_LD;
          const float4* sp = reinterpret_cast<const float4*>(qr + 128 + q * 8);
          const float4 s0 = sp[0], s1 = sp[1];
          const float st[8] = {s0.x, s0.y, s0.z, 0.f, s1.x, s1.y, s1.z, 0.f};
          const float dAd[2] = {dd.x, dd.y}, dhv[2] = {dd.z, dd.w};
          float xl[2], gl[2], inr[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float araw = dots[j] * a.scale;
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
          apply_masks<false>(a, kadd, mr, (pair0 + p) * BH, q, xl, gl);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float S = __expf(xl[j] - st[4 * j]) * st[4 * j + 1];
            const float g = gated ? egt_sigmoid(gl[j]) : 1.0f;
            const float dS = dAd[j] * g;
            const float dGl = gated ? dAd[j] * S * g * (1.0f - g) : 0.f;
            const float dH = S * (dS - st[4 * j + 2]) + dhv[j];
            dA[j] = dH * inr[j] * a.scale;
            at[j] = S * g;
            dge[2 * j] = dGl;
            dge[2 * j + 1] = dH;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[r] += dge[r];
        // (the reads of xba above are older DS operations of this wave than the writes below, which alias it)
        *reinterpret_cast<float4*>(sc1 + p * 16 + 4 * q) = make_float4(dge[0], dge[1], dge[2], dge[3]);
        *reinterpret_cast<float2*>(sc2 + p * 12 + 2 * q) = make_float2(hh[0], hh[1]);
        if (q == 0) sc2[p * 12 + 8] = 1.0f;
        if (q == 1) sc2[p * 12 + 9] = rstd;
        *reinterpret_cast<float4*>(xab + 4 * lane) = make_float4(dA[0], dA[1], at[0], at[1]);
        SCHED_FENCE();
        T6STAMP(3);
        wg_sync();   // ---- barrier 2
        T6STAMP(4);
        // ---- S3: P4 weight-gradient contractions over the 16 pairs of the tile ----
        {
          float bT[4], bR[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bT[s] = sc1[(q + 4 * s) * 16 + p];
            bR[s] = (p < 9) ? sc2[(q + 4 * s) * 12 + p] : 0.f;
          }
          const int lb = 64 * q + 4 * ((p >> 2) ^ q) + (p & 3);
          const float* eb = et + lb;
          const float* db = dt + lb;
#pragma unroll
          for (int t = 0; t < G::TILES; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              accT[t] = MFMA(elem_read_st<DE>(eb, et, p, q, s, t), bT[s], accT[t]);
              accR[t] = MFMA(elem_read_st<DE>(db, dt, p, q, s, t), bR[s], accR[t]);
            }
        }
        SCHED_FENCE();
        {   // dQ[l] partial over this tile's 16 keys -> HBM (VALU work under P4's MFMAs), summed over key tiles by the next
            // prologue (or k_node_bwd)
          float dq[16];
#pragma unroll
          for (int k = 0; k < 8; ++k) { dq[2 * k] = dA[0] * Kf[2 * k]; dq[2 * k + 1] = dA[1] * Kf[2 * k + 1]; }
          a.dqp[(((size_t)b * ntile + mt) * N + l) * 64 + lane] = reduce16_keep_own(dq, p);
        }
        vm_wait<1>();   // e(l+1) has landed (the only younger vector-memory operation of this wave is the store above)
        SCHED_FENCE();
        T6STAMP(5);
        wg_sync();   // ---- barrier 3: the tile buffers and hand-off areas of row l are free
        T6STAMP(6);
      }
    }
    // per-workgroup edge-parameter-gradient partial: T | s | R of this wave -> LDS (every pair is past its last barrier 3:
    // the tile areas are idle), summed over the four A waves below
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[r] = row_sum16(ssum[r]);
    __syncthreads();
    float* ep = sm + kt * G::EP;
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
  } else {
    // ===================================================================== role B
    float Vf[16];
    TileRegs<DE> td;
    tile_gload<DE>(td, dey_in + tile0, lane, 16);
    {
      const float4* vp = reinterpret_cast<const float4*>(a.qkvp + rowm0 * QKVP + 128 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 vv = vp[i];
        Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
      }
    }
    stage_rows_and_slabs();
    if (a.pro) {
      __syncthreads();
      const int nb = a.pro == 2 ? 4 : 2;   // the barriers of the node-side prologue, which waves 0-3 run
      for (int i = 0; i < nb; ++i) __syncthreads();
    }
    __syncthreads();
    T6START();
    for (int mt0 = 0; mt0 < ntile; mt0 += 4) {
      const int mt = mt0 + kt;
      if (mt >= ntile) {
        for (int l = l_begin; l < l_end; ++l) { wg_sync(); wg_sync(); wg_sync(); }
        continue;
      }
      const int m0 = mt * 16, m = m0 + p;
      const size_t rowm = (size_t)b * N + m;
      float dKa[16], dVa[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
      if (mt0 > 0) {
        const float4* vp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 128 + q * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 vv = vp[i];
          Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
        }
        tile_gload<DE>(td, dey_in + (((size_t)b * N + l_begin) * N + m0) * DE, lane, 16);
      }
      const unsigned dump_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dump);
      const unsigned pf_off = (unsigned)lane * 64u < (unsigned)G::TILE_FLOATS * 4u ? (unsigned)lane * 64u : 0u;
      // the lane's slot in the compact dH_ext slab: rows 4q', 4q'+1 carry heads, the others read the zero entry
      const float* wbl = wsB + (((p & 2) == 0) ? (q * 8 + (p >> 2) * 2 + (p & 1)) : 32) * 4;
      for (int l = l_begin; l < l_end; ++l) {
        const int li = l - l_begin;
        const size_t rowl = (size_t)b * N + l;
        const size_t pair0 = rowl * N + m0;
        const float* et = (li & 1) ? etO : etE;
        const float* qr = qd + li * QD_LD;
        const int lane = lane_now(), p = lane & 15, q = lane >> 4;   // (shadow the kernel-scope values: see lane_now)
        // ---- S1: de'(l) -> LDS tile; P2: dH_ext = de'.Wr^T; QK^T / dV_att.V dots ----
        tile_lds_put<DE>(dt, td, lane, 16);   // (the compiler's vmcnt wait for de' sits here)
        lds_sync();
        T6STAMP(0);
        v4f dhx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 dyv = frag_read<DE>(dt, p, q, t);
          const float4 w = *reinterpret_cast<const float4*>(wbl + t * 132);
          dhx = MFMA(w.x, dyv.x, dhx);
          dhx = MFMA(w.y, dyv.y, dhx);
          dhx = MFMA(w.z, dyv.z, dhx);
          dhx = MFMA(w.w, dyv.w, dhx);
        }
        {
          const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
          float e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = dp[i];
            e0 = fmaf(v.x, Vf[4*i], e0);   e1 = fmaf(v.y, Vf[4*i+1], e1);
            e0 = fmaf(v.z, Vf[4*i+2], e0); e1 = fmaf(v.w, Vf[4*i+3], e1);
          }
          *reinterpret_cast<float4*>(xba + 4 * lane) = make_float4(e0, e1, dhx[0], dhx[1]);
        }
        SCHED_FENCE();
        T6STAMP(1);
        wg_sync();   // ---- barrier 1
        T6STAMP(2);
        // ---- S2: de'(l+1) -> L2 (no registers) ----
        if (l + 1 < l_end) tile_prefetch(dump_lds, dey_in + (pair0 + (size_t)N) * DE, pf_off);
        SCHED_FENCE();
        T6STAMP(3);
        wg_sync();   // ---- barrier 2: dGE, rstd, (dA, A~) of row l are in LDS
        T6STAMP(4);
        // ---- S3: dK, dV, dQ partial; P5 ----
        float dge[4];
        {
          const float4 g4 = *reinterpret_cast<const float4*>(sc1 + p * 16 + 4 * q);
          dge[0] = g4.x; dge[1] = g4.y; dge[2] = g4.z; dge[3] = g4.w;
        }
        const float rstd = sc2[p * 12 + 9];
        {
          const float4 da = *reinterpret_cast<const float4*>(xab + 4 * lane);
          const float dA[2] = {da.x, da.y}, at[2] = {da.z, da.w};
          const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
          const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 u = qp[i], v = dp[i];
            dKa[4*i]   = fmaf(dA[0], u.x, dKa[4*i]);   dKa[4*i+1] = fmaf(dA[1], u.y, dKa[4*i+1]);
            dKa[4*i+2] = fmaf(dA[0], u.z, dKa[4*i+2]); dKa[4*i+3] = fmaf(dA[1], u.w, dKa[4*i+3]);
            dVa[4*i]   = fmaf(at[0], v.x, dVa[4*i]);   dVa[4*i+1] = fmaf(at[1], v.y, dVa[4*i+1]);
            dVa[4*i+2] = fmaf(at[0], v.z, dVa[4*i+2]); dVa[4*i+3] = fmaf(at[1], v.w, dVa[4*i+3]);
          }
          // pin the updates here: left alone, hipcc sinks them into the loop latch (their results are not needed before the
          // next row) and keeps the 32 registers of Q / dV_att alive across P5 -- which then spills
#pragma unroll
          for (int i = 0; i < 16; ++i) { asm volatile("" : "+v"(dKa[i])); asm volatile("" : "+v"(dVa[i])); }
        }
        SCHED_FENCE();
        {
          float4 dxh[G::TILES];
          float m1 = 0.f, m2 = 0.f;
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) {
            const float4 xh = frag_read<DE>(et, p, q, t);
            v4f d = {0.f, 0.f, 0.f, 0.f};
            const float4 w = *reinterpret_cast<const float4*>(wsD + (t * 64 + lane) * 4);
            d = MFMA(w.x, dge[0], d);
            d = MFMA(w.y, dge[1], d);
            d = MFMA(w.z, dge[2], d);
            d = MFMA(w.w, dge[3], d);
            dxh[t] = make_float4(d[0], d[1], d[2], d[3]);
            m1 += (d[0] + d[1]) + (d[2] + d[3]);
            m2 = fmaf(d[0], xh.x, m2); m2 = fmaf(d[1], xh.y, m2);
            m2 = fmaf(d[2], xh.z, m2); m2 = fmaf(d[3], xh.w, m2);
          }
          m1 = sum_over_q(m1) * (1.0f / DE);
          m2 = sum_over_q(m2) * (1.0f / DE);
          if (a.flags & EGT_BF_NO_EDGE_LN) { m1 = 0.f; m2 = 0.f; }   // no norm_edge: d e = de' + d(proj input)
          float* orow = dex_o + (pair0 + p) * DE + 4 * q;
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) {
            if (16 * t + 4 * q < DE) {
              const float4 dyv = frag_read<DE>(dt, p, q, t);
              const float4 xh = frag_read<DE>(et, p, q, t);
              float4 o;
              o.x = dyv.x + rstd * (dxh[t].x - m1 - xh.x * m2);
              o.y = dyv.y + rstd * (dxh[t].y - m1 - xh.y * m2);
              o.z = dyv.z + rstd * (dxh[t].z - m1 - xh.z * m2);
              o.w = dyv.w + rstd * (dxh[t].w - m1 - xh.w * m2);
              *reinterpret_cast<float4*>(orow + 16 * t) = o;
            }
          }
        }
        SCHED_FENCE();
        // de'(l+1) from L2 into the registers P5 has just freed; it lands while this wave waits for A's P4 at barrier 3
        if (l + 1 < l_end) tile_gload<DE>(td, dey_in + (pair0 + (size_t)N) * DE, lane, 16);
        SCHED_FENCE();
        T6STAMP(5);
        wg_sync();   // ---- barrier 3
        T6STAMP(6);
      }
      float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
      float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
        vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
      }
    }
    __syncthreads();   // (A stages its partials behind this one)
  }
#ifdef EGT_BWD_TIMING
  if (a.dbg && lane == 0) {
    unsigned* o = a.dbg + ((size_t)wg * 8 + wave) * 16;
    for (int i = 0; i < 8; ++i) o[i] = tacc[i];
    o[13] = tlast - tstart;         // loop total
    o[8] = tstart - tentry;         // kernel entry -> loop (staging, node-side prologue, weight slabs)
    o[9] = (unsigned)__builtin_amdgcn_s_memrealtime() - rstart;   // loop (+ this wave's tail) in 10 ns ticks
    o[10] = (unsigned)__builtin_amdgcn_s_memtime() - tstart;
  }
#endif
  __syncthreads();
  float* out = a.epart + (size_t)wg * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 512)
    out[i] = (sm[i] + sm[G::EP + i]) + (sm[2 * G::EP + i] + sm[3 * G::EP + i]);
}

#ifdef EGT_BWD_TIMING
// measurement builds only (EGT_BWD6_FLAGS=-DEGT_BWD_TIMING): per-stage cycle sums of both roles (synchronises after every launch)
static unsigned* g_bt_dev = nullptr;
static int g_bt_n = 0;
static long g_bt_launch = 0;
static double g_bt6_sum[2][16];
static long g_bt6_waves[2];
static void bwd_timing_report6() {
  static const char* nm[2][7] = {{"A wait e (DMA)", "A P1 LN+proj", "A barrier 1", "A P3 softmax bwd", "A barrier 2", "A P4 wgrad MFMA", "A barrier 3"},
                                 {"B de' put", "B P2 dHext + dots", "B barrier 1", "B de' request", "B barrier 2", "B dQ/dK/dV + P5 + store", "B barrier 3"}};
  for (int r = 0; r < 2; ++r) {
    if (!g_bt6_waves[r]) continue;
    fprintf(stderr, "[egt] k_block_bwd_v6 role %c phase cycles per wave (mean over %ld waves):\n", "AB"[r], g_bt6_waves[r]);
    for (int i = 0; i < 7; ++i)
      fprintf(stderr, "    %-28s %10.0f  (%.1f %%)\n", nm[r][i], g_bt6_sum[r][i] / g_bt6_waves[r], 100.0 * g_bt6_sum[r][i] / g_bt6_sum[r][13]);
    fprintf(stderr, "    %-28s %10.0f\n", "loop total", g_bt6_sum[r][13] / g_bt6_waves[r]);
    fprintf(stderr, "    %-28s %10.0f\n", "entry -> loop", g_bt6_sum[r][8] / g_bt6_waves[r]);
    fprintf(stderr, "    %-28s %10.3f GHz (%.0f cycles in %.2f us)\n", "shader clock over the loop", g_bt6_sum[r][10] / g_bt6_sum[r][9] * 0.1,
            g_bt6_sum[r][10] / g_bt6_waves[r], g_bt6_sum[r][9] / g_bt6_waves[r] * 0.01);
  }
}
#endif

template <int DE>
static void launch6(BlockArgs& a, int nwg, hipStream_t st) {
  const size_t lds = Geo6<DE>::lds_floats(BWD_TL) * 4;
  EGT_MAX_LDS_ONCE(k_block_bwd_v6<DE>);
#ifdef EGT_BWD_TIMING
  if (g_bt_n < nwg) {
    if (g_bt_dev) (void)hipFree(g_bt_dev);
    (void)hipMalloc(&g_bt_dev, (size_t)nwg * 128 * sizeof(unsigned));
    if (!g_bt_n) atexit(bwd_timing_report6);
    g_bt_n = nwg;
  }
  a.dbg = g_bt_dev; a.dbg_t0 = 0;
#endif
  EGT_LAUNCH("k_block_bwd", (k_block_bwd_v6<DE>), dim3(nwg), dim3(512), lds, st, a);
#ifdef EGT_BWD_TIMING
  (void)hipStreamSynchronize(st);
  static std::vector<unsigned> h;
  h.resize((size_t)nwg * 128);
  (void)hipMemcpy(h.data(), a.dbg, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  if (++g_bt_launch > 20)
    for (size_t w = 0; w < (size_t)nwg * 8; ++w) {
      const int r = (int)((w & 7) >> 2);
      for (int i = 0; i < 14; ++i) g_bt6_sum[r][i] += h[w * 16 + i];
      ++g_bt6_waves[r];
    }
#endif
}

void egt_bwd6_launch(BlockArgs& a, int nwg, hipStream_t st) {
  switch (a.De) {
    case 32: launch6<32>(a, nwg, st); break;
    case 48: launch6<48>(a, nwg, st); break;
    default: launch6<64>(a, nwg, st); break;
  }
}
