// Backward pair kernels of the fused attention block (included by egt_block.hip, which holds the dispatch):
// k_block_bwd_v4 (mask tensors / ragged N / bf16), k_block_bwd_v5 (LDS-DMA staged e tiles: the headline),
// k_block_bwd_v4r (narrow edge channels, R rows per iteration) and the weight-gradient epilogue they share.
#pragma once
// Cache-policy hints of the streamed tiles (egt_tile.h) in k_block_bwd_v5: e_l is read once (LDS-DMA, non-temporal), de' was written by
// the launch before and de is the next launch's de' (both cached: 134 MB of the 256 MB memory-side cache at the headline batch).
// Measured (tools/ab.sh, same box): 100.9 -> 97.7 us; de stores non-temporal 99-100 us, de' loads non-temporal: no change.
#ifndef EGT_NT_BWD_E
#define EGT_NT_BWD_E true
#endif
#ifndef EGT_NT_BWD_DY
#define EGT_NT_BWD_DY false
#endif
#ifndef EGT_NT_BWD_ST
#define EGT_NT_BWD_ST false
#endif

// ================================================================ backward =====
// Workgroup = (graph b, TL query rows); wave w owns key tiles w, w+4, ...; for each it walks the TL rows.  Q / dV_att /
// softmax statistics of the rows sit in LDS.
// ---- register-lean general variant ("v4": mask tensors, ragged N, bf16 edge tensors on the wide tiles) ----
// Two wavefronts fit a
// SIMD (<= 256 registers, <= 80 KiB LDS per workgroup): the long per-tile dependency chain
// (LayerNorm -> MFMA chain -> exp/sigmoid -> MFMA chain -> LayerNorm backward) is latency-,
// not throughput-bound, so a second resident wave is worth more than fat register tiles.
//  * all lane-constant MFMA weight operands live in LDS as [t][lane] float4 slabs
//    (conflict-free ds_read_b128) and are fetched right before their MFMA group;
//  * xhat and de' fragments are re-read from their LDS tiles where they are needed again
//    (LayerNorm backward) instead of being held across the tile;
//  * dQ partials go to HBM per key tile (summed in k_node_bwd) instead of an LDS slab;
//  * scheduling fences between the phases keep the compiler from hoisting every LDS read to
//    the top of the tile (which is what blows the register budget);
//  * phase guards: P2, the dQ/dK/dV block, P4 and P5 sit behind `if (!(a.guard & bit))` with
//    a.guard == 0 at run time.  The always-taken uniform branches split the tile body into
//    basic blocks, which stops hipcc from stretching live ranges across phases: 19 -> 4 spilled
//    registers, 116 -> 104 us.
// RAG: N is not a multiple of 16 -- the last key tile is zero-filled past N and its lanes get probability and
// gate exactly 0, the last row group is short (the row loop and the prologue already take nl < 16).
template <int DE, bool ML, bool BF, bool RAG>
__global__ void __launch_bounds__(256, 2) k_block_bwd_v4(BlockArgs a) {
  seed_from_device(a);
  using G = Geo<DE>;
  typedef typename EdgeT<BF>::type ET;   // element type of the edge tensors in HBM
  const ET* e_in = reinterpret_cast<const ET*>(a.e);
  ET* e_o = reinterpret_cast<ET*>(a.e_out);
  const ET* dey_in = reinterpret_cast<const ET*>(a.de_out);
  ET* dex_o = reinterpret_cast<ET*>(a.de);
  (void)e_in; (void)e_o; (void)dey_in; (void)dex_o;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / a.NLR, lr = wg % a.NLR;
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL), nl = l_end - l_begin;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  constexpr int PW = 3 * G::TILE_FLOATS + 256 + 192;
  constexpr int WSLAB = G::TILES * 256;      // one weight slab: [TILES][64 lanes] float4
  float* et = sm + wave * PW;
  float* dt0 = et + G::TILE_FLOATS;
  float* sc1 = dt0 + 2 * G::TILE_FLOATS;
  float* sc2 = sc1 + 256;
  constexpr int AREA = 4 * PW > BWD_PRO_WS ? 4 * PW : BWD_PRO_WS;   // per-wave tiles; also the prologue's scratch
  float* qd = sm + AREA;                     // [TL][QD_LD]
  float* wsA = qd + TL * QD_LD;              // prologue weights   wA[4t+u]
  float* wsB = wsA + WSLAB;                  // dH_ext weights     wrB[4t+u]
  float* wsD = wsB + WSLAB;                  // d(ehat) weights    wD[t][s]
  bwd_stage_rows<256, 3>(a, qd, b, l_begin, nl);
  if (a.pro) {
    __syncthreads();
    bwd_node_prologue<DE>(a, sm, qd, b, l_begin, wg);
  }
  // weight slabs: element (t, lane, u)
  for (int i = threadIdx.x; i < G::TILES * 256; i += 256) {
    const int t = i >> 8, ln = (i >> 2) & 63, u = i & 3, pp = ln & 15, qq = ln >> 4;
    const int c = 16 * t + 4 * qq + u;
    wsA[i] = a.pw[c * 16 + pp];
    const int hd = 2 * (pp >> 2) + (pp & 1);
    wsB[i] = ((pp & 2) == 0 && c < DE) ? a.Wr[hd * DE + c] : 0.f;
    wsD[i] = a.pw[(16 * t + pp) * 16 + 4 * qq + u];
  }
  float c2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];

  v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int ntile = RAG ? (N + 15) / 16 : N / 16;
  for (int mt = wave; mt < ntile; mt += 4) {
    const int m0 = mt * 16, m = m0 + p;
    const int kv = RAG ? min(16, N - m0) : 16;
    const bool kvalid = RAG ? (m < N) : true;
    float Kf[16], Vf[16], dKa[16], dVa[16];
    const size_t rowm = (size_t)b * N + (kvalid ? m : N - 1);
    {
      const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
      const float4* vp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 128 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i], vv = vp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
        Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
    }
    const float kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;

    TileRegs<DE> te, td;
    for (int l = l_begin; l < l_end; ++l) {
      const int li = l - l_begin;
      const size_t rowl = (size_t)b * N + l;
      const size_t pair0 = rowl * N + m0;
      float* dt = dt0 + (li & 1) * G::TILE_FLOATS;
      MaskRegs mr{make_float2(1.f, 1.f), 0};
      mask_gload<ML>(a, mr, pair0 + (kvalid ? p : 0), q);
      // memory order per step: [stores of row l-1] then [loads of row l+1] (see k_block_fwd)
      lds_sync();
      if (li > 0)
        tile_from_lds<DE>(dt0 + ((li - 1) & 1) * G::TILE_FLOATS, dex_o + (pair0 - (size_t)N) * DE, lane, kv);
      const size_t lp0 = (a.guard & 16) ? (size_t)wave * 16 : pair0;
      tile_gload<DE>(td, dey_in + lp0 * DE, lane, kv);   // (two resident waves hide the HBM latency: a register prefetch of the next row only spilled)
      tile_gload<DE>(te, e_in + lp0 * DE, lane, kv);
      tile_lds_put<DE>(et, te, lane, kv);
      tile_lds_put<DE>(dt, td, lane, kv);
      lds_sync();
      SCHED_FENCE();
      // ---- P1: norm_edge, projections (recompute) ----
      float rstd;
      v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
      {
        float4 x[G::TILES];
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(et, p, q, t);
        rstd = ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          frag_write<DE>(et, p, q, t, x[t]);     // xhat stays in the tile for the later phases
          const float4 w = *reinterpret_cast<const float4*>(wsA + (t * 64 + lane) * 4);
          acc = MFMA(w.x, x[t].x, acc);
          acc = MFMA(w.y, x[t].y, acc);
          acc = MFMA(w.z, x[t].z, acc);
          acc = MFMA(w.w, x[t].w, acc);
        }
      }
      SCHED_FENCE();
      // ---- P2: dH_ext = de'.Wr^T ----
      v4f dhx = {0.f, 0.f, 0.f, 0.f};
      if (!(a.guard & 8))
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        const float4 dyv = frag_read<DE>(dt, p, q, t);
        const float4 w = *reinterpret_cast<const float4*>(wsB + (t * 64 + lane) * 4);
        dhx = MFMA(w.x, dyv.x, dhx);
        dhx = MFMA(w.y, dyv.y, dhx);
        dhx = MFMA(w.z, dyv.z, dhx);
        dhx = MFMA(w.w, dyv.w, dhx);
      }
      SCHED_FENCE();
      // ---- P3: logits, softmax/gate backward, dQ/dK/dV ----
      // Q / dV_att fragments are fetched from LDS twice (once for the dot products, once for the
      // dK/dV accumulation) instead of being held across the exp / sigmoid chain.
      float dge[4], hh[2], dA[2], at[2];
      {
        const float* qr = qd + li * QD_LD;
        float dots[2], dAd[2];
        {
          const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
          const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
          float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 u = qp[i], v = dp[i];
            d0 = fmaf(u.x, Kf[4*i], d0);   d1 = fmaf(u.y, Kf[4*i+1], d1);
            d0 = fmaf(u.z, Kf[4*i+2], d0); d1 = fmaf(u.w, Kf[4*i+3], d1);
            e0 = fmaf(v.x, Vf[4*i], e0);   e1 = fmaf(v.y, Vf[4*i+1], e1);
            e0 = fmaf(v.z, Vf[4*i+2], e0); e1 = fmaf(v.w, Vf[4*i+3], e1);
          }
          dots[0] = d0; dots[1] = d1; dAd[0] = e0; dAd[1] = e1;
        }
        const float4* sp = reinterpret_cast<const float4*>(qr + 128 + q * 8);
        const float4 s0 = sp[0], s1 = sp[1];
        const float st[8] = {s0.x, s0.y, s0.z, 0.f, s1.x, s1.y, s1.z, 0.f};
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
        apply_masks<ML>(a, kadd, mr, (pair0 + p) * BH, q, xl, gl);
        if (RAG && !kvalid) { xl[0] = xl[1] = -3.0e38f; gl[0] = gl[1] = -3.0e38f; }   // a key past N: S = 0, gate = 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float S = __expf(xl[j] - st[4 * j]) * st[4 * j + 1];
          const float g = gated ? egt_sigmoid(gl[j]) : 1.0f;
          const float dS = dAd[j] * g;
          const float dGl = gated ? dAd[j] * S * g * (1.0f - g) : 0.f;
          const float dH = S * (dS - st[4 * j + 2]) + dhx[j];
          dA[j] = dH * inr[j] * a.scale;
          at[j] = S * g;
          dge[2 * j] = dGl;
          dge[2 * j + 1] = dH;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[r] += dge[r];
      *reinterpret_cast<float4*>(sc1 + p * 16 + 4 * q) = make_float4(dge[0], dge[1], dge[2], dge[3]);
      *reinterpret_cast<float2*>(sc2 + p * 12 + 2 * q) = make_float2(hh[0], hh[1]);
      if (q == 0) sc2[p * 12 + 8] = 1.0f;
      lds_sync();
      SCHED_FENCE();
      if (!(a.guard & 4)) {
        const float* qr = qd + li * QD_LD;
        const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
        const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
        float dq[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 u = qp[i], v = dp[i];
          dKa[4*i]   = fmaf(dA[0], u.x, dKa[4*i]);   dKa[4*i+1] = fmaf(dA[1], u.y, dKa[4*i+1]);
          dKa[4*i+2] = fmaf(dA[0], u.z, dKa[4*i+2]); dKa[4*i+3] = fmaf(dA[1], u.w, dKa[4*i+3]);
          dVa[4*i]   = fmaf(at[0], v.x, dVa[4*i]);   dVa[4*i+1] = fmaf(at[1], v.y, dVa[4*i+1]);
          dVa[4*i+2] = fmaf(at[0], v.z, dVa[4*i+2]); dVa[4*i+3] = fmaf(at[1], v.w, dVa[4*i+3]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { dq[2 * k] = dA[0] * Kf[2 * k]; dq[2 * k + 1] = dA[1] * Kf[2 * k + 1]; }
        // dQ[l] partial over this tile's 16 keys -> HBM, summed over key tiles by the next prologue (or k_node_bwd)
        a.dqp[(((size_t)b * ntile + mt) * N + l) * 64 + lane] = reduce16_keep_own(dq, p);
      }
      SCHED_FENCE();
      // ---- P4: weight-gradient contractions over the 16 pairs of the tile ----
      if (!(a.guard & 1)) {
        float bT[4], bR[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bT[s] = sc1[(q + 4 * s) * 16 + p];
          bR[s] = (p < 9) ? sc2[(q + 4 * s) * 12 + p] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < G::TILES; ++t)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            accT[t] = MFMA(elem_read<DE>(et, q + 4 * s, 16 * t + p), bT[s], accT[t]);
            accR[t] = MFMA(elem_read<DE>(dt, q + 4 * s, 16 * t + p), bR[s], accR[t]);
          }
      }
      lds_sync();
      SCHED_FENCE();
      // ---- P5: d(ehat) = Wp . dGE, LayerNorm backward, de = de' + ... in place over the de' tile ----
      if (!(a.guard & 2)) {
        float4 dxh[G::TILES];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 w = *reinterpret_cast<const float4*>(wsD + (t * 64 + lane) * 4);
          const float4 xh = frag_read<DE>(et, p, q, t);
          v4f d = {0.f, 0.f, 0.f, 0.f};
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
        lds_sync();
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 dyv = frag_read<DE>(dt, p, q, t);
          const float4 xh = frag_read<DE>(et, p, q, t);
          float4 o;
          o.x = dyv.x + rstd * (dxh[t].x - m1 - xh.x * m2);
          o.y = dyv.y + rstd * (dxh[t].y - m1 - xh.y * m2);
          o.z = dyv.z + rstd * (dxh[t].z - m1 - xh.z * m2);
          o.w = dyv.w + rstd * (dxh[t].w - m1 - xh.w * m2);
          frag_write<DE>(dt, p, q, t, o);
        }
      }
      SCHED_FENCE();
    }
    {  // flush the last row of this key tile
      lds_sync();
      tile_from_lds<DE>(dt0 + ((nl - 1) & 1) * G::TILE_FLOATS,
                        dex_o + (((size_t)b * N + l_end - 1) * N + m0) * DE, lane, kv);
    }
    float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
    float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
    if (kvalid) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
        vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) ssum[r] = row_sum16(ssum[r]);
  __syncthreads();
  float* ep = sm + wave * G::EP;
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
  float* out = a.epart + (size_t)wg * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 256)
    out[i] = (sm[i] + sm[G::EP + i]) + (sm[2 * G::EP + i] + sm[3 * G::EP + i]);
}

// ================================================ backward, LDS-DMA staged ("v5") ====
// k_block_bwd_v4 with the HBM latency of the two streamed tiles taken off the wave's critical path
// WITHOUT spending registers (v4's register prefetch spills at two waves per SIMD):
//  * the e tile of row l+1 is fetched by LDS-DMA (global_load_lds_dwordx4: HBM -> LDS, no VGPR
//    destination) into the second e-tile buffer while row l is being computed; the wave counts its
//    own vmcnt for it (hipcc does not see inline-asm memory operations).  The DMA writes lane-linear
//    1 KiB chunks, so the XOR swizzle of the De = 64 tile goes on the SOURCE address (lane L of
//    chunk i fetches slot (L & 15) ^ row of row 4i + (L >> 4)); reads stay swizzled as before.
//    It is issued after the wave's only compiler-counted loads of the iteration (de') have been
//    waited for, so no compiler wait ever covers a DMA in flight;
//  * the de' tile of row l is requested at the top of the iteration and consumed after P1 (LayerNorm +
//    projections need only e): its latency hides under P1 and the partner wave;
//  * de leaves from the registers that hold it (64-byte row segments, the four stores of a tile fill
//    whole lines) instead of making an LDS round trip through a second de' buffer.
// Same LDS footprint as v4 (two e buffers + one de' buffer instead of one + two): two workgroups per CU.
// fp32 edge tensors, no mask tensors, De a multiple of 16.  RAG: N is not a multiple of 16 -- the last key tile has kv < 16 keys:
// its e tiles are fetched with the missing rows replaced by the last valid one (tile_dma_ragged), its de' rows are zero-filled,
// the lanes of the missing keys get probability and gate exactly 0 (so every product they enter is 0) and store nothing.
// MM = EGT_MM_BF16X3 (opt-in: EGT_BWD_MATMUL=bf16x3): the three channel contractions of a tile (P1 projections,
// P2 dH_ext, P5 d ehat: 48 of the 80 fp32 MFMAs) run as 3-term bfloat16 split products on the bf16 matrix pipe
// (20 MFMAs of 16 cycles; per-product error 2^-16, fp32 accumulate); the weight-gradient contractions over the
// pair axis (P4) and everything else stay exact fp32.  Same LDS footprint: a bf16 hi + lo pair is as large as the
// fp32 value it replaces.
// floats of k_block_bwd_v5's tile area: four waves x (two e buffers + de' tile + hand-off tiles), and at least the four parity-0 e
// buffers + the node-side prologue's scratch behind them
#ifndef V5_BALANCE
#define V5_BALANCE 1   // k_block_bwd_v5: balanced (tile, row) ranges when the key-tile count is not a multiple of 4 (A/B: 0)
#endif
#define V5_AREA(DE_) ((4 * (3 * Geo<DE_>::TILE_FLOATS + 448) > 4 * Geo<DE_>::TILE_FLOATS + BWD_PRO_WS) \
                          ? 4 * (3 * Geo<DE_>::TILE_FLOATS + 448) : 4 * Geo<DE_>::TILE_FLOATS + BWD_PRO_WS)
template <int DE, int MM, bool RAG>
__global__ void __launch_bounds__(256, 2) k_block_bwd_v5(BlockArgs a) {
  seed_from_device(a);
#define PSTAMP(i) do {} while (0)
  constexpr bool SPLIT = MM == EGT_MM_BF16X3;
  constexpr int NS = (Geo<DE>::TILES + 1) / 2;   // 16x16x32 steps over the channel axis
  (void)SPLIT; (void)NS;
  using G = Geo<DE>;
  const float* e_in = a.e;
  const float* dey_in = a.de_out;
  float* dex_o = a.de;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / a.NLR, lr = wg % a.NLR;
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL), nl = l_end - l_begin;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  constexpr int PW = 3 * G::TILE_FLOATS + 256 + 192;
  // one weight slab: [TILES][64 lanes] float4; the bf16 slabs of an odd tile count are padded to whole 32-channel steps
  constexpr int WSLAB = (MM != 0 && NS * 512 > G::TILES * 256) ? NS * 512 : G::TILES * 256;
  // Tile area: the four waves' FIRST e buffers (row parity 0) lie in front of everything else, outside the node-side prologue's
  // scratch, so a wave's first e tile can be on its way (LDS-DMA) while the prologue runs; the second e buffer, the de' tile and
  // the two small hand-off tiles of a wave follow, and the prologue's scratch lies over those.
  constexpr int PW2 = 2 * G::TILE_FLOATS + 256 + 192;
  float* et0 = sm + wave * G::TILE_FLOATS;   // e / xhat tile, row parity 0
  float* et1 = sm + 4 * G::TILE_FLOATS + wave * PW2;   // ... row parity 1
  const int estr = __builtin_amdgcn_readfirstlane((int)(et1 - et0));   // floats from a wave's parity-0 to its parity-1 buffer
  float* dt = et1 + G::TILE_FLOATS;          // de' tile
  float* sc1 = dt + G::TILE_FLOATS;
  float* sc2 = sc1 + 256;
  constexpr int AREA = V5_AREA(DE);          // per-wave tiles; the prologue's scratch starts behind the parity-0 buffers
  static_assert(4 * PW == 4 * G::TILE_FLOATS + 4 * PW2 && AREA >= 4 * PW && AREA >= 4 * G::TILE_FLOATS + BWD_PRO_WS, "tile area");
  float* pro_ws = sm + 4 * G::TILE_FLOATS;
  float* qd = sm + AREA;                     // [TL][QD_LD]
  float* wsA = qd + TL * QD_LD;              // prologue weights   wA[4t+u]
  float* wsB = wsA + WSLAB;                  // dH_ext weights     wrB[4t+u]
  float* wsD = wsB + WSLAB;                  // d(ehat) weights    wD[t][s]
  volatile int* pflag = reinterpret_cast<volatile int*>(wsD + WSLAB);   // [4]: wave w parked the partial of the tile it shares with wave w - 1
  if (threadIdx.x < 4) pflag[threadIdx.x] = 0;
  // LDS byte address of the wave's e buffers (the kernel's only LDS object is the dynamic array: offset 0)
  const unsigned et_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)et0);
  const unsigned off0 = dma_lane_offset<DE>(lane);
  // The wave's first key tile: its first e tile (LDS-DMA into the parity-0 buffer), the K / V fragments of its keys and the key-mask
  // byte are requested FIRST -- ahead of the start-up staging and the node-side prologue, which need ~9 us and touch neither the
  // parity-0 buffers nor these registers -- so the row loop starts on data that has landed (the first row of a workgroup cost
  // 8.1 us against 4.35 us for every later one: DESIGN_LOG.md, round 5, ablation builds).
  const int ntile = RAG ? (N + 15) / 16 : N / 16;
  // Work of a wave: key tiles w, w + 4, ... when the tile count is a multiple of 4 (or < 3); otherwise (N = 37: three tiles for four
  // waves -- one wave idle in the row loop; N = 100: seven) the ntile x nl (tile, row) steps are cut into four CONTIGUOUS equal
  // ranges: a tile that straddles two ranges is shared by neighbouring waves, whose dK / dV meet in the tile's partial slot (the later
  // wave parks its sum there, the earlier one adds its own: same CU, an LDS flag -- as k_narrow_bwd does).  At least three tiles: a
  // range is then never shorter than 3/4 of a tile and no tile has more than two waves.
  const bool balance = RAG && V5_BALANCE && ntile >= 3 && (ntile & 3) != 0;
  const int T_ = ntile * nl;
  const int t0 = balance ? (wave * T_) >> 2 : 0, t1 = balance ? ((wave + 1) * T_) >> 2 : 0;
  const bool have = balance ? t1 > t0 : wave < ntile;
  const int mt_first = balance ? t0 / nl : wave, mt_last = balance ? (t1 - 1) / nl : ntile - 1, mt_step = balance ? 1 : 4;
  const int r0_first = balance ? t0 - mt_first * nl : 0;
  float Kf[16], Vf[16];
  int kmv = 1;
#define V5_TILE_START(MT_, ROW_)                                                                                    \
  do {                                                                                                              \
    const int m0_ = (MT_) * 16, kv_ = RAG ? min(16, N - m0_) : 16;                                                  \
    if (RAG && kv_ < 16) tile_dma_ragged<DE>(et_lds, e_in + (((size_t)b * N + l_begin + (ROW_)) * N + m0_) * DE, lane, kv_); \
    else tile_dma<DE, EGT_NT_BWD_E>(et_lds, e_in + (((size_t)b * N + l_begin + (ROW_)) * N + m0_) * DE, off0);      \
    const size_t rowm_ = (size_t)b * N + (RAG ? min(m0_ + p, N - 1) : m0_ + p);                                     \
    const float4* kp_ = reinterpret_cast<const float4*>(a.qkvp + rowm_ * QKVP + 64 + q * 16);                       \
    const float4* vp_ = reinterpret_cast<const float4*>(a.qkvp + rowm_ * QKVP + 128 + q * 16);                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
      const float4 kv4 = kp_[i], vv4 = vp_[i];                                                                      \
      Kf[4*i] = kv4.x; Kf[4*i+1] = kv4.y; Kf[4*i+2] = kv4.z; Kf[4*i+3] = kv4.w;                                     \
      Vf[4*i] = vv4.x; Vf[4*i+1] = vv4.y; Vf[4*i+2] = vv4.z; Vf[4*i+3] = vv4.w;                                     \
    }                                                                                                               \
    kmv = a.km ? (int)a.km[rowm_] : 1;                                                                              \
  } while (0)
  if (have) V5_TILE_START(mt_first, r0_first);
  // Staged query-side rows AND (fp32 products) the weight slabs: every global load of both is issued before the first LDS store, so
  // the kernel's start-up pays ONE memory round trip for them instead of one before and one behind the node-side prologue (the slabs
  // live behind qd, outside the prologue's scratch: they may be filled before it runs): k_block_bwd_v5 96.9-97.0 -> 95.3-95.7 us (same box).
  BwdProRegs proR;
  {
    float4 sv[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = threadIdx.x + 256 * u, r = i / 40, f = i % 40;
      const bool ok = i < nl * 40 && !(a.pro && f >= 16 && f < 32);   // dV_att comes from the prologue below
      const size_t rowl = (size_t)b * N + l_begin + (ok ? r : 0);
      const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                       : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                                : a.stats + rowl * 32 + (f - 32) * 4;
      sv[u] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sA[G::TILES], sB[G::TILES], sD[G::TILES];
    if constexpr (MM == 0) {
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        const int i = threadIdx.x + 256 * t, ln = (i >> 2) & 63, u = i & 3, pp = ln & 15, qq = ln >> 4;
        const int c = 16 * t + 4 * qq + u;
        const int hd = 2 * (pp >> 2) + (pp & 1);
        sA[t] = a.pw[c * 16 + pp];
        sB[t] = ((pp & 2) == 0 && c < DE) ? a.Wr[hd * DE + c] : 0.f;
        sD[t] = a.pw[(16 * t + pp) * 16 + 4 * qq + u];
      }
    }
    if (a.pro) bwd_node_prologue_load<DE>(a, proR, b, l_begin);   // the prologue's own loads: issued before the first wait of the kernel
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = threadIdx.x + 256 * u, r = i / 40, f = i % 40;
      if (i < nl * 40 && !(a.pro && f >= 16 && f < 32)) {
        float4 v = sv[u];
        if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
        *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
      }
    }
    if constexpr (MM == 0) {
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        const int i = threadIdx.x + 256 * t;
        wsA[i] = sA[t]; wsB[i] = sB[t]; wsD[i] = sD[t];
      }
    }
  }
  PSTAMP(0);
  if (a.pro) {
    __syncthreads();
    bwd_node_prologue_finish<DE>(a, pro_ws, qd, b, l_begin, wg, proR);
  }
  PSTAMP(1);
  // weight slabs: element (t, lane, u)
  if constexpr (MM != 0) {
    // bf16 operands.  wsA / wsB: [step s][part hi|lo][lane][8 slots], slot i <-> channel 16 (2s + (i >> 2)) + 4q + (i & 3);
    // wsD: [tile t][lane][hi(4) | lo(4)] of the lane's 4 dGE columns 4q + u
    uint16_t* A16 = reinterpret_cast<uint16_t*>(wsA);
    uint16_t* B16 = reinterpret_cast<uint16_t*>(wsB);
    uint16_t* D16 = reinterpret_cast<uint16_t*>(wsD);
    auto parts = [](float v, uint16_t& hi, uint16_t& lo) {
      const uint32_t h = pk_bf16(v, 0.f) & 0xFFFFu;
      hi = (uint16_t)h;
      lo = (uint16_t)(pk_bf16(v - __uint_as_float(h << 16), 0.f) & 0xFFFFu);
    };
    for (int idx = threadIdx.x; idx < NS * 512; idx += 256) {
      const int i = idx & 7, ln = (idx >> 3) & 63, s_ = idx >> 9, pp = ln & 15, qq = ln >> 4, t = 2 * s_ + (i >> 2);
      const int c = 16 * t + 4 * qq + (i & 3);
      const bool in = t < G::TILES && c < DE;
      const int hd = 2 * (pp >> 2) + (pp & 1);
      uint16_t hi, lo;
      parts(in ? a.pw[c * 16 + pp] : 0.f, hi, lo);
      A16[((s_ * 2 + 0) * 64 + ln) * 8 + i] = hi; A16[((s_ * 2 + 1) * 64 + ln) * 8 + i] = lo;
      parts((in && (pp & 2) == 0) ? a.Wr[hd * DE + c] : 0.f, hi, lo);
      B16[((s_ * 2 + 0) * 64 + ln) * 8 + i] = hi; B16[((s_ * 2 + 1) * 64 + ln) * 8 + i] = lo;
    }
    for (int idx = threadIdx.x; idx < G::TILES * 256; idx += 256) {
      const int u = idx & 3, ln = (idx >> 2) & 63, t = idx >> 8, pp = ln & 15, qq = ln >> 4;
      uint16_t hi, lo;
      parts(a.pw[(16 * t + pp) * 16 + 4 * qq + u], hi, lo);
      D16[(t * 64 + ln) * 8 + u] = hi; D16[(t * 64 + ln) * 8 + 4 + u] = lo;
    }
  }
  float c2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];

  v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // the prologue's scratch (the tile area behind the parity-0 e buffers) is dead from here
  PSTAMP(2);

  for (int mt = mt_first; have && mt <= mt_last; mt += mt_step) {
    const int r0 = (balance && mt == mt_first) ? r0_first : 0;            // rows [r0, r1) of the workgroup's nl
    const int r1 = (balance && mt == mt_last) ? t1 - mt * nl : nl;
    const int m0 = mt * 16, m = m0 + p;
    const int kv = RAG ? min(16, N - m0) : 16;          // valid keys of the tile (wave-uniform)
    const bool kvalid = RAG ? (p < kv) : true;
    // a later key tile of this wave (N > 64): its first e tile is in flight while K / V are fetched
    if (mt != mt_first) V5_TILE_START(mt, r0);
    float dKa[16], dVa[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
    asm volatile("" : "+v"(kmv));   // (the byte was requested long ago; its first use stays here)
    const float kadd = kmv == 0 ? -EGT_NEG : 0.0f;
    const int l_first = l_begin + r0, l_stop = l_begin + r1;
    for (int l = l_first; l < l_stop; ++l) {
      const int li = l - l_begin;      // row of the workgroup (qd)
      const int lj = l - l_first;      // row of this wave's tile segment (e buffer parity, first-row wait)
      const size_t rowl = (size_t)b * N + l;
      const size_t pair0 = rowl * N + m0;
      float* et = et0 + (lj & 1) * estr;
      MaskRegs mr{make_float2(1.f, 1.f), 0};
      // ---- de'(l): requested now, consumed after P1 ----
      TileRegs<DE> td;
      tile_gload<DE, EGT_NT_BWD_DY>(td, dey_in + pair0 * DE, lane, kv);
      // ---- e(l) has been in flight for a whole iteration (the first one: since kernel entry): retire it.  Younger operations of
      // this wave: row l-1's dQ-partial store and its NI de stores (none before the first row), then the NI de' loads just issued ----
      if (lj == 0) vm_wait<(G::NF4 + 63) / 64>(); else vm_wait<2 * ((G::NF4 + 63) / 64) + 1>();
      SCHED_FENCE();
      // ---- P1: norm_edge, projections (recompute) ----
      float rstd;
      v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
      {
        float4 x[G::TILES];
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(et, p, q, t);
        rstd = ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
        if constexpr (MM != 0) {
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) frag_write<DE>(et, p, q, t, x[t]);     // xhat stays in the tile for the later phases
          v4f xv[G::TILES];
          Bf8 xh[NS], xl[NS];
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) xv[t] = (v4f){x[t].x, x[t].y, x[t].z, x[t].w};
          split_tiles<G::TILES, SPLIT>(xv, xh, xl);
          acc = bf_gemm<NS, SPLIT>(wsA, 0, lane, xh, xl, acc);
        } else {
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          frag_write<DE>(et, p, q, t, x[t]);     // xhat stays in the tile for the later phases
          const float4 w = *reinterpret_cast<const float4*>(wsA + (t * 64 + lane) * 4);
          acc = MFMA(w.x, x[t].x, acc);
          acc = MFMA(w.y, x[t].y, acc);
          acc = MFMA(w.z, x[t].z, acc);
          acc = MFMA(w.w, x[t].w, acc);
        }
        }
      }
      SCHED_FENCE();
      tile_lds_put<DE>(dt, td, lane, kv);   // (the compiler's own vmcnt wait for de' sits here; rows past kv are zero-filled)
      lds_sync();
      // ---- e(l+1) -> the other e buffer (its last reader, row l-1's P5, retired its LDS reads) ----
      if (l + 1 < l_stop) {
        if (RAG && kv < 16) tile_dma_ragged<DE>(et_lds + (unsigned)(((lj + 1) & 1) * estr * 4), e_in + (pair0 + (size_t)N) * DE, lane, kv);
        else tile_dma<DE, EGT_NT_BWD_E>(et_lds + (unsigned)(((lj + 1) & 1) * estr * 4), e_in + (pair0 + (size_t)N) * DE, off0);
      }
      SCHED_FENCE();
      // ---- P2: dH_ext = de'.Wr^T ----
      v4f dhx = {0.f, 0.f, 0.f, 0.f};
      if (!(a.guard & 8)) {
      if constexpr (MM != 0) {
        v4f dv[G::TILES];
        Bf8 dh_[NS], dl_[NS];
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) { const float4 d4 = frag_read<DE>(dt, p, q, t); dv[t] = (v4f){d4.x, d4.y, d4.z, d4.w}; }
        split_tiles<G::TILES, SPLIT>(dv, dh_, dl_);
        dhx = bf_gemm<NS, SPLIT>(wsB, 0, lane, dh_, dl_, dhx);
      } else {
#pragma unroll
      for (int t = 0; t < G::TILES; ++t) {
        const float4 dyv = frag_read<DE>(dt, p, q, t);
        const float4 w = *reinterpret_cast<const float4*>(wsB + (t * 64 + lane) * 4);
        dhx = MFMA(w.x, dyv.x, dhx);
        dhx = MFMA(w.y, dyv.y, dhx);
        dhx = MFMA(w.z, dyv.z, dhx);
        dhx = MFMA(w.w, dyv.w, dhx);
      }
      }
      }
      SCHED_FENCE();
      // ---- P3: logits, softmax/gate backward, dQ/dK/dV ----
      float dge[4], hh[2], dA[2], at[2];
      {
        const float* qr = qd + li * QD_LD;
        float dots[2], dAd[2];
        {
          const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
          const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
          float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 u = qp[i], v = dp[i];
            d0 = fmaf(u.x, Kf[4*i], d0);   d1 = fmaf(u.y, Kf[4*i+1], d1);
            d0 = fmaf(u.z, Kf[4*i+2], d0); d1 = fmaf(u.w, Kf[4*i+3], d1);
            e0 = fmaf(v.x, Vf[4*i], e0);   e1 = fmaf(v.y, Vf[4*i+1], e1);
            e0 = fmaf(v.z, Vf[4*i+2], e0); e1 = fmaf(v.w, Vf[4*i+3], e1);
          }
          dots[0] = d0; dots[1] = d1; dAd[0] = e0; dAd[1] = e1;
        }
        const float4* sp = reinterpret_cast<const float4*>(qr + 128 + q * 8);
        const float4 s0 = sp[0], s1 = sp[1];
        const float st[8] = {s0.x, s0.y, s0.z, 0.f, s1.x, s1.y, s1.z, 0.f};
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
        if (RAG && !kvalid) { xl[0] = xl[1] = -3.0e38f; gl[0] = gl[1] = -3.0e38f; }   // a key past N: S = 0, gate = 0 (exactly)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float S = __expf(xl[j] - st[4 * j]) * st[4 * j + 1];
          const float g = (gated || (RAG && !kvalid)) ? egt_sigmoid(gl[j]) : 1.0f;
          const float dS = dAd[j] * g;
          const float dGl = gated ? dAd[j] * S * g * (1.0f - g) : 0.f;
          const float dH = S * (dS - st[4 * j + 2]) + dhx[j];
          dA[j] = dH * inr[j] * a.scale;
          at[j] = S * g;
          dge[2 * j] = dGl;
          dge[2 * j + 1] = dH;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[r] += dge[r];
      *reinterpret_cast<float4*>(sc1 + p * 16 + 4 * q) = make_float4(dge[0], dge[1], dge[2], dge[3]);
      *reinterpret_cast<float2*>(sc2 + p * 12 + 2 * q) = make_float2(hh[0], hh[1]);
      if (q == 0) sc2[p * 12 + 8] = 1.0f;
      lds_sync();
      SCHED_FENCE();
      if (!(a.guard & 4))
      {
        const float* qr = qd + li * QD_LD;
        const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
        const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
        float dq[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 u = qp[i], v = dp[i];
          dKa[4*i]   = fmaf(dA[0], u.x, dKa[4*i]);   dKa[4*i+1] = fmaf(dA[1], u.y, dKa[4*i+1]);
          dKa[4*i+2] = fmaf(dA[0], u.z, dKa[4*i+2]); dKa[4*i+3] = fmaf(dA[1], u.w, dKa[4*i+3]);
          dVa[4*i]   = fmaf(at[0], v.x, dVa[4*i]);   dVa[4*i+1] = fmaf(at[1], v.y, dVa[4*i+1]);
          dVa[4*i+2] = fmaf(at[0], v.z, dVa[4*i+2]); dVa[4*i+3] = fmaf(at[1], v.w, dVa[4*i+3]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { dq[2 * k] = dA[0] * Kf[2 * k]; dq[2 * k + 1] = dA[1] * Kf[2 * k + 1]; }
        // dQ[l] partial over this tile's 16 keys -> HBM, summed over key tiles by the next prologue (or k_node_bwd)
        a.dqp[(((size_t)b * ntile + mt) * N + l) * 64 + lane] = reduce16_keep_own(dq, p);
      }
      SCHED_FENCE();
      // ---- P4: weight-gradient contractions over the 16 pairs of the tile ----
      if (!(a.guard & 1))
      {
        float bT[4], bR[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bT[s] = sc1[(q + 4 * s) * 16 + p];
          bR[s] = (p < 9) ? sc2[(q + 4 * s) * 12 + p] : 0.f;
        }
        {
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
      }
      lds_sync();
      SCHED_FENCE();
      // ---- P5: d(ehat) = Wp . dGE, LayerNorm backward, de = de' + ... straight to HBM ----
      if (!(a.guard & 2)) {
        float4 dxh[G::TILES];
        float m1 = 0.f, m2 = 0.f;
        Bf8 gB1, gB2;   // B operands of the stacked K = 16 product: [dGE_hi | dGE_hi] and [dGE_lo | 0]
        if constexpr (MM != 0) {
          const uint32_t h0 = pk_bf16(dge[0], dge[1]), h1 = pk_bf16(dge[2], dge[3]);
          gB1.u[0] = h0; gB1.u[1] = h1; gB1.u[2] = h0; gB1.u[3] = h1;
          gB2.u[0] = pk_bf16(dge[0] - __uint_as_float(h0 << 16), dge[1] - __uint_as_float(h0 & 0xFFFF0000u));
          gB2.u[1] = pk_bf16(dge[2] - __uint_as_float(h1 << 16), dge[3] - __uint_as_float(h1 & 0xFFFF0000u));
          gB2.u[2] = 0u; gB2.u[3] = 0u;
        }
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 xh = frag_read<DE>(et, p, q, t);
          v4f d = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MM != 0) {
            Bf8 wa, wb;   // [W_hi(4) | W_lo(4)] . [d_hi | d_hi] = W_hi.d_hi + W_lo.d_hi ;  [W_hi(4) | 0] . [d_lo | 0]
            wa.q = *reinterpret_cast<const uint4*>(wsD + (t * 64 + lane) * 4);
            d = MFMA_BF(wa.v, gB1.v, d);
            if (SPLIT) {
              wb.u[0] = wa.u[0]; wb.u[1] = wa.u[1]; wb.u[2] = 0u; wb.u[3] = 0u;
              d = MFMA_BF(wb.v, gB2.v, d);
            }
          } else {
          const float4 w = *reinterpret_cast<const float4*>(wsD + (t * 64 + lane) * 4);
          d = MFMA(w.x, dge[0], d);
          d = MFMA(w.y, dge[1], d);
          d = MFMA(w.z, dge[2], d);
          d = MFMA(w.w, dge[3], d);
          }
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
            if (kvalid) { if (EGT_NT_BWD_ST) egt_st4_nt(orow + 16 * t, o); else *reinterpret_cast<float4*>(orow + 16 * t) = o; }
          }
        }
        lds_sync();   // the tile reads above retire before the next iteration overwrites dt / DMAs into et
      }
      SCHED_FENCE();
    }
    float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
    float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
    if (balance && r1 < nl && r0 == 0) {   // the tile's last rows were done by the next wave, at the START of its range: its partial sits in the slot
      while (pflag[wave + 1] == 0) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (kvalid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 kk = ko[i], vv = vo[i];
          dKa[4*i] += kk.x; dKa[4*i+1] += kk.y; dKa[4*i+2] += kk.z; dKa[4*i+3] += kk.w;
          dVa[4*i] += vv.x; dVa[4*i+1] += vv.y; dVa[4*i+2] += vv.z; dVa[4*i+3] += vv.w;
        }
      }
    }
    if (kvalid) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
        vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
      }
    }
    if (balance && r0 > 0) {   // the tile's first rows belong to the previous wave: what was just stored is this wave's partial, parked in the
                               // slot (same CU: the stores are acknowledged by the L2 before the flag goes up, the slot was never in this L1)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) pflag[wave] = 1;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) ssum[r] = row_sum16(ssum[r]);
  __syncthreads();
  float* ep = sm + wave * G::EP;
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
  float* out = a.epart + (size_t)wg * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 256)
    out[i] = (sm[i] + sm[G::EP + i]) + (sm[2 * G::EP + i] + sm[3 * G::EP + i]);
}

// ---------------------------------------------------------------- backward, narrow edge channels ---
// k_block_bwd_v4 with the row loop unrolled by R (De <= 16, N % 16 == 0, no mask tensors): one
// iteration = the wave's key tile x R query rows.  The rows' P1..P5 chains are independent, the LDS
// hand-offs are shared (5 per R rows instead of 5 per row), the e / de' tiles of the next R rows are
// in flight during the arithmetic.  Same arguments, partial layouts and prologue as v4.
template <int DE, bool BF, int R>
__global__ void __launch_bounds__(256, 2) k_block_bwd_v4r(BlockArgs a) {   // R rows per iteration
  seed_from_device(a);
  using G = Geo<DE>;
  typedef typename EdgeT<BF>::type ET;
  const ET* e_in = reinterpret_cast<const ET*>(a.e);
  const ET* dey_in = reinterpret_cast<const ET*>(a.de_out);
  ET* dex_o = reinterpret_cast<ET*>(a.de);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int b, lr;
  egt_group_order(wg, a.B, a.NLR, N, b, lr, TL);
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL), nl = l_end - l_begin;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  constexpr int TF = G::TILE_FLOATS;
  constexpr int PWR = R * (2 * TF + 256 + 192);   // per wave: R rows x (xhat tile, de' tile, dGE, H_hat)
  constexpr int WSLAB = G::TILES * 256;
  float* et0 = sm + wave * PWR;        // [R][TF]
  float* dt0 = et0 + R * TF;           // [R][TF]
  float* sc10 = dt0 + R * TF;          // [R][256]
  float* sc20 = sc10 + R * 256;        // [R][192]
  constexpr int AREA = 4 * PWR > BWD_PRO_WS ? 4 * PWR : BWD_PRO_WS;
  static_assert(4 * G::EP <= AREA, "edge partial staging must fit the LDS tile area");
  float* qd = sm + AREA;               // [TL][QD_LD]
  float* wsA = qd + TL * QD_LD;
  float* wsB = wsA + WSLAB;
  float* wsD = wsB + WSLAB;
  float* park = wsD + WSLAB;           // [waves 1..3][32][64]: dK / dV of a key tile shared with the previous wave (balanced ranges)
  volatile int* pflag = reinterpret_cast<volatile int*>(park + 3 * 2048);
  if (threadIdx.x < 4) pflag[threadIdx.x] = 0;
  bwd_stage_rows<256, 3>(a, qd, b, l_begin, nl);
  if (a.pro) {
    __syncthreads();
    bwd_node_prologue<DE>(a, sm, qd, b, l_begin, wg);
  }
  for (int i = threadIdx.x; i < G::TILES * 256; i += 256) {
    const int t = i >> 8, ln = (i >> 2) & 63, u = i & 3, pp = ln & 15, qq = ln >> 4;
    const int c = 16 * t + 4 * qq + u;
    wsA[i] = a.pw[c * 16 + pp];
    const int hd = 2 * (pp >> 2) + (pp & 1);
    wsB[i] = ((pp & 2) == 0 && c < DE) ? a.Wr[hd * DE + c] : 0.f;
    wsD[i] = a.pw[(16 * t + pp) * 16 + 4 * qq + u];
  }
  float c2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];
  v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // Ragged N: the last key tile has kv < 16 valid keys (loads clamped / zero-filled, the lanes of the
  // missing keys get probability and gate exactly 0), the last row group has nl < 16 rows (skipped).
  const int ntile = (N + 15) / 16;
  // Work of a wave: key tiles w, w+4, ... when the tile count is a multiple of 4 (or < 4); otherwise the ntile x (row pairs)
  // steps are cut into four contiguous equal ranges and a tile that straddles two ranges is shared by neighbouring waves
  // (the later wave meets it first and parks its dK / dV partial in LDS, the earlier one meets it last and adds it) -- as in
  // k_narrow_bwd.
  const int npair = (nl + R - 1) / R;
#ifdef EGT_V4R_NO_BALANCE
  const bool balance = false;
#else
  const bool balance = ntile >= 4 && (ntile & 3) != 0;
#endif
  const int TT = ntile * npair;
  const int t0 = balance ? (wave * TT) >> 2 : 0, t1 = balance ? ((wave + 1) * TT) >> 2 : 0;
  const int mt_first = balance ? t0 / npair : wave, mt_last = balance ? (t1 - 1) / npair : ntile - 1, mt_step = balance ? 1 : 4;
  for (int mt = mt_first; mt <= mt_last; mt += mt_step) {
    const int q0 = (balance && mt == mt_first) ? t0 - mt * npair : 0;          // row pairs [q0, q1) of the workgroup's npair
    const int q1 = (balance && mt == mt_last) ? t1 - mt * npair : npair;
    const int m0 = mt * 16, m = m0 + p, kv = min(16, N - m0);
    const bool kvalid = m < N;
    float Kf[16], Vf[16], dKa[16], dVa[16];
    const size_t rowm = (size_t)b * N + (kvalid ? m : N - 1);
    {
      const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
      const float4* vp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 128 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i], vv = vp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
        Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
    }
    const float kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;
    const MaskRegs mr{make_float2(1.f, 1.f), 0};
    TileRegs<DE> te[R], td[R];
    auto prefetch = [&](int lq) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const size_t pair0 = ((size_t)b * N + l_begin + min(R * lq + i, nl - 1)) * N + m0;
        tile_gload<DE>(te[i], e_in + pair0 * DE, lane, kv);
        tile_gload<DE>(td[i], dey_in + pair0 * DE, lane, kv);
      }
    };
    prefetch(q0);
    for (int lq = q0; lq < q1; ++lq) {
      const int lb = l_begin + R * lq;
      const int nr = min(R, nl - R * lq);   // rows of this step that exist
      size_t pair0[R];
#pragma unroll
      for (int i = 0; i < R; ++i) pair0[i] = ((size_t)b * N + min(lb + i, l_end - 1)) * N + m0;
      lds_sync();   // the de tiles of the previous four rows have left the LDS tiles
#pragma unroll
      for (int i = 0; i < R; ++i) {
        tile_lds_put<DE>(et0 + i * TF, te[i], lane, kv);
        tile_lds_put<DE>(dt0 + i * TF, td[i], lane, kv);
      }
      if (lq + 1 < q1) prefetch(lq + 1);
      lds_sync();
      // ---- P1: norm_edge, projections (recompute) ; P2: dH_ext = de'.Wr^T ----
      float rstd[R];
      v4f acc[R], dhx[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (i >= nr) continue;
        float* et = et0 + i * TF;
        const float* dt = dt0 + i * TF;
        acc[i] = (v4f){c2r[0], c2r[1], c2r[2], c2r[3]};
        dhx[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        float4 x[G::TILES];
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(et, p, q, t);
        rstd[i] = ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          frag_write<DE>(et, p, q, t, x[t]);     // xhat stays in the tile for the later phases
          const float4 w = *reinterpret_cast<const float4*>(wsA + (t * 64 + lane) * 4);
          acc[i] = MFMA(w.x, x[t].x, acc[i]);
          acc[i] = MFMA(w.y, x[t].y, acc[i]);
          acc[i] = MFMA(w.z, x[t].z, acc[i]);
          acc[i] = MFMA(w.w, x[t].w, acc[i]);
          const float4 dyv = frag_read<DE>(dt, p, q, t);
          const float4 wb = *reinterpret_cast<const float4*>(wsB + (t * 64 + lane) * 4);
          dhx[i] = MFMA(wb.x, dyv.x, dhx[i]);
          dhx[i] = MFMA(wb.y, dyv.y, dhx[i]);
          dhx[i] = MFMA(wb.z, dyv.z, dhx[i]);
          dhx[i] = MFMA(wb.w, dyv.w, dhx[i]);
        }
      }
      // ---- P3: logits, softmax/gate backward ----
      float dge[R][4], dA[R][2], at[R][2];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (i >= nr) continue;
        const float* qr = qd + (R * lq + i) * QD_LD;
        float dots[2], dAd[2], hh[2];
        {
          const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
          const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
          float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int u4 = 0; u4 < 4; ++u4) {
            const float4 u = qp[u4], v = dp[u4];
            d0 = fmaf(u.x, Kf[4*u4], d0);   d1 = fmaf(u.y, Kf[4*u4+1], d1);
            d0 = fmaf(u.z, Kf[4*u4+2], d0); d1 = fmaf(u.w, Kf[4*u4+3], d1);
            e0 = fmaf(v.x, Vf[4*u4], e0);   e1 = fmaf(v.y, Vf[4*u4+1], e1);
            e0 = fmaf(v.z, Vf[4*u4+2], e0); e1 = fmaf(v.w, Vf[4*u4+3], e1);
          }
          dots[0] = d0; dots[1] = d1; dAd[0] = e0; dAd[1] = e1;
        }
        const float4* sp = reinterpret_cast<const float4*>(qr + 128 + q * 8);
        const float4 s0 = sp[0], s1 = sp[1];
        const float st[8] = {s0.x, s0.y, s0.z, 0.f, s1.x, s1.y, s1.z, 0.f};
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
          hh[j] = ah + acc[i][2 * j + 1];
          xl[j] = hh[j];
          gl[j] = acc[i][2 * j];
        }
        apply_masks<false>(a, kadd, mr, (pair0[i] + p) * BH, q, xl, gl);
        if (!kvalid) { xl[0] = xl[1] = -3.0e38f; gl[0] = gl[1] = -3.0e38f; }   // a key past N: S = 0, gate = 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float S = __expf(xl[j] - st[4 * j]) * st[4 * j + 1];
          const float g = gated ? egt_sigmoid(gl[j]) : 1.0f;
          const float dS = dAd[j] * g;
          const float dGl = gated ? dAd[j] * S * g * (1.0f - g) : 0.f;
          const float dH = S * (dS - st[4 * j + 2]) + dhx[i][j];
          dA[i][j] = dH * inr[j] * a.scale;
          at[i][j] = S * g;
          dge[i][2 * j] = dGl;
          dge[i][2 * j + 1] = dH;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[r] += dge[i][r];
        float* sc1 = sc10 + i * 256;
        float* sc2 = sc20 + i * 192;
        *reinterpret_cast<float4*>(sc1 + p * 16 + 4 * q) = make_float4(dge[i][0], dge[i][1], dge[i][2], dge[i][3]);
        *reinterpret_cast<float2*>(sc2 + p * 12 + 2 * q) = make_float2(hh[0], hh[1]);
        if (q == 0) sc2[p * 12 + 8] = 1.0f;
        SCHED_FENCE();   // one row's Q / dV_att fragments (32 registers) at a time
      }
      lds_sync();
      // ---- dK / dV accumulation, dQ partials ; P4: weight-gradient contractions ----
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (i >= nr) continue;
        const float* qr = qd + (R * lq + i) * QD_LD;
        const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
        const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
        float dq[16];
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
          const float4 u = qp[u4], v = dp[u4];
          dKa[4*u4]   = fmaf(dA[i][0], u.x, dKa[4*u4]);   dKa[4*u4+1] = fmaf(dA[i][1], u.y, dKa[4*u4+1]);
          dKa[4*u4+2] = fmaf(dA[i][0], u.z, dKa[4*u4+2]); dKa[4*u4+3] = fmaf(dA[i][1], u.w, dKa[4*u4+3]);
          dVa[4*u4]   = fmaf(at[i][0], v.x, dVa[4*u4]);   dVa[4*u4+1] = fmaf(at[i][1], v.y, dVa[4*u4+1]);
          dVa[4*u4+2] = fmaf(at[i][0], v.z, dVa[4*u4+2]); dVa[4*u4+3] = fmaf(at[i][1], v.w, dVa[4*u4+3]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { dq[2 * k] = dA[i][0] * Kf[2 * k]; dq[2 * k + 1] = dA[i][1] * Kf[2 * k + 1]; }
        a.dqp[(((size_t)b * ntile + mt) * N + lb + i) * 64 + lane] = reduce16_keep_own(dq, p);
        const float* et = et0 + i * TF;
        const float* dt = dt0 + i * TF;
        const float* sc1 = sc10 + i * 256;
        const float* sc2 = sc20 + i * 192;
        float bT[4], bR[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          bT[s4] = sc1[(q + 4 * s4) * 16 + p];
          bR[s4] = (p < 9) ? sc2[(q + 4 * s4) * 12 + p] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < G::TILES; ++t)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            accT[t] = MFMA(elem_read<DE>(et, q + 4 * s4, 16 * t + p), bT[s4], accT[t]);
            accR[t] = MFMA(elem_read<DE>(dt, q + 4 * s4, 16 * t + p), bR[s4], accR[t]);
          }
        SCHED_FENCE();
      }
      lds_sync();
      // ---- P5: d(ehat) = Wp . dGE, LayerNorm backward, de = de' + ... in place over the de' tiles ----
      float4 dxh[R][G::TILES];
      float m1[R], m2[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (i >= nr) continue;
        const float* et = et0 + i * TF;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 w = *reinterpret_cast<const float4*>(wsD + (t * 64 + lane) * 4);
          const float4 xh = frag_read<DE>(et, p, q, t);
          v4f d = {0.f, 0.f, 0.f, 0.f};
          d = MFMA(w.x, dge[i][0], d);
          d = MFMA(w.y, dge[i][1], d);
          d = MFMA(w.z, dge[i][2], d);
          d = MFMA(w.w, dge[i][3], d);
          dxh[i][t] = make_float4(d[0], d[1], d[2], d[3]);
          s1 += (d[0] + d[1]) + (d[2] + d[3]);
          s2 = fmaf(d[0], xh.x, s2); s2 = fmaf(d[1], xh.y, s2);
          s2 = fmaf(d[2], xh.z, s2); s2 = fmaf(d[3], xh.w, s2);
        }
        m1[i] = sum_over_q(s1) * (1.0f / DE);
        m2[i] = sum_over_q(s2) * (1.0f / DE);
        if (a.flags & EGT_BF_NO_EDGE_LN) { m1[i] = 0.f; m2[i] = 0.f; }   // no norm_edge: d e = de' + d(proj input)
      }
      lds_sync();
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (i >= nr) continue;
        const float* et = et0 + i * TF;
        float* dt = dt0 + i * TF;
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
          const float4 dyv = frag_read<DE>(dt, p, q, t);
          const float4 xh = frag_read<DE>(et, p, q, t);
          float4 o;
          o.x = dyv.x + rstd[i] * (dxh[i][t].x - m1[i] - xh.x * m2[i]);
          o.y = dyv.y + rstd[i] * (dxh[i][t].y - m1[i] - xh.y * m2[i]);
          o.z = dyv.z + rstd[i] * (dxh[i][t].z - m1[i] - xh.z * m2[i]);
          o.w = dyv.w + rstd[i] * (dxh[i][t].w - m1[i] - xh.w * m2[i]);
          frag_write<DE>(dt, p, q, t, o);
        }
      }
      lds_sync();   // stream out the four de tiles
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (i < nr) tile_from_lds<DE>(dt0 + i * TF, dex_o + pair0[i] * DE, lane, kv);
    }
    if (q0 > 0) {   // the tile's first rows belong to the previous wave: park this partial for it
      float* pk = park + (wave - 1) * 2048 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) { pk[i * 64] = dKa[i]; pk[(16 + i) * 64] = dVa[i]; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) pflag[wave] = 1;
    } else {
      if (q1 < npair) {   // the tile's last rows were done by the next wave at the very start of its range
        while (pflag[wave + 1] == 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const float* pk = park + wave * 2048 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) { dKa[i] += pk[i * 64]; dVa[i] += pk[(16 + i) * 64]; }
      }
      float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
      float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
      if (kvalid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
          vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) ssum[r] = row_sum16(ssum[r]);
  __syncthreads();
  float* ep = sm + wave * G::EP;
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
  float* out = a.epart + (size_t)wg * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 256)
    out[i] = (sm[i] + sm[G::EP + i]) + (sm[2 * G::EP + i] + sm[3 * G::EP + i]);
}

