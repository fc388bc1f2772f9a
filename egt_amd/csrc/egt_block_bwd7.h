// k_block_bwd_v7: the fused backward of the attention block at THREE wavefronts per SIMD (included by egt_block.hip).
// OPT-IN (EGT_BWD_V7 = 1 | 2, egt_block.hip: layout()).  Built in round 5 as the experiment VERDICT r4 item 1 asked for ("a third
// resident workgroup per CU"); measured at the headline batch it takes 104.7 us per launch against k_block_bwd_v5's 98.6 us
// (profiles/r05_v7_*): the third wave per SIMD does not pay.  Staggered wave starts and static wave priorities change nothing
// (104.5 - 107 us), i.e. the waves are not phase-locked on their tile requests: v5 already sits at ~77 % of its instruction-issue
// bound and ~84 % of the sustainable HBM rate inside its row loop (profiles/r03_mfma_valu_issue.md), and this kernel pays ~15 % more
// VALU instructions per step for the address arithmetic that fits it into 168 registers.  Kept as a tested alternative.
//
// k_block_bwd_v5 runs two workgroups of four waves per CU (218 VGPRs, 77 KB of LDS each).  A third four-wave workgroup per CU would
// need <= 53 KB of LDS and 50 % more workgroups on shorter row groups, whose per-workgroup costs (node-side prologue, weight slabs,
// K / V, partial sums) do not shrink with the rows.  This kernel gets the third wave per SIMD the other way round:
//  * ONE workgroup of TWELVE waves per CU owns 32 query rows of a graph: the per-workgroup costs halve per row instead of growing
//    (one weight-slab fill, one edge-partial slot and one dK / dV partial per 32 rows; the prologue still runs per 16 rows, two
//    groups of four waves side by side, the remaining four waves only keep its barriers);
//  * a wave owns (key tile, row chunk): the 12 / ntile waves of a key tile split the workgroup's rows into equal contiguous chunks,
//    so no two waves ever touch the same LDS tile (no barrier inside the row loop); the chunks' dK / dV accumulators are added
//    through LDS after the loop;
//  * 160 VGPRs, 0 spills in the loop (v5: 218).  What it took: BOTH streamed tiles of a row, e and de', arrive by LDS-DMA (no
//    register staging), one buffer each -- the request for row l+1 goes out as soon as P5 has the row's xhat / de' fragments in
//    registers, ahead of its 16 MFMAs and of the de stores; V of the graph's keys lives in LDS (16 KB, XOR-swizzled pieces) instead
//    of 16 registers per lane; the bias-gradient sums ride on P4's B operands (1 register instead of 4); every LDS / global address
//    is derived inside its phase from nine lane constants (hipcc otherwise carries 39 address registers through the loop:
//    tools/isa_pressure.py shows the live ranges);
//  * LDS: 12 x (e tile + de' tile + dGE + H_hat) = 114 KB, 32 staged node rows 20 KB, weight slabs 10 KB (the dH_ext slab without
//    its zero rows), V 16 KB = exactly the 160 KB of a CU.
// Geometry: fp32 edge tensors of 64 channels, no mask tensors, N = 32 or 64.  A row step is k_block_bwd_v5's arithmetic; the sums
// over row chunks (dK, dV, the weight-gradient partials), the order of the dV_att pieces and the bias-gradient sums associate
// differently: equal to v5 within a few ulps, not bit for bit (tests/test_bwd_v7_gpu.py).
#pragma once

#define V7_WAVES 12
#define V7_ROWS 32
constexpr int cmax(int x, int y) { return x > y ? x : y; }

// LDS accesses by byte address (native vector types: the HIP float4 class does not bind to address-space pointers)
typedef float v2f __attribute__((ext_vector_type(2)));
#define V7_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ float4 v7_ld4(unsigned addr) {
  const v4f v = *reinterpret_cast<const V7_AS3 v4f*>((size_t)addr);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float v7_ld1(unsigned addr) { return *reinterpret_cast<const V7_AS3 float*>((size_t)addr); }
__device__ __forceinline__ void v7_st4(unsigned addr, float4 v) { *reinterpret_cast<V7_AS3 v4f*>((size_t)addr) = (v4f){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void v7_st2(unsigned addr, float2 v) { *reinterpret_cast<V7_AS3 v2f*>((size_t)addr) = (v2f){v.x, v.y}; }

template <int DE>
__global__ void __launch_bounds__(64 * V7_WAVES) k_block_bwd_v7(BlockArgs a) {
  seed_from_device(a);
  using G = Geo<DE>;
  constexpr int NI = G::NF4 / 64;            // LDS-DMA instructions (= 16-byte stores per lane) of one tile
  const float* e_in = a.e;
  const float* dey_in = a.de_out;
  float* dex_o = a.de;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / a.NLR, lr = wg % a.NLR;
  const int l_begin = lr * V7_ROWS, l_end = min(N, l_begin + V7_ROWS), nl = l_end - l_begin;
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  constexpr int PW = 2 * G::TILE_FLOATS + 256 + 128;   // per wave: e / xhat tile, de' tile, dGE [16][16], H_hat [16][8]
  constexpr int WSLAB = G::TILES * 256;                // one weight slab: [TILES][64 lanes] float4
  // the tile area is also the prologue's scratch (two groups), the dK / dV merge buffer and the edge partials' staging
  constexpr int AREA = cmax(cmax(V7_WAVES * PW, 3 * BWD_PRO_WS), cmax(V7_WAVES * 2048, V7_WAVES * G::EP));
  float* et = sm + wave * PW;
  float* dt = et + G::TILE_FLOATS;
  float* sc1 = dt + G::TILE_FLOATS;
  float* sc2 = sc1 + 256;
  float* qd = sm + AREA;                     // [V7_ROWS][QD_LD]
  float* wsA = qd + V7_ROWS * QD_LD;         // projection weights   wA[4t+u]
  float* wsB = wsA + WSLAB;                  // dH_ext weights       wrB[4t+u]
  float* wsD = wsB + WSLAB / 2;              // d(ehat) weights      wD[t][s]
  float* vl = wsD + WSLAB;                   // V of the graph's keys: [4 head pairs][N keys][16], 16-byte pieces XOR-swizzled with (key >> 2) & 3
  const unsigned et_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)et);
  const unsigned dt_lds = et_lds + (unsigned)(G::TILE_FLOATS * 4);
  const unsigned off0 = dma_lane_offset<DE>(lane);
  bwd_stage_rows<64 * V7_WAVES, 2>(a, qd, b, l_begin, nl);
  if (a.pro) {
    __syncthreads();
    // two 16-row groups of four waves side by side, each on its own scratch; waves 8..11 only keep the barriers (pnv = 0: every load
    // clamped to a valid row, nothing stored outside their own scratch)
    const int grp = wave >> 2;
    const int g2 = grp < 2 ? grp : 0;
    const int gl = l_begin + 16 * g2;
    const int gnv = grp < 2 ? max(0, min(16, l_end - gl)) : 0;
    bwd_node_prologue<DE>(a, sm + grp * BWD_PRO_WS, qd + 16 * g2 * QD_LD, b, min(gl, N - 1), 2 * wg + g2, nullptr, threadIdx.x & 255, gnv);
  }
  for (int i = threadIdx.x; i < G::TILES * 256; i += 64 * V7_WAVES) {   // weight slabs: element (t, lane, u)
    const int t = i >> 8, ln = (i >> 2) & 63, u = i & 3, pp = ln & 15, qq = ln >> 4;
    const int c = 16 * t + 4 * qq + u;
    wsA[i] = a.pw[c * 16 + pp];
    const int hd = 2 * (pp >> 2) + (pp & 1);
    // dH_ext rows r = 2, 3 of a lane's MFMA result are never read, so the A rows (pp & 2) != 0 need no zeros: every lane reads the
    // row of head 2 (pp >> 2) + (pp & 1) from a compact [t][8 heads][4 qq] slab (half the v5 slab)
    if ((pp & 2) == 0) wsB[((t * 8 + hd) * 4 + qq) * 4 + u] = c < DE ? a.Wr[hd * DE + c] : 0.f;
    wsD[i] = a.pw[(16 * t + pp) * 16 + 4 * qq + u];
  }
  for (int i = threadIdx.x; i < N * 16; i += 64 * V7_WAVES) {   // V rows of the graph (K stays in registers; V is read once per step)
    const int key = i >> 4, f = i & 15, qq = f >> 2, pc = f & 3;
    const float4 v = *reinterpret_cast<const float4*>(a.qkvp + ((size_t)b * N + key) * QKVP + 128 + qq * 16 + pc * 4);
    *reinterpret_cast<float4*>(vl + (qq * N + key) * 16 + 4 * (pc ^ ((key >> 2) & 3))) = v;
  }
  float c2r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];

  v4f accT[G::TILES], accR[G::TILES];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { accT[t] = (v4f){0.f, 0.f, 0.f, 0.f}; accR[t] = (v4f){0.f, 0.f, 0.f, 0.f}; }
  float ssum = 0.f;   // column p of dGE summed over the pairs q + 4s of every step (P4's B operands)
  __syncthreads();   // the prologue's scratch (= the tile area) is dead from here: DMA may land in it

  // this wave's (key tile, row chunk)
  const int ntile = N / 16, wpt = V7_WAVES / ntile;
  const int mt = wave / wpt, ch = wave - mt * wpt;
  const int r_lo = l_begin + (ch * nl) / wpt, r_hi = l_begin + ((ch + 1) * nl) / wpt;
  const int m0 = mt * 16, m = m0 + p;
  const int ph = (p >> 2) & 3;      // this lane's piece swizzle of the V rows (m0 is a multiple of 16)
  float Kf[16], dKa[16], dVa[16];   // dVa[4 i + c] belongs to channel 4 (i ^ ph) + c: V / dV_att pieces are read in swizzled order
#pragma unroll
  for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
  if (r_lo < r_hi) {
    const size_t pair_first = ((size_t)b * N + r_lo) * N + m0;
    tile_dma<DE>(et_lds, e_in + pair_first * DE, off0);
    tile_dma<DE>(dt_lds, dey_in + pair_first * DE, off0);
    const size_t rowm = (size_t)b * N + m;
    {
      const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
      }
    }
    const float kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;
    // ---- the loop's LDS / global addressing: a handful of lane constants ("roots", byte addresses); every other address is
    // derived from them INSIDE the phase that uses it (V7_OPQ hides a root from the optimiser at the head of a phase, so nothing
    // derived is hoisted out of the loop or carried across phases: hipcc otherwise keeps 39 address registers live through the
    // loop, which is what does not fit three waves per SIMD) ----
    const unsigned sm_b = (unsigned)(size_t)sm;
    unsigned aF = et_lds + (unsigned)(p * 256 + ((q ^ (p & 3)) << 4) + (ph << 6));    // fragment (p, q, t) of the e tile: aF ^ (t << 6); de' tile: + TILE bytes
    unsigned aT = et_lds + (unsigned)(4 * (64 * q + 4 * ((p >> 2) ^ q) + (p & 3)));  // transposed element (s, t): + 4 (256 s + 16 (t ^ s))
    unsigned aW = (unsigned)(size_t)wsA + (unsigned)lane * 16u;                       // weight slabs: wsA tile t + 1024 t; wsD behind wsB
    unsigned aWB = (unsigned)(size_t)wsB + (unsigned)(((2 * (p >> 2) + (p & 1)) * 4 + q) * 16);   // compact dH_ext slab: + 512 t
    unsigned aV = (unsigned)(size_t)vl + (unsigned)((q * N + m) * 64);               // the lane's V row: pieces + 16 i
    unsigned cDV = (unsigned)(256 + q * 64 + ph * 16);                               // dV_att pieces of a staged row: (row + cDV) ^ (i << 4)
    unsigned vp = (unsigned)p, vq = (unsigned)q;
    unsigned gOff = (unsigned)((p * DE + 4 * q) * 4);                                // the lane's bytes inside a de tile row block
    static_assert(DE == 64, "k_block_bwd_v7: the explicit tile addressing is the De = 64 swizzle");
    constexpr unsigned TB = G::TILE_FLOATS * 4;
    constexpr unsigned oB = WSLAB * 4, oD = oB + WSLAB * 2;   // wsB / wsD behind wsA (bytes)
    (void)sm_b; (void)oB;
    const unsigned qd_b = (unsigned)(size_t)qd, sc1_b = (unsigned)(size_t)sc1, sc2_b = (unsigned)(size_t)sc2;
#define V7_OPQ(x) asm volatile("" : "+v"(x))
#define LD4(addr) v7_ld4(addr)
#define LD1(addr) v7_ld1(addr)
#define ST4(addr, v) v7_st4((addr), (v))
#define ST2(addr, v) v7_st2((addr), (v))
    for (int l = r_lo; l < r_hi; ++l) {
      const unsigned rb = qd_b + (unsigned)((l - l_begin) * (QD_LD * 4));   // the staged node row (wave-uniform)
      const size_t pair0 = ((size_t)b * N + l) * N + m0;
      // ---- e(l), de'(l) have been in flight since the end of the previous step; the only younger operations of this wave are
      // that step's NI de stores (first step: the K / V loads, which the compiler's own wait has retired) ----
      if (l == r_lo) vm_wait<0>(); else vm_wait<NI>();
      SCHED_FENCE();
      // ---- P1: norm_edge, projections (recompute) ----
      float rstd;
      v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
      {
        V7_OPQ(aF); V7_OPQ(aW);
        const unsigned f0 = aF, f1 = aF ^ 64u, f2 = aF ^ 128u, f3 = aF ^ 192u;
        float4 x[4];
        x[0] = LD4(f0); x[1] = LD4(f1); x[2] = LD4(f2); x[3] = LD4(f3);
        rstd = ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
        ST4(f0, x[0]); ST4(f1, x[1]); ST4(f2, x[2]); ST4(f3, x[3]);   // xhat stays in the tile for the later phases
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 w = LD4(aW + 1024u * t);
          acc = MFMA(w.x, x[t].x, acc);
          acc = MFMA(w.y, x[t].y, acc);
          acc = MFMA(w.z, x[t].z, acc);
          acc = MFMA(w.w, x[t].w, acc);
        }
      }
      SCHED_FENCE();
      // ---- P2: dH_ext = de'.Wr^T ----
      v4f dhx = {0.f, 0.f, 0.f, 0.f};
      if (!(a.guard & 8)) {   // (always taken: the uniform branches cut the step into basic blocks, which keeps hipcc from stretching live ranges across phases)
        V7_OPQ(aF); V7_OPQ(aWB);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 dyv = LD4((aF ^ (unsigned)(t << 6)) + TB);
          const float4 w = LD4(aWB + 512u * t);
          dhx = MFMA(w.x, dyv.x, dhx);
          dhx = MFMA(w.y, dyv.y, dhx);
          dhx = MFMA(w.z, dyv.z, dhx);
          dhx = MFMA(w.w, dyv.w, dhx);
        }
      }
      SCHED_FENCE();
      // ---- P3: logits, softmax/gate backward ----
      float dge[4], hh[2], dA[2], at[2];
      {
        float dots[2], dAd[2];
        V7_OPQ(vq);
        const unsigned aq = rb + vq * 64u;
        {
          float d0 = 0.f, d1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 u = LD4(aq + 16u * i);
            d0 = fmaf(u.x, Kf[4*i], d0);   d1 = fmaf(u.y, Kf[4*i+1], d1);
            d0 = fmaf(u.z, Kf[4*i+2], d0); d1 = fmaf(u.w, Kf[4*i+3], d1);
          }
          dots[0] = d0; dots[1] = d1;
        }
        SCHED_FENCE();
        {
          V7_OPQ(cDV); V7_OPQ(aV);
          const unsigned ad = rb + cDV;
          float e0 = 0.f, e1 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = LD4(ad ^ (unsigned)(i << 4)), vv = LD4(aV + 16u * i);
            e0 = fmaf(v.x, vv.x, e0); e1 = fmaf(v.y, vv.y, e1);
            e0 = fmaf(v.z, vv.z, e0); e1 = fmaf(v.w, vv.w, e1);
            if (i == 1) { V7_OPQ(e0); V7_OPQ(e1); SCHED_FENCE(); }   // two pieces of V and dV_att in flight at a time (the pins keep the FMAs on this side)
          }
          dAd[0] = e0; dAd[1] = e1;
        }
        SCHED_FENCE();
        V7_OPQ(vq);
        const unsigned as = rb + 512u + vq * 32u;
        const float4 s0 = LD4(as), s1 = LD4(as + 16u);
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
        {
          V7_OPQ(vp);
          const MaskRegs mr{make_float2(1.f, 1.f), 0};
          apply_masks<false>(a, kadd, mr, ((size_t)(uint32_t)pair0 + vp) * BH, (int)vq, xl, gl);
        }
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
        V7_OPQ(vp); V7_OPQ(vq);
        ST4(sc1_b + vp * 64u + vq * 16u, make_float4(dge[0], dge[1], dge[2], dge[3]));
        ST2(sc2_b + vp * 32u + vq * 8u, make_float2(hh[0], hh[1]));
      }
      lds_sync();
      SCHED_FENCE();
      // ---- dK, dV, dQ ----
      if (!(a.guard & 4)) {
        V7_OPQ(vq);
        const unsigned aq = rb + vq * 64u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 u = LD4(aq + 16u * i);
          dKa[4*i]   = fmaf(dA[0], u.x, dKa[4*i]);   dKa[4*i+1] = fmaf(dA[1], u.y, dKa[4*i+1]);
          dKa[4*i+2] = fmaf(dA[0], u.z, dKa[4*i+2]); dKa[4*i+3] = fmaf(dA[1], u.w, dKa[4*i+3]);
        }
        SCHED_FENCE();
        V7_OPQ(cDV);
        const unsigned ad = rb + cDV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = LD4(ad ^ (unsigned)(i << 4));
          dVa[4*i]   = fmaf(at[0], v.x, dVa[4*i]);   dVa[4*i+1] = fmaf(at[1], v.y, dVa[4*i+1]);
          dVa[4*i+2] = fmaf(at[0], v.z, dVa[4*i+2]); dVa[4*i+3] = fmaf(at[1], v.w, dVa[4*i+3]);
        }
        SCHED_FENCE();
        float dq[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) { dq[2 * k] = dA[0] * Kf[2 * k]; dq[2 * k + 1] = dA[1] * Kf[2 * k + 1]; }
        // dQ[l] partial over this tile's 16 keys -> HBM, summed over key tiles by the next prologue (or k_node_bwd)
        V7_OPQ(vp);
        const float dqr = reduce16_keep_own(dq, (int)vp);
        V7_OPQ(vp); V7_OPQ(vq);
        (a.dqp + (((size_t)b * ntile + mt) * N + l) * 64)[vp + 16u * vq] = dqr;
      }
      SCHED_FENCE();
      // ---- P4: weight-gradient contractions over the 16 pairs of the tile ----
      if (!(a.guard & 1)) {
        V7_OPQ(vp); V7_OPQ(vq); V7_OPQ(aT);
        float bT[4], bR[4];
        const unsigned r1 = sc1_b + vq * 64u + vp * 4u, r2 = sc2_b + vq * 32u + (vp & 7u) * 4u;
        const bool hcol = vp < 8u;
        const float one8 = vp == 8u ? 1.0f : 0.f;   // [H_hat | 1 | 0]: column 8 collects the bias gradient
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bT[s] = LD1(r1 + 256u * s);
          const float hv = LD1(r2 + 128u * s);
          bR[s] = hcol ? hv : one8;
        }
        ssum += (bT[0] + bT[1]) + (bT[2] + bT[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            accT[t] = MFMA(LD1(aT + 4u * (256 * s + 16 * (t ^ s))), bT[s], accT[t]);
            accR[t] = MFMA(LD1(aT + TB + 4u * (256 * s + 16 * (t ^ s))), bR[s], accR[t]);
          }
          if (t == 1) SCHED_FENCE();   // two channel tiles' operands in flight at a time
        }
      }
      lds_sync();
      SCHED_FENCE();
      // ---- P5: d(ehat) = Wp . dGE, LayerNorm backward, de = de' + ... ----
      // The xhat / de' fragments go to registers FIRST: with them read, both LDS tiles are dead and the next row's tiles are requested
      // before this phase's 16 MFMAs and its LayerNorm arithmetic instead of behind them (each tile has one buffer: the request cannot
      // go out earlier than its last reader, and its latency is what the other two waves of the SIMD have to cover).
      if (!(a.guard & 2)) {
        float4 xh[4], dy[4];
        V7_OPQ(aF);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned ft = aF ^ (unsigned)(t << 6);
          xh[t] = LD4(ft);
          dy[t] = LD4(ft + TB);
        }
        lds_sync();   // the last reads of both tiles have retired: the next row may land
        SCHED_FENCE();
        if (l + 1 < r_hi) {
          tile_dma<DE>(et_lds, e_in + (pair0 + (size_t)N) * DE, off0);
          tile_dma<DE>(dt_lds, dey_in + (pair0 + (size_t)N) * DE, off0);
        }
        SCHED_FENCE();
        float4 dxh[4];
        float m1 = 0.f, m2 = 0.f;
        V7_OPQ(aW);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v4f d = {0.f, 0.f, 0.f, 0.f};
          const float4 w = LD4(aW + oD + 1024u * t);
          d = MFMA(w.x, dge[0], d);
          d = MFMA(w.y, dge[1], d);
          d = MFMA(w.z, dge[2], d);
          d = MFMA(w.w, dge[3], d);
          dxh[t] = make_float4(d[0], d[1], d[2], d[3]);
          m1 += (d[0] + d[1]) + (d[2] + d[3]);
          m2 = fmaf(d[0], xh[t].x, m2); m2 = fmaf(d[1], xh[t].y, m2);
          m2 = fmaf(d[2], xh[t].z, m2); m2 = fmaf(d[3], xh[t].w, m2);
        }
        m1 = sum_over_q(m1) * (1.0f / DE);
        m2 = sum_over_q(m2) * (1.0f / DE);
        if (a.flags & EGT_BF_NO_EDGE_LN) { m1 = 0.f; m2 = 0.f; }   // no norm_edge: d e = de' + d(proj input)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dxh[t].x = dy[t].x + rstd * (dxh[t].x - m1 - xh[t].x * m2);
          dxh[t].y = dy[t].y + rstd * (dxh[t].y - m1 - xh[t].y * m2);
          dxh[t].z = dy[t].z + rstd * (dxh[t].z - m1 - xh[t].z * m2);
          dxh[t].w = dy[t].w + rstd * (dxh[t].w - m1 - xh[t].w * m2);
        }
        SCHED_FENCE();
        V7_OPQ(gOff);
        char* orow = reinterpret_cast<char*>(dex_o + pair0 * DE) + gOff;
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(orow + 64 * t) = dxh[t];
      }
      SCHED_FENCE();
    }
#undef V7_OPQ
#undef LD4
#undef LD1
#undef ST4
#undef ST2
  }
  ssum = sum_over_q(ssum);
  __syncthreads();   // every wave is out of its row loop: the tile area is free
  {   // dK / dV of a key tile = the sum over its row chunks: [wave][32 values][64 lanes], added by the tile's first wave
    float* mg = sm + wave * 2048 + lane;
    if (ch != 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { mg[i * 64] = dKa[i]; mg[(16 + i) * 64] = dVa[i]; }
    }
    __syncthreads();
    if (ch == 0) {
      for (int c = 1; c < wpt; ++c) {
        const float* o = sm + (wave + c) * 2048 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) { dKa[i] += o[i * 64]; dVa[i] += o[(16 + i) * 64]; }
      }
      float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
      float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
        vo[i ^ ph] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);   // un-swizzle the pieces
      }
    }
    __syncthreads();
  }
  float* ep = sm + wave * G::EP;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ep[(16 * t + 4 * q + r) * 16 + p] = accT[t][r];
      ep[G::DEP * 16 + 16 + (16 * t + 4 * q + r) * 16 + p] = accR[t][r];
    }
  if (q == 0) ep[G::DEP * 16 + p] = ssum;
  __syncthreads();
  float* out = a.epart + (size_t)wg * G::EP;
  for (int i = threadIdx.x; i < G::EP; i += 64 * V7_WAVES) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < V7_WAVES; w += 4)
      s += (sm[w * G::EP + i] + sm[(w + 1) * G::EP + i]) + (sm[(w + 2) * G::EP + i] + sm[(w + 3) * G::EP + i]);
    out[i] = s;
  }
}
