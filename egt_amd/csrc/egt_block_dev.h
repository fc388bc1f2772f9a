// Device helpers shared by the pair kernels of the fused attention block (egt_block_fwd.h / egt_block_bwd.h: MFMA-tile kernels;
// egt_narrow.hip: De = 8 VALU kernels): mask application in the reference order, the node-side epilogue of the
// forward and the node-side prologue of the backward.
#pragma once
#include "egt_block.h"
#include "egt_tile.h"

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

// Per-pair mask inputs, fetched with the tile prefetch (unconditional, clamped).
struct MaskRegs { float2 mv; unsigned short rb; };

// EGT_BF_SEED_DEVICE: the mask stream's seed is completed on the device (uniform scalar loads), so that a
// captured launch (hipGraph) draws a fresh sample on every replay.  First statement of every pair kernel.
__device__ __forceinline__ void seed_from_device(BlockArgs& a) {
  if (a.sd) { a.s0 ^= a.sd[0]; a.s1 ^= a.sd[1]; }
}

template <bool ML>
__device__ __forceinline__ void mask_gload(const BlockArgs& a, MaskRegs& mr, size_t pairc, int q) {
  // pairc: linear pair index of a VALID pair (clamped by the caller)
  if (!ML) return;
  if (a.M) mr.mv = *reinterpret_cast<const float2*>(a.M + pairc * BH + 2 * q);
  if (a.rm) mr.rb = *reinterpret_cast<const unsigned short*>(a.rm + pairc * BH + 2 * q);
}

// logits x and gate-logits gl of the lane's pair, heads 2q+j; masks ADDED in the
// reference's order (egt_layers.py:91-108): key padding, attention mask, random mask
template <bool ML>
__device__ __forceinline__ void apply_masks(const BlockArgs& a, float kadd, const MaskRegs& mr,
                                            size_t idx8, int q, float (&x)[2], float (&gl)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float xv = x[j], gv = gl[j];
    if (a.km) { xv += kadd; gv += kadd; }
    if (ML && a.M) {
      const float mm = ((j ? mr.mv.y : mr.mv.x) - 1.0f) * EGT_NEG;
      xv += mm; gv += mm;
    }
    if ((ML && a.rm) || a.rng_rm) {
      bool hit;
      if (ML && a.rm) hit = ((mr.rb >> (8 * j)) & 0xFF) != 0;
      else hit = (egt_hash32((uint32_t)(idx8 + 2 * q + j), a.s0, a.s1) >> 8) < a.rm_thr;
      const float mrv = hit ? -EGT_NEG : 0.0f;
      xv += mrv; gv += mrv;
    }
    x[j] = xv; gl[j] = gv;
  }
}

// transpose-reduce 16 per-lane values over the 16 key lanes (same q): lane p returns the
// sum of element p.  VALU and MFMA issue add up on a gfx950 SIMD (profiles/r03_mfma_valu_issue.md), so instruction count is
// what this costs: the "xor 8" and "xor 4" levels are DPP adds whose SECOND instruction of a pair writes only the banks
// (groups of 4 lanes) that keep the other element -- no selects, no moves: 2 instructions per result instead of 6-7
// (v_cndmask x2, v_mov 0, v_mov_dpp x1-2, v_add).  Same operands, same association order as the select form: bit-identical.
//   row_ror:8 = lane ^ 8;  row_ror:12 reads lane + 4 (valid for banks 0 and 2), row_ror:4 reads lane - 4 (banks 1 and 3).
// The s_nop 1 ahead of each block are the two wait states between a VALU write of a VGPR and a DPP read of it (hipcc pads
// nothing inside an asm statement); inside a block no DPP source was written by one of the two instructions before it.
__device__ __forceinline__ float reduce16_keep_own(const float (&v)[16], int p) {
  float t0, t1, t2, t3, t4, t5, t6, t7, u0, u1, u2, u3;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
        "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %6, %6 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %7, %7 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)
      : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7));
  const float w4[4] = {u0, u1, u2, u3};
  float w2[2];
  const bool b1 = (p & 2) != 0, b0 = (p & 1) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float snd = b1 ? w4[i] : w4[i + 2], kp = b1 ? w4[i + 2] : w4[i];
    w2[i] = kp + lane_xor<2>(snd);
  }
  const float snd = b0 ? w2[0] : w2[1], kp = b0 ? w2[1] : w2[0];
  return kp + lane_xor<1>(snd);
}

#define KV_LD 132  // LDS row stride (floats) of the staged [K|V] rows
#define QS_LD 68   // LDS row stride of the staged Q rows (reused for the V_att rows of the epilogue)

// Stage K | V of a graph's N keys into LDS rows of stride KV_LD, eight 16-byte loads per thread in flight per round trip (the plain
// one-element loop compiles to load -> wait -> store per iteration: N * 32 / NT dependent round trips at kernel entry).
template <int NT>
__device__ __forceinline__ void fwd_stage_kv(float* kvs, const float* kvsrc, int N) {
  for (int base = 0; base < N * 32; base += 8 * NT) {
    float4 kvb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + threadIdx.x + NT * u, row = min(i >> 5, N - 1), f = i & 31;
      kvb[u] = *reinterpret_cast<const float4*>(kvsrc + (size_t)row * QKVP + 64 + f * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + threadIdx.x + NT * u, row = i >> 5, f = i & 31;
      if (i < N * 32) *reinterpret_cast<float4*>(kvs + row * KV_LD + f * 4) = kvb[u];
    }
  }
}

// ---- node-side epilogue: the workgroup holds V_att of its 16 rows in qs ----
//   epi >= 1: h' = V_att.Wo + bo + h                      (dense_mha + res_mha, :136,140)
//   epi == 2: qkv of the NEXT block = LN(h').Wqkv' + bqkv' (norm_mha + dense_qkv, :109,113), packed
// Contraction order k = 4s + q on both MFMA operands; weights come straight from L2.
// Node width Dh = 8 * DK <= 64 (H = 8 heads): the tiles are always four 16-column tiles of a zero-padded 64-wide row -- channel
// c = k * 8 + head is the same index in the padded and in the natural row, so the padding is the columns c >= Dh: their weights,
// biases and inputs read as zero, their outputs are never stored, the LayerNorm divides by Dh and keeps them at zero.  D64: Dh is
// the compile-time 64 of the headline geometry (no guards).
// The epilogue's global inputs (weight fragments, bias, residual rows) live in this register set between `fwd_epilogue_load` (every load
// issued: one memory round trip, which a kernel can put under a phase of its own -- k_narrow_fwd: under the merge of its key ranges) and
// `fwd_epilogue_compute`.
struct FwdEpiRegs { float wo[16], wq[3][16], bo, hres[4]; };
template <bool D64>
__device__ __forceinline__ void fwd_epilogue_load(const BlockArgs& a, FwdEpiRegs& R, int b, int l0, int N, int wave, int p, int q) {
    const int Dh = D64 ? 64 : a.Dh;
    float (&wo)[16] = R.wo; float (&wq)[3][16] = R.wq;
    const int c = wave * 16 + p;
    const bool cok = D64 || c < Dh;
    {   // B operands from the fragment-major weight copies (egt_block.h: WFRAG_FWO / WFRAG_FWQ; zero where the padded row has no
        // channel): lane-linear 16-byte loads, 16 per lane instead of 64 dword loads at a row stride
      const int ln = 16 * q + p;
      const float4* fwo = reinterpret_cast<const float4*>(a.wfrag + WFRAG_FWO) + wave * 256 + ln;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 v = fwo[s4 * 64];
        wo[4 * s4] = v.x; wo[4 * s4 + 1] = v.y; wo[4 * s4 + 2] = v.z; wo[4 * s4 + 3] = v.w;
      }
      if (a.epi == 2) {
        const float4* fwq = reinterpret_cast<const float4*>(a.nx_wfrag + WFRAG_FWQ) + wave * 768 + ln;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const float4 v = fwq[(j * 4 + s4) * 64];
            wq[j][4 * s4] = v.x; wq[j][4 * s4 + 1] = v.y; wq[j][4 * s4 + 2] = v.z; wq[j][4 * s4 + 3] = v.w;
          }
      }
    }
    R.bo = cok ? a.bo[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) R.hres[r] = cok ? a.h[((size_t)b * N + min(l0 + 4 * q + r, N - 1)) * Dh + c] : 0.f;
}
template <bool D64>
__device__ __forceinline__ void fwd_epilogue_compute(const BlockArgs& a, const FwdEpiRegs& R, float* sm, float* qs, int b, int l0, int nrows, int N,
                                                     int wave, int p, int q) {   // rows [l0, l0 + nrows) of graph b, nrows <= 16 (<= 0: nothing to store)
    const int Dh = D64 ? 64 : a.Dh;
    const float (&wo)[16] = R.wo; const float (&wq)[3][16] = R.wq; const float (&hres)[4] = R.hres;
    const float bo = R.bo;
    const int c = wave * 16 + p;
    const bool cok = D64 || c < Dh;
    __syncthreads();   // every row's V_att is in qs; the tile area is idle from here on
    float* hs = sm;    // [16][QS_LD]
    v4f acc = {bo, bo, bo, bo};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = MFMA(qs[p * QS_LD + 4 * s + q], wo[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r, l = l0 + row;
      const float hv = acc[r] + hres[r];
      if (row < nrows && cok) a.h_out[((size_t)b * N + l) * Dh + c] = hv;
      hs[row * QS_LD + c] = hv;
    }
    if (a.epi == 2) {
      __syncthreads();
      {   // LayerNorm of row 4*wave + q: 16 lanes x 4 columns
        float* x = hs + (4 * wave + q) * QS_LD;
        const float inv = D64 ? 1.0f / 64 : 1.0f / (float)Dh;
        float v[4], sm1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = x[p + 16 * i]; sm1 += v[i]; }
        const float mu = row_sum16(sm1) * inv;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = (D64 || p + 16 * i < Dh) ? v[i] - mu : 0.f; ss = fmaf(v[i], v[i], ss); }
        const float rstd = rsqrtf(row_sum16(ss) * inv + a.ln_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ci = p + 16 * i;
          const bool ok = D64 || ci < Dh;
          x[ci] = ok ? fmaf(v[i] * rstd, a.nx_nm_g[ci], a.nx_nm_b[ci]) : 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int cq = (wave + 4 * j) * 16 + p;
        const float bq = (D64 || (cq & 63) < Dh) ? a.nx_bqkv[(cq >> 6) * Dh + (cq & 63)] : 0.f;
        v4f aq = {bq, bq, bq, bq};
#pragma unroll
        for (int s = 0; s < 16; ++s) aq = MFMA(hs[p * QS_LD + 4 * s + q], wq[j][s], aq);
        const int sx = cq >> 6, cc = cq & 63, kk = cc >> 3, hh = cc & 7;
        const int pos = sx * 64 + (hh >> 1) * 16 + kk * 2 + (hh & 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int l = l0 + 4 * q + r;
          if (4 * q + r < nrows) a.nx_qkvp[((size_t)b * N + l) * QKVP + pos] = aq[r];   // (padding columns store the zeros the pair kernels expect there)
        }
      }
    }
  }
template <bool D64>
__device__ __forceinline__ void fwd_node_epilogue_t(const BlockArgs& a, float* sm, float* qs, int b, int l0, int nrows, int N,
                                                    int wave, int p, int q) {
  FwdEpiRegs R;
  fwd_epilogue_load<D64>(a, R, b, l0, N, wave, p, q);
  fwd_epilogue_compute<D64>(a, R, sm, qs, b, l0, nrows, N, wave, p, q);
}
__device__ __forceinline__ void fwd_node_epilogue(const BlockArgs& a, float* sm, float* qs, int b, int l0, int nrows, int N,
                                                  int wave, int p, int q) {
  if (a.Dh == 64) fwd_node_epilogue_t<true>(a, sm, qs, b, l0, nrows, N, wave, p, q);
  else fwd_node_epilogue_t<false>(a, sm, qs, b, l0, nrows, N, wave, p, q);
}
// the two halves on their own (a.Dh == 64 decided by the caller once)
__device__ __forceinline__ void fwd_node_epilogue_load(const BlockArgs& a, FwdEpiRegs& R, int b, int l0, int N, int wave, int p, int q) {
  if (a.Dh == 64) fwd_epilogue_load<true>(a, R, b, l0, N, wave, p, q);
  else fwd_epilogue_load<false>(a, R, b, l0, N, wave, p, q);
}
__device__ __forceinline__ void fwd_node_epilogue_finish(const BlockArgs& a, const FwdEpiRegs& R, float* sm, float* qs, int b, int l0, int nrows, int N,
                                                         int wave, int p, int q) {
  if (a.Dh == 64) fwd_epilogue_compute<true>(a, R, sm, qs, b, l0, nrows, N, wave, p, q);
  else fwd_epilogue_compute<false>(a, R, sm, qs, b, l0, nrows, N, wave, p, q);
}

// the barriers of fwd_node_epilogue, for waves of a workgroup that take no part in it (a.epi is uniform)
__device__ __forceinline__ void fwd_node_epilogue_idle(const BlockArgs& a) {
  __syncthreads();
  if (a.epi == 2) { __syncthreads(); __syncthreads(); }
}

#define QD_LD 160  // per row: Q[64] | dV_att[64] | stats[32]
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Stage the query-side rows of a backward workgroup into qd ([row][QD_LD]: Q | dV_att | softmax statistics with the row sum
// inverted): every thread issues ALL its 16-byte loads before the first LDS store -- ONE memory round trip.  (Written as a plain
// `for (i = tid; i < nl * 40; i += NT) qd[..] = src[..]` hipcc emits load -> s_waitcnt vmcnt(0) -> ds_write per iteration: two to
// three dependent round trips at kernel entry, found in round 5 in every backward kernel.)  NT threads, at most NT * NU / 40 rows.
template <int NT, int NU>
__device__ __forceinline__ void bwd_stage_rows(const BlockArgs& a, float* qd, int b, int l_begin, int nl) {
  float4 sv[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = threadIdx.x + NT * u, r = i / 40, f = i % 40;
    const bool ok = i < nl * 40 && !(a.pro && f >= 16 && f < 32);   // (with the fused prologue dV_att is computed there)
    const size_t rowl = (size_t)b * a.N + l_begin + (ok ? r : 0);
    const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                     : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                              : a.stats + rowl * 32 + (f - 32) * 4;
    sv[u] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = threadIdx.x + NT * u, r = i / 40, f = i % 40;
    if (i < nl * 40 && !(a.pro && f >= 16 && f < 32)) {
      float4 v = sv[u];
      if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
      *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
    }
  }
}

// ---- node-side prologue of the backward pair kernel (16 rows, 256 threads; Dh = 8 DK <= 64 as a zero-padded 64-wide row,
//      see fwd_node_epilogue_t: D64 = the compile-time 64 of the headline geometry) -------------
// What k_node_bwd does between two pair kernels is local to a node row, so the workgroup that
// owns 16 query rows of layer L does it itself before its tile loop:
//   [pro == 2] rows of the layer above (L+1): dQKV = packed dQ + dK/dV partial sums (written by the
//              pair kernel of L+1), d h_ln = dQKV.Wqkv^T (48 MFMA / wave), LayerNorm backward +
//              residual -> dh(L+1) = dh'(L); dQKV and dh rows go to HBM for the deferred weight-
//              gradient kernel; bias / LN-parameter column sums -> up_spart
//   [pro == 1] dh'(L) rows = dh_out (top of the chain)
//   then       dV_att(L) = dh'.Wo^T (16 MFMA / wave, packed straight into the qd rows in LDS),
//              delta = sum_k dV_att*V_att into the qd statistics, dbo column sums -> sbo
// One launch per layer on the dh critical path instead of two; dV_att / delta never touch HBM.
// `ws`: the (still idle) per-wave tile area; `qd`: the staged [16][QD_LD] rows.
#ifndef EGT_PRO_UNROLL
#define EGT_PRO_UNROLL 4   // partial loads in flight per element of the dQ / dK / dV gather
#endif
#define BWD_PRO_WS 8192   // floats of LDS scratch the prologue needs (dQKV, xhat, d h_ln, dh' rows, partials)
// HOIST: the loads of the dV_att step (Wo columns, V_att rows) are issued with the first round of global loads instead of
// after the dh' rows exist: one memory round trip on the kernel's critical path instead of two (costs 20 registers across
// the first part)
#define NSTAMP(i) do {} while (0)
// The prologue's global inputs live in this register set between `bwd_prologue_load` (every load of the step issued: ONE
// memory round trip, which a kernel can overlap with its other start-up requests) and `bwd_prologue_compute`.
struct BwdProRegs { float4 hx, gq[3], wq[12], wo[4]; float dho[4], gmm[4], va[4]; };

#define BWD_PRO_COMMON()                                                                                   \
  constexpr int LD = 68, LD3 = 196;                                                                        \
  const int t = ptid, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);                        \
  const int p = lane & 15, q = lane >> 4, N = a.N;                                                         \
  const size_t row0 = (size_t)b * N + l_begin;                                                             \
  const int lnrow = 4 * wave + q;    /* LayerNorm mapping: row 4*wave + q, columns p + 16 i */            \
  /* ragged last row group (N not a multiple of 16): loads are clamped to the graph's last row, the rows  \
     past the end contribute zeros to every sum and are never stored */                                   \
  const int nv = pnv >= 0 ? pnv : min(a.TL, N - l_begin);  /* valid rows of this 16-row group (<= 16) */                          \
  auto rc = [&](int r) { return row0 + max(min(r, nv - 1), 0); };   /* clamped global row */               \
  const int Dh = D64 ? 64 : a.Dh;                                                                          \
  (void)LD; (void)LD3; (void)lnrow; (void)p; (void)q; (void)rc; (void)Dh

// Wo columns and V_att rows of the dV_att step
template <int DE, bool D64>
__device__ __forceinline__ void bwd_prologue_load_wo_va(const BlockArgs& a, int b, int l_begin, BwdProRegs& R, int ptid, int pnv) {
  BWD_PRO_COMMON();
  const bool iok = D64 || 16 * wave + p < Dh;   // the lane's dV_att channel exists
#pragma unroll
  for (int s = 0; s < 4; ++s)
    R.wo[s] = reinterpret_cast<const float4*>(a.wfrag + WFRAG_BWO)[(wave * 4 + s) * 64 + lane];   // fragment-major Wo (zero-padded)
#pragma unroll
  for (int r = 0; r < 4; ++r) R.va[r] = iok ? a.v_att[rc(4 * q + r) * Dh + 16 * wave + p] : 0.f;
}
// [pro == 2] every global input of the dQKV / d h_ln / LayerNorm-backward step
template <int DE, bool D64>
__device__ __forceinline__ void bwd_prologue_load_main(const BlockArgs& a, int b, int l_begin, BwdProRegs& R, int ptid, int pnv) {
  BWD_PRO_COMMON();
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- every global input in one round trip ----
    R.hx = (D64 || (t & 15) * 4 < Dh) ? *reinterpret_cast<const float4*>(a.up_h + rc(t >> 4) * Dh + (t & 15) * 4) : z4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = D64 || p + 16 * i < Dh;
      R.dho[i] = ok ? a.up_dh_out[rc(lnrow) * Dh + p + 16 * i] : 0.f;
      R.gmm[i] = ok ? a.up_nm_g[p + 16 * i] : 0.f;
    }
    // partial counts of the layer above (same geometry, same kernels): dQ partials per row = a.NQP (key tiles for the
    // MFMA-tile kernels, 1 for k_narrow_bwd, which reduces its key tiles itself), dK/dV partials per key = a.NLR (row groups)
    // The three elements of a thread are gathered together, EGT_PRO_UNROLL partials of each per round: 3 * EGT_PRO_UNROLL loads
    // in flight and ONE wait per round (a loop per element compiles to a `s_waitcnt vmcnt(0)` per element: three serialized
    // round trips at the head of every backward kernel).  A partial index past an element's count re-reads its last partial
    // (a cache hit) and adds nothing; each element's partials are still summed in index order.
    {
      const float* base[3]; size_t pstride[3]; int NPu[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = t + u * 256, r = i / 48, pos4 = (i % 48) * 4, sx = pos4 >> 6;
        const int rr = max(min(r, nv - 1), 0);
        NPu[u] = sx == 0 ? a.NQP : a.NLR;
        base[u] = sx == 0 ? a.up_dqp + ((size_t)b * NPu[u] * N + l_begin + rr) * 64 + pos4
                          : a.up_dkvp + (((size_t)b * NPu[u] * N + l_begin + rr) * 2 + (sx - 1)) * 64 + (pos4 & 63);
        pstride[u] = sx == 0 ? (size_t)N * 64 : (size_t)N * 128;
        R.gq[u] = z4;
      }
      const int NPm = max(a.NQP, a.NLR);
      for (int p0 = 0; p0 < NPm; p0 += EGT_PRO_UNROLL) {
        float4 w[3][EGT_PRO_UNROLL];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int j = 0; j < EGT_PRO_UNROLL; ++j)
            w[u][j] = *reinterpret_cast<const float4*>(base[u] + (size_t)max(min(p0 + j, NPu[u] - 1), 0) * pstride[u]);
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int j = 0; j < EGT_PRO_UNROLL; ++j) {
            const bool in = p0 + j < NPu[u];
            R.gq[u].x += in ? w[u][j].x : 0.f; R.gq[u].y += in ? w[u][j].y : 0.f;
            R.gq[u].z += in ? w[u][j].z : 0.f; R.gq[u].w += in ? w[u][j].w : 0.f;
          }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if ((t + u * 256) / 48 >= nv) R.gq[u] = z4;
    }
    // B operand of d h_ln = dQKV.Wqkv^T: Wqkv[kk = 16 wave + p][48 q .. 48 q + 47] (contraction order c = 48 q + s, c a column of
    // the padded [3][64] row: section c >> 6, channel c & 63)
    // -- read from the fragment-major copy (WFRAG_BWQ, zero-padded): one lane-linear 16-byte load per step
#pragma unroll
    for (int s = 0; s < 12; ++s)
      R.wq[s] = reinterpret_cast<const float4*>(a.up_wfrag + WFRAG_BWQ)[(wave * 12 + s) * 64 + lane];
}

template <int DE, bool D64>
__device__ __forceinline__ void bwd_prologue_compute(const BlockArgs& a, float* ws, float* qd, int b, int l_begin, int wg, BwdProRegs& R,
                                                     bool wo_loaded, unsigned* tp, int ptid, int pnv) {
  float* dqs = ws;                   // dQKV  [16][196]
  float* xs = dqs + 16 * 196;        // xhat  [16][68]
  float* dls = xs + 16 * 68;         // d h_ln
  float* dhs = dls + 16 * 68;        // dh'
  float* rs = dhs + 16 * 68;         // rstd  [16]
  float* dlp = rs + 16;              // delta partials [4][16][8]
  static_assert(16 * 196 + 3 * 16 * 68 + 16 + 4 * 16 * 8 <= BWD_PRO_WS, "prologue scratch");
  BWD_PRO_COMMON();
  float4 (&wo)[4] = R.wo;
  float (&va)[4] = R.va;
  if (a.pro == 2) {
    float4 (&gq)[3] = R.gq; float4 (&wq)[12] = R.wq; float (&dho)[4] = R.dho; float (&gmm)[4] = R.gmm; const float4 hx = R.hx;
    NSTAMP(0);
    *reinterpret_cast<float4*>(xs + (t >> 4) * LD + (t & 15) * 4) = hx;
    NSTAMP(1);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = t + u * 256, r = i / 48, pos4 = (i % 48) * 4, sx = pos4 >> 6;
      const int qq = (pos4 >> 4) & 3, k0 = (pos4 >> 1) & 7;
      float* d = dqs + r * LD3 + sx * 64 + k0 * 8 + 2 * qq;
      const bool k0ok = D64 || k0 * 8 < Dh, k1ok = D64 || (k0 + 1) * 8 < Dh;   // per-head channels k0, k0 + 1 < DK
      d[0] = k0ok ? gq[u].x : 0.f; d[1] = k0ok ? gq[u].y : 0.f; d[8] = k1ok ? gq[u].z : 0.f; d[9] = k1ok ? gq[u].w : 0.f;
    }
    __syncthreads();
    const float invD = D64 ? 1.0f / 64 : 1.0f / (float)Dh;
    {   // LN forward statistics -> xhat in place
      float* xr = xs + lnrow * LD;
      float v[4], s1 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = xr[p + 16 * i]; s1 += v[i]; }
      const float mu = row_sum16(s1) * invD;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = (D64 || p + 16 * i < Dh) ? v[i] - mu : 0.f; ss = fmaf(v[i], v[i], ss); }
      const float rstd = rsqrtf(row_sum16(ss) * invD + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[p + 16 * i] = v[i] * rstd;
      if (p == 0) rs[lnrow] = rstd;
    }
    {   // d h_ln[row][kk] for kk tile = wave
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      const float* ar = dqs + p * LD3 + 48 * q;
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        const float4 av = *reinterpret_cast<const float4*>(ar + 4 * s);
        acc = MFMA(av.x, wq[s].x, acc);
        acc = MFMA(av.y, wq[s].y, acc);
        acc = MFMA(av.z, wq[s].z, acc);
        acc = MFMA(av.w, wq[s].w, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dls[(4 * q + r) * LD + 16 * wave + p] = acc[r];
    }
    NSTAMP(2);
    for (int i = t; i < nv * 48; i += 256) {   // dQKV rows out (natural channel order, [3][Dh] per row) for k_node_wgrads
      const int r = i / 48, c4 = (i % 48) * 4;
      if (D64 || (c4 & 63) < Dh)
        *reinterpret_cast<float4*>(a.up_dqkv_sv + (row0 + r) * (3 * Dh) + (c4 >> 6) * Dh + (c4 & 63)) = *reinterpret_cast<const float4*>(dqs + r * LD3 + c4);
    }
    __syncthreads();
    {   // LayerNorm backward + residual -> dh' rows (HBM + LDS)
      const float* xr = xs + lnrow * LD;
      const float* dl = dls + lnrow * LD;
      float dx[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dx[i] = dl[p + 16 * i] * gmm[i];
        m1 += dx[i];
        m2 = fmaf(dx[i], xr[p + 16 * i], m2);
      }
      m1 = row_sum16(m1) * invD;
      m2 = row_sum16(m2) * invD;
      const float rstd = rs[lnrow];
      float* dh_out_rw = const_cast<float*>(a.dh_out);   // this layer's dh' IS the upper layer's dh
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = p + 16 * i;
        const bool ok = lnrow < nv && (D64 || c < Dh);
        const float dv = ok ? dho[i] + rstd * (dx[i] - m1 - xr[c] * m2) : 0.f;
        if (ok) dh_out_rw[(row0 + lnrow) * Dh + c] = dv;
        dhs[lnrow * LD + c] = dv;
      }
    }
    {   // column sums over the 16 rows: dbqkv | dgamma, dbeta of the layer above
      float* sp = a.up_spart + (size_t)wg * (5 * Dh);   // [3 Dh: dbqkv | Dh: dgamma | Dh: dbeta]
      if (t < 192) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) { s0 += dqs[r * LD3 + t]; s1 += dqs[(r + 1) * LD3 + t]; }
        if ((D64 || (t & 63) < Dh) && nv > 0) sp[(t >> 6) * Dh + (t & 63)] = s0 + s1;
      } else {
        const int c = t - 192;
        float g0 = 0.f, b0 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d0 = dls[r * LD + c]; g0 = fmaf(d0, xs[r * LD + c], g0); b0 += d0; }
        if ((D64 || c < Dh) && nv > 0) {
          sp[3 * Dh + c] = g0;
          sp[4 * Dh + c] = b0;
        }
      }
    }
  } else {
    const bool ok = (t >> 4) < nv && (D64 || (t & 15) * 4 < Dh);
    const float4 dv4 = ok ? *reinterpret_cast<const float4*>(a.dh_out + rc(t >> 4) * Dh + (t & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dhs + (t >> 4) * LD + (t & 15) * 4) =
        make_float4(ok ? dv4.x : 0.f, ok ? dv4.y : 0.f, ok ? dv4.z : 0.f, ok ? dv4.w : 0.f);
  }
  // ---- dV_att = dh'.Wo^T for i tile = wave (contraction order c = 16 q + s), delta, dbo ----
  NSTAMP(3);
  if (!wo_loaded) bwd_prologue_load_wo_va<DE, D64>(a, b, l_begin, R, ptid, pnv);
  __syncthreads();
  NSTAMP(4);
  {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const float* ar = dhs + p * LD + 16 * q;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 av = *reinterpret_cast<const float4*>(ar + 4 * s);
      acc = MFMA(av.x, wo[s].x, acc);
      acc = MFMA(av.y, wo[s].y, acc);
      acc = MFMA(av.z, wo[s].z, acc);
      acc = MFMA(av.w, wo[s].w, acc);
    }
    const int i = 16 * wave + p, k = i >> 3, hh = i & 7;
    const int pos = (hh >> 1) * 16 + k * 2 + (hh & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      // the kernels stage a.TL (<= 16) rows and put other buffers right behind them: rows past the group's last one are not
      // qd's to write (an unguarded store here raced with the weight slabs that k_block_bwd_v4r fills next, at TL < 16)
      if (row < nv) qd[row * QD_LD + 64 + pos] = acc[r];
      float pr = acc[r] * va[r];
      pr += lane_xor<8>(pr);   // the tile's two k values of head p & 7
      if (p < 8) dlp[(wave * 16 + row) * 8 + p] = pr;
    }
  }
  NSTAMP(5);
  if (t < 64) {   // dbo: column sums of dh'
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) { s0 += dhs[r * LD + t]; s1 += dhs[(r + 1) * LD + t]; }
    if ((D64 || t < Dh) && nv > 0) a.sbo[(size_t)wg * Dh + t] = s0 + s1;
  }
  __syncthreads();
  if (t < 128 && (t >> 3) < nv) {   // delta of (row, head)
    const int row = t >> 3, hd = t & 7;
    qd[row * QD_LD + 128 + hd * 4 + 2] = (dlp[(0 * 16 + row) * 8 + hd] + dlp[(1 * 16 + row) * 8 + hd]) +
                                         (dlp[(2 * 16 + row) * 8 + hd] + dlp[(3 * 16 + row) * 8 + hd]);
  }
}

template <int DE, bool D64, bool HOIST>
__device__ __forceinline__ void bwd_node_prologue_t(const BlockArgs& a, float* ws, float* qd, int b, int l_begin, int wg, unsigned* tp, int ptid, int pnv) {
  BwdProRegs R;
  if (HOIST) bwd_prologue_load_wo_va<DE, D64>(a, b, l_begin, R, ptid, pnv);
  if (a.pro == 2) bwd_prologue_load_main<DE, D64>(a, b, l_begin, R, ptid, pnv);
  bwd_prologue_compute<DE, D64>(a, ws, qd, b, l_begin, wg, R, HOIST, tp, ptid, pnv);
}
// The two halves on their own, for a kernel that issues the prologue's loads together with its other start-up requests (one memory
// round trip for all of them) and runs the arithmetic later: k_block_bwd_v5.
template <int DE>
__device__ __forceinline__ void bwd_node_prologue_load(const BlockArgs& a, BwdProRegs& R, int b, int l_begin) {
  if (a.pro != 2) return;
  if (a.Dh == 64) bwd_prologue_load_main<DE, true>(a, b, l_begin, R, threadIdx.x, -1);
  else bwd_prologue_load_main<DE, false>(a, b, l_begin, R, threadIdx.x, -1);
}
template <int DE>
__device__ __forceinline__ void bwd_node_prologue_finish(const BlockArgs& a, float* ws, float* qd, int b, int l_begin, int wg, BwdProRegs& R) {
  if (a.Dh == 64) bwd_prologue_compute<DE, true>(a, ws, qd, b, l_begin, wg, R, false, nullptr, threadIdx.x, -1);
  else bwd_prologue_compute<DE, false>(a, ws, qd, b, l_begin, wg, R, false, nullptr, threadIdx.x, -1);
}
// the barriers of bwd_node_prologue, for waves of a workgroup that take no part in it (a.pro is uniform)
__device__ __forceinline__ void bwd_node_prologue_idle(const BlockArgs& a) {
  if (a.pro == 2) { __syncthreads(); __syncthreads(); }
  __syncthreads();
  __syncthreads();
}
// ptid / pnv: the calling thread's index inside its 256-thread group and the group's valid rows -- defaults: the workgroup IS the
// group (threadIdx.x, min(a.TL, N - l_begin)).  (A workgroup of several 256-thread groups -- round 5's twelve-wave experiment, in the git
// history -- passes its own; every thread of the workgroup must make the call: the barriers inside are workgroup barriers.)
template <int DE, bool HOIST = false>
__device__ __forceinline__ void bwd_node_prologue(const BlockArgs& a, float* ws, float* qd, int b, int l_begin, int wg, unsigned* tp = nullptr,
                                                  int ptid = -1, int pnv = -1) {
  if (ptid < 0) ptid = threadIdx.x;
  if (a.Dh == 64) bwd_node_prologue_t<DE, true, HOIST>(a, ws, qd, b, l_begin, wg, tp, ptid, pnv);
  else bwd_node_prologue_t<DE, false, HOIST>(a, ws, qd, b, l_begin, wg, tp, ptid, pnv);
}

