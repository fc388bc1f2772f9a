// Fused EGT attention block for gfx950 — the data-parallel hot path.
//   (h', e') = edge_update_residual(h, e) around mha_block, pre-norm
//   (lib/models/graph_xformer_model_base.py:192-223 + :106-145, inner op
//    lib/models/egt_layers.py:57-143).
//
// Data flow per 16-pair tile (one query row l, 16 consecutive keys m; the
// [B,N,N,De] edge tensor makes that 16*De*4 contiguous bytes):
//   HBM --coalesced 16B loads--> LDS tile (XOR-swizzled 16B slots)
//   LDS --ds_read_b128--> MFMA B-fragments: lane (p = lane&15, q = lane>>4) owns
//        pair p and channels {16*t + 4*q + r}
//   LayerNorm (norm_edge) in registers (two-pass moments; 2 cross-lane adds)
//   v_mfma_f32_16x16x4_f32 x (De/4):  [Wg|We]^T(16 x De) . ehat^T(De x 16 pairs)
//        -> lane (p,q) receives G,E of pair p for heads 2q,2q+1  (no shuffles)
//   QK^T (d <= 8: 16 FMAs/lane), clip, +E, additive masks, per-lane ONLINE
//   softmax x sigmoid gate, A.V accumulated per lane; merged across the 16 key
//   lanes once per query row.
//   v_mfma x (De/16*2): Wr^T . H_hat^T -> residual update written back into the
//   LDS tile, streamed out with coalesced 16B stores.
// E, G, H_hat, A_tild never touch HBM.  Backward recomputes all of it from e,
// keeps per-(row,head) softmax statistics + V_att from the forward (flash-style
// delta), and does every weight gradient as MFMA contractions over the pair
// axis with deterministic per-workgroup partials.
// The node side of the block (norm_mha, dense_qkv, dense_mha + residual and their backward) is
// row-local, so it rides along: the forward's epilogue finishes h' and already produces the next
// block's packed QKV, the backward's prologue turns the dQ/dK/dV partials of the block above into
// dh and this block's dV_att / delta -- one launch per layer per direction.
//
// Lane roles follow the 16x16x4 f32 MFMA register maps (A: row=lane&15,
// k=lane>>4; B: k=lane>>4, col=lane&15; D: row=4*(lane>>4)+reg, col=lane&15).
#include "egt_common.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

#include "egt_block.h"
#include "egt_tile.h"

#include "egt_block_dev.h"
#include "egt_dma.h"

// ================================================================= forward =====
// Workgroup = (graph b, 16 query rows); wave w owns rows l = 16*lg + w + 4*i.
// KVL: K/V of the graph, Q of the 16 rows and the key-mask adds are staged in LDS.
// ML: attention-mask / injected-random-mask byte streams are present (their loads are
// compiled out of the headline kernel).
// FULL: N is a multiple of 16 (no ragged key tile): validity selects and address clamps fold away.
template <int DE, bool KVL, bool ML, bool FULL, bool BF>
__global__ void __launch_bounds__(256, ((DE <= 16 && !KVL) ? 4 : 2)) k_block_fwd(BlockArgs a) {   // narrow tiles without K/V in LDS: more resident waves (with K/V in LDS the LDS footprint caps a CU at two workgroups anyway)
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
  const int N = a.N;
  const int lgroups = (N + 15) / 16;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / lgroups, lg = wg % lgroups;
  float* tl0 = sm + wave * 2 * G::TILE_FLOATS;  // two tiles per wave (ping-pong)
  float* kvs = sm + 8 * G::TILE_FLOATS;         // [N][KV_LD]   (KVL)
  float* qs = kvs + (KVL ? N * KV_LD : 0);     // [16][QS_LD]  (KVL)
  float* kms = qs + (KVL ? 16 * QS_LD : 0);    // [N]          (KVL)
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;

  if (KVL) {
    const float* src = a.qkvp + (size_t)b * N * QKVP;
    for (int i = threadIdx.x; i < N * 32; i += 256) {
      const int row = i >> 5, f = i & 31;
      *reinterpret_cast<float4*>(kvs + row * KV_LD + f * 4) =
          *reinterpret_cast<const float4*>(src + (size_t)row * QKVP + 64 + f * 4);
    }
    for (int i = threadIdx.x; i < 16 * 16; i += 256) {
      const int row = i >> 4, f = i & 15, l = min(lg * 16 + row, N - 1);
      *reinterpret_cast<float4*>(qs + row * QS_LD + f * 4) =
          *reinterpret_cast<const float4*>(src + (size_t)l * QKVP + f * 4);
    }
    for (int i = threadIdx.x; i < N; i += 256)
      kms[i] = (a.km && a.km[(size_t)b * N + i] == 0) ? -EGT_NEG : 0.0f;
  }

  // lane-constant MFMA operands
  float wA[4 * G::TILES], wrA[G::TILES][2], c2r[4];
  float4 brv[G::TILES];
#pragma unroll
  for (int t = 0; t < 4 * G::TILES; ++t) wA[t] = a.pw[(16 * (t >> 2) + 4 * q + (t & 3)) * 16 + p];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    const int c = 16 * t + p;
    wrA[t][0] = c < DE ? a.Wr[(2 * q + 0) * DE + c] : 0.f;
    wrA[t][1] = c < DE ? a.Wr[(2 * q + 1) * DE + c] : 0.f;
    const int cb = 16 * t + 4 * q;   // scalar loads: parameter tensors need not be 16-byte aligned
    brv[t] = (cb < DE) ? make_float4(a.br[cb], a.br[cb + 1], a.br[cb + 2], a.br[cb + 3])
                       : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (KVL) __syncthreads();

  const int ntile = (N + 15) / 16;
  int nrows = 0;
  for (int li = 0; li < 4; ++li) nrows += (lg * 16 + wave + 4 * li < N) ? 1 : 0;
  const int total = nrows * ntile;

  // e tiles in flight per wave.  A De <= 16 tile is 0.5 - 1 KB, so narrow tiles travel PFD
  // iterations ahead in a ring of register sets (more bytes in flight per CU).  The ring is indexed
  // statically (loop unrolled by PFD); the iterations that pad the last group re-run the last tile
  // with every global write predicated off, which keeps the body straight-line (exact waits).
  constexpr int PFD = (DE <= 16 && KVL) ? 4 : 1;   // measured: +3.5 % at De = 8, N = 120; without K/V in LDS the extra registers spill
  TileRegs<DE> ring[PFD];
  auto prefetch = [&](TileRegs<DE>& tr, int it) {
    if (PFD > 1) it = min(it, total - 1);
    const int l = lg * 16 + wave + 4 * (it / ntile), m0 = (it % ntile) * 16;
    const size_t pair0 = ((size_t)b * N + l) * N + m0;
    tile_gload<DE>(tr, e_in + pair0 * DE, lane, FULL ? 16 : min(16, N - m0));
  };
  if (total > 0) {
#pragma unroll
    for (int k = 0; k < PFD; ++k) prefetch(ring[k], k);
  }

  float Qf[16], mx[2], sum[2], O[16];
  auto step = [&](const int it_, TileRegs<DE>& tr) __attribute__((always_inline)) {
    const bool live = PFD == 1 || it_ < total;
    const int it = PFD == 1 ? it_ : min(it_, total - 1);
    const int li = it / ntile, mt = it % ntile;
    const int l = lg * 16 + wave + 4 * li, m0 = mt * 16, m = m0 + p;
    const bool valid = FULL ? true : (m < N);
    const int rows_valid = FULL ? 16 : min(16, N - m0);
    const size_t rowl = (size_t)b * N + l;
    const size_t pair0 = rowl * N + m0;
    if (mt == 0) {
      const float4* qp = KVL ? reinterpret_cast<const float4*>(qs + (wave + 4 * li) * QS_LD + q * 16)
                             : reinterpret_cast<const float4*>(a.qkvp + rowl * QKVP + q * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = qp[i]; Qf[4*i] = v.x; Qf[4*i+1] = v.y; Qf[4*i+2] = v.z; Qf[4*i+3] = v.w; }
      mx[0] = mx[1] = -3.0e38f; sum[0] = sum[1] = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) O[i] = 0.f;
    }
    // ---- stage this tile, start the next one's loads ----
    MaskRegs mr{make_float2(1.f, 1.f), 0};   // issued before the prefetch, consumed after the MFMAs
    mask_gload<ML>(a, mr, pair0 + (valid ? p : 0), q);
    // Memory order per step: [stores of tile it-1] then [loads of tile it+1]; the wait in front
    // of the next LDS staging therefore never covers a store younger than the loads it needs.
    float* tl = tl0 + (it_ & 1) * G::TILE_FLOATS;
    lds_sync();
    if (it_ > 0 && live) {   // stream out e' of the previous tile from the other buffer
      const int itp = it - 1, lp = lg * 16 + wave + 4 * (itp / ntile), m0p = (itp % ntile) * 16;
      tile_from_lds<DE>(tl0 + (itp & 1) * G::TILE_FLOATS, e_o + (((size_t)b * N + lp) * N + m0p) * DE,
                        lane, FULL ? 16 : min(16, N - m0p));
    }
    tile_lds_put<DE>(tl, tr, lane, rows_valid);
    if (PFD == 1) { if (it + 1 < total) prefetch(tr, it + 1); }
    else prefetch(tr, it_ + PFD);   // this slot's next tile (clamped past the end)
    lds_sync();
    float4 x[G::TILES];
#pragma unroll
    for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(tl, p, q, t);
    // ---- K/V fragments of key m ----
    float Kf[16], Vf[16], kadd = 0.f;
    {
      const int mc = valid ? m : 0;
      const float4* kp;
      const float4* vp;
      if (KVL) {
        kp = reinterpret_cast<const float4*>(kvs + mc * KV_LD + q * 16);
        vp = reinterpret_cast<const float4*>(kvs + mc * KV_LD + 64 + q * 16);
        kadd = kms[mc];
      } else {
        const size_t rowm = (size_t)b * N + mc;
        kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
        vp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 128 + q * 16);
        kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 kv = kp[i], vv = vp[i];
        Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
        Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
      }
    }
    // ---- norm_edge + [attention_gates | dense_edge_b] ----
    v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
    {
      ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
      acc = project<DE>(x, wA, acc);
    }
    // ---- scaled QK^T, clip, + E (egt_layers.py:79-86) ----
    float hh[2] = {0.f, 0.f}, xl[2] = {0.f, 0.f}, gl[2] = {0.f, 0.f};
    {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
      float ah = dot * a.scale;
      if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
      hh[j] = ah + acc[2 * j + 1];
      xl[j] = hh[j];
      gl[j] = acc[2 * j];
    }
    }
    apply_masks<ML>(a, kadd, mr, (pair0 + p) * BH, q, xl, gl);
    // ---- online softmax x gate, A.V (per lane) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float xv = xl[j];
      const float mn = valid ? fmaxf(mx[j], xv) : mx[j];
      const float alpha = __expf(mx[j] - mn);
      const float pe = valid ? __expf(xv - mn) : 0.f;
      mx[j] = mn;
      sum[j] = fmaf(sum[j], alpha, pe);
      const float av = gated ? pe * egt_sigmoid(gl[j]) : pe;
#pragma unroll
      for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(O[2 * k + j], alpha, av * Vf[2 * k + j]);
    }
    // ---- dense_edge_r + res_edge: e' = e + H_hat.Wr + br ----
    const float h0 = valid ? hh[0] : 0.f, h1 = valid ? hh[1] : 0.f;
#pragma unroll
    for (int t = 0; t < G::TILES; ++t) {
      v4f d = {brv[t].x, brv[t].y, brv[t].z, brv[t].w};
      d = MFMA(wrA[t][0], h0, d);
      d = MFMA(wrA[t][1], h1, d);
      const float4 ev = frag_read<DE>(tl, p, q, t);
      frag_write<DE>(tl, p, q, t, make_float4(ev.x + d[0], ev.y + d[1], ev.z + d[2], ev.w + d[3]));
    }
    if (it_ + 1 == total) {   // last tile of the wave: flush
      lds_sync();
      tile_from_lds<DE>(tl, e_o + pair0 * DE, lane, rows_valid);
    }
    if (mt == ntile - 1 && live) {
      // ---- merge the 16 key lanes (same q): max, then sums ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mr2 = row_max16(mx[j]);
        const float f = __expf(mx[j] - mr2);
        mx[j] = mr2;
        sum[j] = row_sum16(sum[j] * f);
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] *= f;
      }
      const float o = reduce16_keep_own(O, p);
      // element i = p: k = p>>1, j = p&1 -> head 2q + j
      const int k = p >> 1, j = p & 1;
      const float sj = j ? sum[1] : sum[0];
      const float vo = o / sj;
      if (k < a.DK) a.v_att[rowl * a.Dh + k * BH + 2 * q + j] = vo;
      // the row's Q is dead (read at mt == 0 by this wave only): its slot keeps V_att for the epilogue
      if (KVL && a.epi) qs[(wave + 4 * li) * QS_LD + k * BH + 2 * q + j] = vo;
      if (p < 2) {
        float* st = a.stats + (rowl * BH + 2 * q + p) * 4;
        st[0] = p ? mx[1] : mx[0];
        st[1] = sj;
      }
    }
  };
  if (PFD == 1) {
    for (int it = 0; it < total; ++it) step(it, ring[0]);
  } else {
    for (int it0 = 0; it0 < total; it0 += PFD) {
#pragma unroll
      for (int k = 0; k < PFD; ++k) step(it0 + k, ring[k]);
    }
  }

  // ---- node-side epilogue (Dh = 64): the workgroup holds V_att of its 16 rows ----
  //   epi >= 1: h' = V_att.Wo + bo + h                      (dense_mha + res_mha, :136,140)
  //   epi == 2: qkv of the NEXT block = LN(h').Wqkv' + bqkv' (norm_mha + dense_qkv, :109,113), packed
  // Contraction order k = 4s + q on both MFMA operands; weights come straight from L2.
  if (KVL && a.epi) fwd_node_epilogue(a, sm, qs, b, lg, N, wave, p, q);
}

// ---------------------------------------------------------------- forward, narrow edge channels ---
// k_block_fwd walks (row, key tile) pairs one 16-pair tile at a time; for De <= 16 such a tile is
// 0.5 - 1 KB and a third of the time is the per-iteration skeleton (staging, LDS hand-offs, index
// arithmetic), the rest a chain of short dependent phases (see the ablation in DESIGN.md).  This
// variant turns the loop inside out: one iteration = ONE key tile for all FOUR rows of the wave.
// The K / V fragments and the key-mask add are fetched once per iteration and shared by the four
// rows, the four e tiles are staged with one pair of LDS hand-offs, and the four rows' LN ->
// projection -> softmax -> A.V -> dense_edge_r chains are independent, so the scheduler interleaves
// them (ILP x4 instead of one latency-bound chain).  Q stays in LDS (re-read per key tile), the
// running softmax state of the four rows lives in registers.  K/V of the graph in LDS (KVL) only;
// no mask tensors (the ML variants stay on k_block_fwd).
// NW = 4: 16 rows per workgroup, two workgroups per CU; NW = 8: 32 rows (two 16-row halves), ONE
// workgroup per CU -- same occupancy, but K/V of a graph up to N ~ 250 still fits in LDS and is staged
// once per 32 rows.
template <int DE, bool FULL, int NW, bool BF>
__global__ void __launch_bounds__(64 * NW, 2) k_block_fwd_r4(BlockArgs a) {
  seed_from_device(a);
  typedef typename EdgeT<BF>::type ET;   // element type of the edge tensors in HBM
  constexpr int RW = 4 * NW, NT = 64 * NW;   // rows / threads per workgroup
  using G = Geo<DE>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int hf = wave >> 2, wv = wave & 3;   // 16-row half of the workgroup, wave inside it
  const int N = a.N;
  const int lgroups = (N + RW - 1) / RW;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / lgroups, lg = wg % lgroups;
  float* tl = sm + wave * 4 * G::TILE_FLOATS;    // the wave's four tiles (one per row)
  float* kvs = sm + 4 * NW * G::TILE_FLOATS;     // [N][KV_LD]
  float* qs = kvs + N * KV_LD;                   // [RW][QS_LD]
  float* kms = qs + RW * QS_LD;                  // [N]
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;
  {
    const float* src = a.qkvp + (size_t)b * N * QKVP;
    for (int i = threadIdx.x; i < N * 32; i += NT) {
      const int row = i >> 5, f = i & 31;
      *reinterpret_cast<float4*>(kvs + row * KV_LD + f * 4) =
          *reinterpret_cast<const float4*>(src + (size_t)row * QKVP + 64 + f * 4);
    }
    for (int i = threadIdx.x; i < RW * 16; i += NT) {
      const int row = i >> 4, f = i & 15, l = min(lg * RW + row, N - 1);
      *reinterpret_cast<float4*>(qs + row * QS_LD + f * 4) =
          *reinterpret_cast<const float4*>(src + (size_t)l * QKVP + f * 4);
    }
    for (int i = threadIdx.x; i < N; i += NT)
      kms[i] = (a.km && a.km[(size_t)b * N + i] == 0) ? -EGT_NEG : 0.0f;
  }
  float wA[4 * G::TILES], wrA[G::TILES][2], c2r[4];
  float4 brv[G::TILES];
#pragma unroll
  for (int t = 0; t < 4 * G::TILES; ++t) wA[t] = a.pw[(16 * (t >> 2) + 4 * q + (t & 3)) * 16 + p];
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[G::DEP * 16 + 4 * q + r];
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    const int c = 16 * t + p;
    wrA[t][0] = c < DE ? a.Wr[(2 * q + 0) * DE + c] : 0.f;
    wrA[t][1] = c < DE ? a.Wr[(2 * q + 1) * DE + c] : 0.f;
    const int cb = 16 * t + 4 * q;
    brv[t] = (cb < DE) ? make_float4(a.br[cb], a.br[cb + 1], a.br[cb + 2], a.br[cb + 3])
                       : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  const int ntile = (N + 15) / 16;
  int nrows = 0;
  const int row0 = hf * 16 + wv;   // the wave's rows inside the workgroup: row0 + 4i
#pragma unroll
  for (int i = 0; i < 4; ++i) nrows += (lg * RW + row0 + 4 * i < N) ? 1 : 0;   // rows i < nrows exist
  if (nrows > 0) {
    const ET* e_in = reinterpret_cast<const ET*>(a.e);
    ET* e_o = reinterpret_cast<ET*>(a.e_out);
    size_t rowl[4];   // rows past the end alias the last real row: loaded, never computed or stored
#pragma unroll
    for (int i = 0; i < 4; ++i) rowl[i] = (size_t)b * N + lg * RW + row0 + 4 * min(i, nrows - 1);
    TileRegs<DE> tr[4];
    auto prefetch = [&](int mt) {
      const int m0 = mt * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        tile_gload<DE>(tr[i], e_in + (rowl[i] * N + m0) * DE, lane, FULL ? 16 : min(16, N - m0));
    };
    prefetch(0);
    float mx[4][2], sum[4][2], O[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mx[i][0] = mx[i][1] = -3.0e38f; sum[i][0] = sum[i][1] = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) O[i][k] = 0.f;
    }
    for (int mt = 0; mt < ntile; ++mt) {
      const int m0 = mt * 16, m = m0 + p;
      const bool valid = FULL ? true : (m < N);
      const int rows_valid = FULL ? 16 : min(16, N - m0);
      lds_sync();   // the e' tiles of the previous key tile have left the LDS tiles
#pragma unroll
      for (int i = 0; i < 4; ++i) tile_lds_put<DE>(tl + i * G::TILE_FLOATS, tr[i], lane, rows_valid);
      if (mt + 1 < ntile) prefetch(mt + 1);
      lds_sync();
      // ---- K/V fragments of key m: shared by the four rows ----
      float Kf[16], Vf[16];
      const int mc = valid ? m : 0;
      {
        const float4* kp = reinterpret_cast<const float4*>(kvs + mc * KV_LD + q * 16);
        const float4* vp = reinterpret_cast<const float4*>(kvs + mc * KV_LD + 64 + q * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 kv = kp[i], vv = vp[i];
          Kf[4*i] = kv.x; Kf[4*i+1] = kv.y; Kf[4*i+2] = kv.z; Kf[4*i+3] = kv.w;
          Vf[4*i] = vv.x; Vf[4*i+1] = vv.y; Vf[4*i+2] = vv.z; Vf[4*i+3] = vv.w;
        }
      }
      const float kadd = kms[mc];
      const MaskRegs mr{make_float2(1.f, 1.f), 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < nrows) {
          float* tli = tl + i * G::TILE_FLOATS;
          float4 x[G::TILES];
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) x[t] = frag_read<DE>(tli, p, q, t);
          // ---- norm_edge + [attention_gates | dense_edge_b] ----
          ln_frags<DE>(x, q, a.ln_eps, (a.flags & EGT_BF_NO_EDGE_LN) == 0);
          v4f acc = {c2r[0], c2r[1], c2r[2], c2r[3]};
          acc = project<DE>(x, wA, acc);
          // ---- scaled QK^T, clip, + E (egt_layers.py:79-86); Q of the row from LDS ----
          float Qf[16];
          {
            const float4* qp = reinterpret_cast<const float4*>(qs + (row0 + 4 * i) * QS_LD + q * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float4 v = qp[u]; Qf[4*u] = v.x; Qf[4*u+1] = v.y; Qf[4*u+2] = v.z; Qf[4*u+3] = v.w; }
          }
          float hh[2], xl[2], gl[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
            float ah = dot * a.scale;
            if (clip) ah = fminf(fmaxf(ah, a.clip_lo), a.clip_hi);
            hh[j] = ah + acc[2 * j + 1];
            xl[j] = hh[j];
            gl[j] = acc[2 * j];
          }
          apply_masks<false>(a, kadd, mr, (rowl[i] * N + m0 + p) * BH, q, xl, gl);
          // ---- online softmax x gate, A.V (per lane) ----
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float xv = xl[j];
            const float mn = valid ? fmaxf(mx[i][j], xv) : mx[i][j];
            const float alpha = __expf(mx[i][j] - mn);
            const float pe = valid ? __expf(xv - mn) : 0.f;
            mx[i][j] = mn;
            sum[i][j] = fmaf(sum[i][j], alpha, pe);
            const float av = gated ? pe * egt_sigmoid(gl[j]) : pe;
#pragma unroll
            for (int k = 0; k < 8; ++k) O[i][2 * k + j] = fmaf(O[i][2 * k + j], alpha, av * Vf[2 * k + j]);
          }
          // ---- dense_edge_r + res_edge: e' = e + H_hat.Wr + br ----
          const float h0 = valid ? hh[0] : 0.f, h1 = valid ? hh[1] : 0.f;
#pragma unroll
          for (int t = 0; t < G::TILES; ++t) {
            v4f d = {brv[t].x, brv[t].y, brv[t].z, brv[t].w};
            d = MFMA(wrA[t][0], h0, d);
            d = MFMA(wrA[t][1], h1, d);
            const float4 ev = frag_read<DE>(tli, p, q, t);
            frag_write<DE>(tli, p, q, t, make_float4(ev.x + d[0], ev.y + d[1], ev.z + d[2], ev.w + d[3]));
          }
        }
      }
      lds_sync();   // stream out the four e' tiles
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nrows) tile_from_lds<DE>(tl + i * G::TILE_FLOATS, e_o + (rowl[i] * N + m0) * DE, lane, rows_valid);
    }
    // ---- per row: merge the 16 key lanes (same q): max, then sums ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nrows) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float mr2 = row_max16(mx[i][j]);
          const float f = __expf(mx[i][j] - mr2);
          mx[i][j] = mr2;
          sum[i][j] = row_sum16(sum[i][j] * f);
#pragma unroll
          for (int k = 0; k < 8; ++k) O[i][2 * k + j] *= f;
        }
        const float o = reduce16_keep_own(O[i], p);
        const int k = p >> 1, j = p & 1;   // element p: k = p>>1, j = p&1 -> head 2q + j
        const float sj = j ? sum[i][1] : sum[i][0];
        const float vo = o / sj;
        if (k < a.DK) a.v_att[rowl[i] * a.Dh + k * BH + 2 * q + j] = vo;
        if (a.epi) qs[(row0 + 4 * i) * QS_LD + k * BH + 2 * q + j] = vo;   // the row's Q is dead from here on
        if (p < 2) {
          float* st = a.stats + (rowl[i] * BH + 2 * q + p) * 4;
          st[0] = p ? mx[i][1] : mx[i][0];
          st[1] = sj;
        }
      }
    }
  }
  // each 16-row half runs the 4-wave epilogue on its own rows (staging area hs = its own tiles)
  if (a.epi) fwd_node_epilogue(a, sm + hf * 16 * QS_LD, qs + hf * 16 * QS_LD, b, lg * (RW / 16) + hf, N, wv, p, q);
}

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
  for (int i = threadIdx.x; i < nl * 40; i += 256) {
    const int r = i / 40, f = i % 40;
    const size_t rowl = (size_t)b * N + l_begin + r;
    if (a.pro && f >= 16 && f < 32) continue;   // dV_att comes from the prologue below
    const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                     : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                              : a.stats + rowl * 32 + (f - 32) * 4;
    float4 v = *reinterpret_cast<const float4*>(src);
    if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
    *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
  }
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
// fp32 edge tensors, no mask tensors, N a multiple of 16, De a multiple of 16.
// MM = EGT_MM_BF16X3 (opt-in: EGT_BWD_MATMUL=bf16x3): the three channel contractions of a tile (P1 projections,
// P2 dH_ext, P5 d ehat: 48 of the 80 fp32 MFMAs) run as 3-term bfloat16 split products on the bf16 matrix pipe
// (20 MFMAs of 16 cycles; per-product error 2^-16, fp32 accumulate); the weight-gradient contractions over the
// pair axis (P4) and everything else stay exact fp32.  Same LDS footprint: a bf16 hi + lo pair is as large as the
// fp32 value it replaces.
template <int DE, int MM>
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
  float* et0 = sm + wave * PW;               // e / xhat tile, two buffers (row parity)
  float* dt = et0 + 2 * G::TILE_FLOATS;      // de' tile
  float* sc1 = dt + G::TILE_FLOATS;
  float* sc2 = sc1 + 256;
  constexpr int AREA = 4 * PW > BWD_PRO_WS ? 4 * PW : BWD_PRO_WS;   // per-wave tiles; also the prologue's scratch
  float* qd = sm + AREA;                     // [TL][QD_LD]
  float* wsA = qd + TL * QD_LD;              // prologue weights   wA[4t+u]
  float* wsB = wsA + WSLAB;                  // dH_ext weights     wrB[4t+u]
  float* wsD = wsB + WSLAB;                  // d(ehat) weights    wD[t][s]
  // LDS byte address of the wave's e buffers (the kernel's only LDS object is the dynamic array: offset 0)
  const unsigned et_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)et0);
  const unsigned off0 = dma_lane_offset<DE>(lane);
  for (int i = threadIdx.x; i < nl * 40; i += 256) {
    const int r = i / 40, f = i % 40;
    const size_t rowl = (size_t)b * N + l_begin + r;
    if (a.pro && f >= 16 && f < 32) continue;   // dV_att comes from the prologue below
    const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                     : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                              : a.stats + rowl * 32 + (f - 32) * 4;
    float4 v = *reinterpret_cast<const float4*>(src);
    if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
    *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
  }
  PSTAMP(0);
  if (a.pro) {
    __syncthreads();
    bwd_node_prologue<DE>(a, sm, qd, b, l_begin, wg);
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
  } else
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
  __syncthreads();   // the prologue's scratch (= the tile area) is dead from here: DMA may land in it
  PSTAMP(2);

  const int ntile = N / 16;
  for (int mt = wave; mt < ntile; mt += 4) {
    const int m0 = mt * 16, m = m0 + p;
    // first e tile of this key tile: in flight while K / V are fetched
    tile_dma<DE>(et_lds, e_in + (((size_t)b * N + l_begin) * N + m0) * DE, off0);
    float Kf[16], Vf[16], dKa[16], dVa[16];
    const size_t rowm = (size_t)b * N + m;
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
    for (int l = l_begin; l < l_end; ++l) {
      const int li = l - l_begin;
      const size_t rowl = (size_t)b * N + l;
      const size_t pair0 = rowl * N + m0;
      float* et = et0 + (li & 1) * G::TILE_FLOATS;
      MaskRegs mr{make_float2(1.f, 1.f), 0};
      // ---- de'(l): requested now, consumed after P1 ----
      TileRegs<DE> td;
      tile_gload<DE>(td, dey_in + pair0 * DE, lane, 16);
      // ---- e(l) has been in flight for a whole iteration: retire it.  Younger operations of this wave:
      // row l-1's dQ-partial store and its NI de stores, then the NI de' loads just issued ----
      if (li == 0) vm_wait<0>(); else vm_wait<2 * ((G::NF4 + 63) / 64) + 1>();
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
      tile_lds_put<DE>(dt, td, lane, 16);   // (the compiler's own vmcnt wait for de' sits here)
      lds_sync();
      // ---- e(l+1) -> the other e buffer (its last reader, row l-1's P5, retired its LDS reads) ----
      if (l + 1 < l_end)
        tile_dma<DE>(et_lds + (unsigned)(((li + 1) & 1) * G::TILE_FLOATS * 4), e_in + (pair0 + (size_t)N) * DE, off0);
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
            *reinterpret_cast<float4*>(orow + 16 * t) = o;
          }
        }
        lds_sync();   // the tile reads above retire before the next iteration overwrites dt / DMAs into et
      }
      SCHED_FENCE();
    }
    float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 0) * 4 + q) * 16);
    float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + m) * 2 + 1) * 4 + q) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ko[i] = make_float4(dKa[4*i], dKa[4*i+1], dKa[4*i+2], dKa[4*i+3]);
      vo[i] = make_float4(dVa[4*i], dVa[4*i+1], dVa[4*i+2], dVa[4*i+3]);
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
  for (int i = threadIdx.x; i < nl * 40; i += 256) {
    const int r = i / 40, f = i % 40;
    const size_t rowl = (size_t)b * N + l_begin + r;
    if (a.pro && f >= 16 && f < 32) continue;   // dV_att comes from the prologue below
    const float* src = f < 16 ? a.qkvp + rowl * QKVP + f * 4
                     : f < 32 ? a.dvp + rowl * 64 + (f - 16) * 4
                              : a.stats + rowl * 32 + (f - 32) * 4;
    float4 v = *reinterpret_cast<const float4*>(src);
    if (f >= 32) v.y = 1.0f / v.y;   // softmax row sum -> reciprocal
    *reinterpret_cast<float4*>(qd + r * QD_LD + f * 4) = v;
  }
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

// ================================================================ host glue ====


// Run-time switches (read ONCE per process: the launch path does no getenv()) -- the complete list, see README.md:
//   EGT_NO_NARROW / EGT_NO_NARROW_FWD / EGT_NO_NARROW_BWD: De = 8 falls back from the De = 8 pair kernels (egt_narrow.hip) to the
//     MFMA-tile kernels (tests exercise both);  EGT_BWD_MATMUL=bf16x3: the backward's channel contractions as 3-term bf16 split
//     products (opt-in; default exact fp32);  EGT_BWD_TL: query rows per backward workgroup (tests / sweeps).
struct EgtBlockEnv {
  bool no_narrow_fwd, no_narrow_bwd;
  int bwd_mm;
};
static bool env_flag_raw(const char* name) {
  const char* v = getenv(name);
  return v && v[0] && v[0] != '0';
}
static const EgtBlockEnv& block_env() {
  static const EgtBlockEnv e = [] {
    EgtBlockEnv v{};
    v.no_narrow_fwd = env_flag_raw("EGT_NO_NARROW_FWD") || env_flag_raw("EGT_NO_NARROW");
    v.no_narrow_bwd = env_flag_raw("EGT_NO_NARROW_BWD") || env_flag_raw("EGT_NO_NARROW");
    const char* mm = getenv("EGT_BWD_MATMUL");
    v.bwd_mm = (mm && !strcmp(mm, "bf16x3")) ? EGT_MM_BF16X3 : EGT_MM_F32;
    return v;
  }();
  return e;
}
// compute units of the current device (dispatch decisions that depend on whether a launch fills the chip)
int egt_device_cus() {
  static const int n = [] {
    int dev = 0, cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    return cu > 0 ? cu : 256;
  }();
  return n;
}

static int block_check(const egt_block_desc* d, bool report) {
#define BAD(code, ...) do { if (report) egt_set_error(__VA_ARGS__); return (code); } while (0)
  if (!d) BAD(EGT_E_NULL, "desc is NULL");
  if (d->dtype != EGT_F32 && d->dtype != EGT_BF16) BAD(EGT_E_DTYPE, "dtype must be EGT_F32 or EGT_BF16 (got %d)", d->dtype);
  if (d->B <= 0 || d->N <= 0) BAD(EGT_E_SHAPE, "B and N must be positive");
  if (d->H != BH) BAD(EGT_E_SHAPE, "fused block is built for num_heads=8 (got %d)", d->H);
  if (d->d < 1 || d->d > 8) BAD(EGT_E_SHAPE, "fused block covers per-head dim <= 8 (got %d)", d->d);
  switch (d->De) {
    case 8: case 16: case 32: case 48: case 64: break;
    default: BAD(EGT_E_SHAPE, "fused block covers edge_width in {8,16,32,48,64} (got %d)", d->De);
  }
  if ((size_t)d->B * d->N * d->N * d->H > 0xFFFFFFFFull) BAD(EGT_E_SHAPE, "B*N*N*H exceeds the 32-bit RNG counter");
  if ((d->flags & EGT_BF_SEED_DEVICE) && !d->seed_device) BAD(EGT_E_NULL, "EGT_BF_SEED_DEVICE set but seed_device is NULL");
  return EGT_OK;
#undef BAD
}

extern "C" int egt_block_supported(const egt_block_desc* d) { return block_check(d, false) == EGT_OK; }

static size_t al(size_t x) { return (x + 63) & ~(size_t)63; }  // in floats

struct BlockLayout {
  size_t v_att, stats, qkvp, pw_sv, saved_total;   // pw_sv: LN-folded edge weights, prepared by the forward, reused by the backward
  // workspace = [common: dvp dqp[2] dkvp[2]] + per layer [pw epart spart sbo wpart ered dqkv dhbuf]
  // (dqp / dkvp alternate by layer parity: the prologue of layer l-1 reads layer l's partials while
  //  other workgroups of that launch already write their own)
  size_t dvp, dqp, dkvp, dqp_sz, dkvp_sz, common_total;
  size_t pw, epart, spart, sbo, wpart, ered, dqkv, dhbuf, layer_total;
  int TL, NLR, nwg_bwd, EP;   // TL: query rows per backward workgroup
};

// Query rows per backward workgroup (<= 16: the MFMA tiles of the node-side prologue):
//  * equal groups: N = 150 is ten groups of 15 rather than nine of 16 and one of 6 -- same workgroup count, no short group
//    (config 3: 225 -> 216 us per launch; N = 120: 162 -> 156 us);
//  * a launch of at most one 16-row workgroup per CU (BASELINE config 4 as specified: B = 16, N = 120 -> 128 workgroups on
//    256 CUs) takes 8 rows per workgroup: twice the partial slots, each prologue on a half-filled tile, every CU busy
//    (pattern500k_n120, per launch: B = 16: 59.0 us at 16 rows, 43.3 at 8, 47.9 at 6; B = 32: 63.0 / 59.5 / 65.5).
//    Smaller groups never pay beyond that: the per-workgroup work that does not shrink with the rows (prologue, K / V tiles,
//    partial sums) takes over (B = 128, N = 150: 225 us at 16 rows, 253 at 12, 295 at 8).
// EGT_BWD_TL = 4 .. 16 overrides, for every De (tests, sweeps: tools/dbg/nrw_tlsweep.sh).
static int bwd_rows_per_wg(const egt_block_desc* d) {
  static const int forced = getenv("EGT_BWD_TL") ? atoi(getenv("EGT_BWD_TL")) : 0;
  const int groups = (d->N + BWD_TL - 1) / BWD_TL;
  if (forced >= 4 && forced <= BWD_TL) return forced;
  if (d->De != 8) {   // MFMA-tile kernels: two workgroups per CU
    // equal groups (16 whenever N is a multiple of 16); a launch that leaves slots empty takes more, shorter groups while they
    // still fit one round and keep 8 rows (ZINC-100K, B = 128, N = 37: 3 x 13 rows = 384 workgroups on 512 slots -> 4 x 10 rows:
    // k_block_bwd 73.6 -> 64.3 us; 5 x 8 rows = 640 workgroups is a second round: 92.9 us)
    const int slots = 2 * egt_device_cus();
    int g = slots / (d->B > 0 ? d->B : 1);
    if (g > (d->N + 7) / 8) g = (d->N + 7) / 8;
    if (g < groups) g = groups;
    return (d->N + g - 1) / g;
  }
  if (d->B * groups <= egt_device_cus()) return d->N > 8 ? 8 : BWD_TL;
  return (d->N + groups - 1) / groups;
}

static BlockLayout layout(const egt_block_desc* d) {
  BlockLayout L{};
  const size_t rows = (size_t)d->B * d->N;
  const int Dh = d->d * d->H;
  const int TILES = (d->De + 15) / 16, DEP = TILES * 16;
  L.EP = DEP * 16 + 16 + DEP * 16;
  size_t o = 0;
  L.v_att = o; o += al(rows * Dh);
  L.stats = o; o += al(rows * BH * 4);
  L.qkvp = o; o += al(rows * QKVP);
  L.pw_sv = o; o += al((size_t)DEP * 16 + 16);
  L.saved_total = o;
  L.TL = bwd_rows_per_wg(d);
  L.NLR = (d->N + L.TL - 1) / L.TL;
  L.nwg_bwd = d->B * L.NLR;
  o = 0;
  L.dvp = o; o += al(rows * 64);
  L.dqp_sz = al(rows * 64 * (size_t)((d->N + 15) / 16));
  L.dkvp_sz = al((size_t)d->B * L.NLR * d->N * 128);
  L.dqp = o; o += 2 * L.dqp_sz;
  L.dkvp = o; o += 2 * L.dkvp_sz;
  L.common_total = o;
  o = 0;
  L.pw = o; o += al((size_t)DEP * 16 + 16);
  L.epart = o; o += al((size_t)L.nwg_bwd * L.EP);
  {
    const size_t nmax = (size_t)(L.nwg_bwd > d->B * ((d->N + NODE_RC - 1) / NODE_RC) ? L.nwg_bwd
                                                                                        : d->B * ((d->N + NODE_RC - 1) / NODE_RC));
    L.spart = o; o += al(nmax * (5 * Dh));
    L.sbo = o; o += al(nmax * Dh);
  }
  L.wpart = o; o += al((size_t)egt_node_wgrad_chunks((int)rows) * (Dh * 3 * Dh + Dh * Dh));
  L.ered = o; o += al(L.EP);
  L.dqkv = o; o += al(rows * 3 * Dh);
  L.dhbuf = o; o += al(rows * Dh);
  L.layer_total = o;
  return L;
}

// workspace pointers of one layer: `wc` = common region, `wl` = that layer's region
static void bind_ws(const BlockLayout& L, BlockArgs& a, float* wc, float* wl, int parity = 0) {
  a.dvp = wc + L.dvp; a.dqp = wc + L.dqp + parity * L.dqp_sz; a.dkvp = wc + L.dkvp + parity * L.dkvp_sz;
  a.epart = wl + L.epart; a.spart = wl + L.spart; a.sbo = wl + L.sbo; a.wpart = wl + L.wpart;
  a.spart_n = a.sbo_n = a.B * ((a.N + NODE_RC - 1) / NODE_RC);   // k_node_bwd's workgroups (prologue path overrides)
  a.ered = wl + L.ered; a.dqkv_sv = wl + L.dqkv;
  a.TL = L.TL; a.NLR = L.NLR; a.NQP = 1;
  a.xcd = 1;
}

extern "C" size_t egt_block_saved_bytes(const egt_block_desc* d) {
  if (block_check(d, false)) return 0;
  return layout(d).saved_total * sizeof(float);
}
extern "C" size_t egt_block_workspace_bytes(const egt_block_desc* d) {
  if (block_check(d, false)) return 0;
  const BlockLayout L = layout(d);
  return (L.common_total + L.layer_total) * sizeof(float);
}

static int fill_block(const egt_block_desc* d, const egt_block_params* p, BlockArgs& a) {
  int rc = block_check(d, true);
  if (rc) return rc;
  if (!p) EGT_FAIL(EGT_E_NULL, "params is NULL");
  const void* const* pp = reinterpret_cast<const void* const*>(p);
  const bool gated = (d->flags & EGT_BF_GATE) != 0;
  for (int i = 0; i < 14; ++i) {
    if (!gated && (i == 2 || i == 3)) continue;
    if (!pp[i]) EGT_FAIL(EGT_E_NULL, "block parameter #%d is NULL", i);
  }
  a = BlockArgs{};
  a.B = d->B; a.N = d->N; a.De = d->De; a.DK = d->d; a.Dh = d->d * d->H;
  a.flags = d->flags;
  a.bf16 = d->dtype == EGT_BF16;
  a.clip_lo = d->clip_lo; a.clip_hi = d->clip_hi;
  a.scale = 1.0f / sqrtf((float)d->d);
  a.ln_eps = d->ln_eps;
  a.rm_thr = egt_threshold24(d->random_mask_prob);
  a.s0 = (uint32_t)(d->seed & 0xFFFFFFFFull);
  a.s1 = (uint32_t)(d->seed >> 32);
  a.sd = (d->flags & EGT_BF_SEED_DEVICE) ? (const uint32_t*)d->seed_device : nullptr;
  a.ne_g = (const float*)p->norm_edge_gamma; a.ne_b = (const float*)p->norm_edge_beta;
  a.Wg = (const float*)p->attention_gates_kernel; a.bg = (const float*)p->attention_gates_bias;
  a.We = (const float*)p->dense_edge_b_kernel; a.be = (const float*)p->dense_edge_b_bias;
  a.nm_g = (const float*)p->norm_mha_gamma; a.nm_b = (const float*)p->norm_mha_beta;
  a.Wqkv = (const float*)p->dense_qkv_kernel; a.bqkv = (const float*)p->dense_qkv_bias;
  a.Wo = (const float*)p->dense_mha_kernel; a.bo = (const float*)p->dense_mha_bias;
  a.Wr = (const float*)p->dense_edge_r_kernel; a.br = (const float*)p->dense_edge_r_bias;
  return EGT_OK;
}

static void bind_common(const egt_block_desc* d, BlockArgs& a, const void* h, const void* e,
                        const uint8_t* km, const void* M, const uint8_t* rm, float* saved, float* ws) {
  const BlockLayout L = layout(d);
  a.h = (const float*)h; a.e = (const float*)e; a.km = km;
  a.M = (d->flags & EGT_BF_ATTN_MASK) ? (const float*)M : nullptr;
  a.rm = nullptr; a.rng_rm = 0;
  if ((d->flags & EGT_BF_TRAINING) && d->random_mask_prob > 0.0f) {
    if (rm) a.rm = rm; else a.rng_rm = 1;
  }
  a.v_att = saved + L.v_att; a.stats = saved + L.stats; a.qkvp = saved + L.qkvp;
  bind_ws(L, a, ws, ws + L.common_total);
  a.pw = saved + L.pw_sv;
  a.prep = 1;
}

#define DISPATCH_BDE(De, CALL)                        \
  switch (De) {                                       \
    case 8: { constexpr int DE = 8; CALL; } break;    \
    case 16: { constexpr int DE = 16; CALL; } break;  \
    case 32: { constexpr int DE = 32; CALL; } break;  \
    case 48: { constexpr int DE = 48; CALL; } break;  \
    default: { constexpr int DE = 64; CALL; } break;  \
  }

// The node side inside the pair kernels (fwd_node_epilogue / bwd_node_prologue): H = 8 heads of DK <= 8 channels, i.e. node width
// Dh = 8 DK <= 64 as a zero-padded 64-wide row; the prologue reads Wo / Wqkv rows as 16-byte pieces.
static bool node_fused_ok(const BlockArgs& a) {
  return a.Dh == BH * a.DK && a.DK >= 1 && a.DK <= 8 &&
         ((reinterpret_cast<uintptr_t>(a.Wo) | reinterpret_cast<uintptr_t>(a.Wqkv)) & 15) == 0;
}

// Forward of one block.  `skip_pre`: qkvp (and pw) of this block were already produced (by the
// previous block's epilogue / k_edge_prep).  a.epi is the epilogue the caller would like; the
// value actually used is returned (0 when the geometry is outside the epilogue's cover, in
// which case k_node_post runs and the next block needs its own k_node_pre).
template <int DE>
static int launch_fwd(BlockArgs& a, hipStream_t st, bool skip_pre) {
  const int lgroups = (a.N + 15) / 16;
  if (!skip_pre) egt_node_launch_pre(a, st);   // norm_mha + dense_qkv (packed) [+ edge-weight prep]
  const size_t lds_tiles = (size_t)8 * Geo<DE>::TILE_FLOATS * 4;
  const size_t lds_kv = ((size_t)a.N * KV_LD + 16 * QS_LD + a.N) * 4;
  const bool kvl = lds_tiles + lds_kv <= 80 * 1024 - 512;   // two workgroups per CU keep their K/V in LDS
  const int epi_req = a.epi;
  if (!(kvl && node_fused_ok(a))) a.epi = 0;
  const bool ml = a.M != nullptr || a.rm != nullptr;
  a.guard = 0;   // (the always-taken phase branches of the kernels only shape hipcc's scheduling regions)
  const dim3 grid(a.B * lgroups), block(256);
  const size_t lds = lds_tiles + (kvl ? lds_kv : 0);
#define FWD_VARIANT_T(KVL_, ML_, FULL_, BF_)                                                           \
  do {                                                                                                 \
    EGT_MAX_LDS_ONCE(k_block_fwd<DE, KVL_, ML_, FULL_, BF_>);                 \
    EGT_LAUNCH("k_block_fwd", (k_block_fwd<DE, KVL_, ML_, FULL_, BF_>), grid, block, lds, st, a); \
  } while (0)
#define FWD_VARIANT(KVL_, ML_, FULL_)                                                                  \
  do { if (a.bf16) FWD_VARIANT_T(KVL_, ML_, FULL_, true); else FWD_VARIANT_T(KVL_, ML_, FULL_, false); } while (0)
  const bool full = (a.N % 16) == 0;
  // narrow edge channels: four rows per iteration (k_block_fwd_r4).  16-row workgroups when K/V + tiles fit
  // twice in a CU, else 32-row workgroups (one per CU) as long as K/V fits at all
  const size_t lds_r4 = (size_t)16 * Geo<DE>::TILE_FLOATS * 4 + lds_kv;
  const size_t lds_r8 = (size_t)32 * Geo<DE>::TILE_FLOATS * 4 + ((size_t)a.N * KV_LD + 32 * QS_LD + a.N) * 4;
  const bool r4 = lds_r4 <= 80 * 1024 - 512, r8 = !r4 && lds_r8 <= 156 * 1024;
  bool narrow = false;
  if constexpr (DE == 8) {   // VALU pair kernel (egt_narrow.hip): lane = (row, head/channel pair), 4 key quarters per workgroup
    if (!ml && !block_env().no_narrow_fwd) {
      narrow = true;
      if (!node_fused_ok(a)) a.epi = 0; else a.epi = epi_req;
      egt_narrow_launch_fwd(a, st);
    }
  }
  if constexpr (DE <= 16) {
    if (!narrow) {   // (wider channels would not fit the four rows' state in 256 VGPRs: not instantiated)
    narrow = !ml && (r4 || r8);
    if (narrow) {
    if (!node_fused_ok(a)) a.epi = 0; else a.epi = epi_req;
#define R4_LAUNCH_T(FULL_, NW_, BF_)                                                                              \
  do {                                                                                                            \
    EGT_MAX_LDS_ONCE(k_block_fwd_r4<DE, FULL_, NW_, BF_>); \
    EGT_LAUNCH("k_block_fwd", (k_block_fwd_r4<DE, FULL_, NW_, BF_>), dim3(a.B * ((a.N + 4 * NW_ - 1) / (4 * NW_))), dim3(64 * NW_), \
               NW_ == 4 ? lds_r4 : lds_r8, st, a);                                                                \
  } while (0)
#define R4_LAUNCH(FULL_, NW_) do { if (a.bf16) R4_LAUNCH_T(FULL_, NW_, true); else R4_LAUNCH_T(FULL_, NW_, false); } while (0)
    if (r4) { if (full) R4_LAUNCH(true, 4); else R4_LAUNCH(false, 4); }
    else { if (full) R4_LAUNCH(true, 8); else R4_LAUNCH(false, 8); }
#undef R4_LAUNCH
#undef R4_LAUNCH_T
    }
    }
  }
  if (narrow) {}
  else if (full && kvl && !ml) FWD_VARIANT(true, false, true);      // the headline variant
  else if (kvl) { if (ml) FWD_VARIANT(true, true, false); else FWD_VARIANT(true, false, false); }
  else { if (ml) FWD_VARIANT(false, true, false); else FWD_VARIANT(false, false, false); }
#undef FWD_VARIANT_T
#undef FWD_VARIANT
  if (a.epi == 0) egt_node_launch_post(a, st);  // dense_mha + res_mha
  return a.epi;
}

// Backward of one block.  `top`: first block of the chain (its dV_att / delta come from an own
// launch); otherwise they were produced by the node kernel of the block above.  `below`: the
// next block of the chain (NULL at the bottom), whose dV_att / delta this block's node kernel
// produces.  GEMM-shaped weight gradients and all partial reductions are left to the caller.
template <int DE>
static void launch_bwd(BlockArgs& a, const BlockLayout& L, hipStream_t st, bool top, BlockArgs* below, BlockArgs* above, bool fuse) {
  using GG = Geo<DE>;
  // node-side prologue inside the pair kernel (see bwd_node_prologue): v4 geometry with Dh = 64
  const bool ml = a.M != nullptr || a.rm != nullptr;
  // narrow edge channels without mask tensors run k_block_bwd_v4r, which (with the prologue) also covers ragged N
  const bool narrow_r = DE <= 16 && !ml;
  static_assert(DE % 16 == 0 || DE == 8, "edge widths of the pair kernels");
  const bool pro = fuse;   // node_fused_ok() of EVERY block of the chain (every backward kernel and the prologue take N that is not a multiple of 16)
  a.pro = 0;
  if (pro) {
    a.pro = top ? 1 : 2;
    a.sbo_n = L.nwg_bwd;
    if (!top) {
      a.up_h = above->h; a.up_nm_g = above->nm_g; a.up_Wqkv = above->Wqkv; a.up_dh_out = above->dh_out;
      a.up_dqp = above->dqp; a.up_dkvp = above->dkvp; a.up_dqkv_sv = above->dqkv_sv; a.up_spart = above->spart;
      above->spart_n = L.nwg_bwd;
    }
    if (a.prep) egt_node_launch_prep(&a, 1, st);   // single-block call: the LN-folded edge weights of this layer
  } else if (top) {
    egt_node_launch_bwd(a, &a, false, st);  // dV_att (packed), delta, dbo sums [+ edge-weight prep]
  }
  constexpr int PW = 3 * GG::TILE_FLOATS + 256 + 192;
  const bool full = (a.N % 16) == 0;
  {
    {
      const size_t lds_v4 = ((size_t)(4 * PW > BWD_PRO_WS ? 4 * PW : BWD_PRO_WS) + (size_t)BWD_TL * QD_LD + 3 * ((GG::TILES + 1) / 2) * 512) * 4;   // slabs padded to whole 32-channel steps (bf16 operands)
      a.NQP = (a.N + 15) / 16;
      a.guard = 0;   // (the always-taken phase branches of the kernels only shape hipcc's scheduling regions)
#define V4_VARIANT_R(ML_, BF_, RAG_)                                                                       \
  do {                                                                                                 \
    EGT_MAX_LDS_ONCE(k_block_bwd_v4<DE, ML_, BF_, RAG_>);                      \
    EGT_LAUNCH("k_block_bwd", (k_block_bwd_v4<DE, ML_, BF_, RAG_>), dim3(L.nwg_bwd), dim3(256), lds_v4, st, a); \
  } while (0)
#define V4_VARIANT(ML_, BF_) do { if (full) V4_VARIANT_R(ML_, BF_, false); else V4_VARIANT_R(ML_, BF_, true); } while (0)
      if constexpr (DE == 8) {
        // De = 8 pair kernel (egt_narrow.hip: k_narrow_bwd_m, three workgroups per CU), ragged N included.  Measured at config 3
        // against v4r / the round-2 quad-lane kernel (two waves per SIMD both): bf16 edge tensors 226 vs 314 us, fp32 243 vs 320 us
        if (narrow_r && !block_env().no_narrow_bwd) {
          egt_narrow_launch_bwd(a, L.nwg_bwd, st);
          goto pair_done;
        }
      }
      if constexpr (DE <= 16) {
        if (narrow_r) {   // narrow edge channels: R rows per iteration, ragged N included
          constexpr int RR = 2;   // rows per iteration: four spill (the rows' carried state + prefetch exceed 256 VGPRs)
          constexpr int PWR = RR * (2 * GG::TILE_FLOATS + 256 + 192);
          const size_t lds_r = ((size_t)(4 * PWR > BWD_PRO_WS ? 4 * PWR : BWD_PRO_WS) + (size_t)BWD_TL * QD_LD + 3 * GG::TILES * 256 + 3 * 2048 + 4) * 4;
          if (a.bf16) {
            EGT_MAX_LDS_ONCE(k_block_bwd_v4r<DE, true, RR>);
            EGT_LAUNCH("k_block_bwd", (k_block_bwd_v4r<DE, true, RR>), dim3(L.nwg_bwd), dim3(256), lds_r, st, a);
          } else {
            EGT_MAX_LDS_ONCE(k_block_bwd_v4r<DE, false, RR>);
            EGT_LAUNCH("k_block_bwd", (k_block_bwd_v4r<DE, false, RR>), dim3(L.nwg_bwd), dim3(256), lds_r, st, a);
          }
          goto pair_done;
        }
      }
      if constexpr (DE >= 32) {
        if (full && !ml && !a.bf16) {   // LDS-DMA staged e tiles (k_block_bwd_v5)
          const bool x3 = block_env().bwd_mm == EGT_MM_BF16X3;
          EGT_MAX_LDS_ONCE(k_block_bwd_v5<DE, 0>);
          EGT_MAX_LDS_ONCE(k_block_bwd_v5<DE, EGT_MM_BF16X3>);
          if (x3) EGT_LAUNCH("k_block_bwd", (k_block_bwd_v5<DE, EGT_MM_BF16X3>), dim3(L.nwg_bwd), dim3(256), lds_v4, st, a);
          else EGT_LAUNCH("k_block_bwd", (k_block_bwd_v5<DE, 0>), dim3(L.nwg_bwd), dim3(256), lds_v4, st, a);
          goto pair_done;
        }
      }
      if (a.bf16) { if (ml) V4_VARIANT(true, true); else V4_VARIANT(false, true); }
      else if (ml) V4_VARIANT(true, false);
      else V4_VARIANT(false, false);
#undef V4_VARIANT
#undef V4_VARIANT_R
    }
  }
pair_done:
  if (!pro) egt_node_launch_bwd(a, below, true, st);   // dQKV -> dh, bias/LN sums; dV_att + delta of the block below
  else if (!below) egt_node_launch_bwd(a, nullptr, true, st);   // bottom of the chain: only dQKV -> dh is left
}


extern "C" int egt_block_fwd(const egt_block_desc* desc, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, const uint8_t* rand_mask, void* h_out,
                             void* e_out, void* saved, void* workspace, void* stream) {
  BlockArgs a;
  int rc = fill_block(desc, params, a);
  if (rc) return rc;
  if (!h || !e || !h_out || !e_out || !saved || !workspace)
    EGT_FAIL(EGT_E_NULL, "h/e/h_out/e_out/saved/workspace is NULL");
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  bind_common(desc, a, h, e, key_mask, attn_mask, rand_mask, (float*)saved, (float*)workspace);
  a.h_out = (float*)h_out; a.e_out = (float*)e_out;
  a.epi = 1;
  DISPATCH_BDE(desc->De, launch_fwd<DE>(a, (hipStream_t)stream, false));
  EGT_HIP_LAUNCH_CHECK("egt_block_fwd");
  return EGT_OK;
}

extern "C" int egt_block_bwd(const egt_block_desc* desc, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, const uint8_t* rand_mask, const void* saved,
                             const void* d_h_out, const void* d_e_out, void* d_h, void* d_e,
                             const egt_block_params* grads, void* workspace, void* stream) {
  BlockArgs a;
  int rc = fill_block(desc, params, a);
  if (rc) return rc;
  if (!h || !e || !saved || !d_h_out || !d_e_out || !d_h || !d_e || !grads || !workspace)
    EGT_FAIL(EGT_E_NULL, "h/e/saved/d_h_out/d_e_out/d_h/d_e/grads/workspace is NULL");
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  // every layer's dh' is read again after dh was written (deferred dWo contraction): no in-place dh
  if (d_h == d_h_out) EGT_FAIL(EGT_E_FLAGS, "d_h must not alias d_h_out (d_e may alias d_e_out)");
  const bool gated = (desc->flags & EGT_BF_GATE) != 0;
  {
    const void* const* gp = reinterpret_cast<const void* const*>(grads);
    for (int i = 0; i < 14; ++i) {
      if (!gated && (i == 2 || i == 3)) continue;
      if (!gp[i]) EGT_FAIL(EGT_E_NULL, "gradient pointer #%d is NULL", i);
    }
  }
  bind_common(desc, a, h, e, key_mask, attn_mask, rand_mask, (float*)saved, (float*)workspace);
  a.prep = 0;   // prepared by the forward, kept in `saved`
  a.dh_out = (const float*)d_h_out; a.de_out = (const float*)d_e_out;
  a.dh = (float*)d_h; a.de = (float*)d_e;
  a.g_ne_g = (float*)grads->norm_edge_gamma; a.g_ne_b = (float*)grads->norm_edge_beta;
  a.g_Wg = (float*)grads->attention_gates_kernel; a.g_bg = (float*)grads->attention_gates_bias;
  a.g_We = (float*)grads->dense_edge_b_kernel; a.g_be = (float*)grads->dense_edge_b_bias;
  a.g_nm_g = (float*)grads->norm_mha_gamma; a.g_nm_b = (float*)grads->norm_mha_beta;
  a.g_Wqkv = (float*)grads->dense_qkv_kernel; a.g_bqkv = (float*)grads->dense_qkv_bias;
  a.g_Wo = (float*)grads->dense_mha_kernel; a.g_bo = (float*)grads->dense_mha_bias;
  a.g_Wr = (float*)grads->dense_edge_r_kernel; a.g_br = (float*)grads->dense_edge_r_bias;
  const BlockLayout L = layout(desc);
  DISPATCH_BDE(desc->De, launch_bwd<DE>(a, L, (hipStream_t)stream, true, nullptr, nullptr, node_fused_ok(a)));
  egt_node_launch_wgrads(&a, 1, (hipStream_t)stream);
  egt_node_launch_reduce(&a, 1, L.nwg_bwd, L.EP, (hipStream_t)stream);  // partial sums + edge param grads
  EGT_HIP_LAUNCH_CHECK("egt_block_bwd");
  return EGT_OK;
}


// ============================================================ layer stack =====
// The model_height loop over attention blocks (graph_xformer_model_base.py:336-339) as ONE
// call per direction: Ly x {node_pre, block_fwd, node_post} enqueued back to back, and in
// backward the per-workgroup partial sums of ALL layers reduced by a single launch at the end
// (they are off the dh/de critical path).  Layer l draws its random mask from
// seed ^ golden * (l + 1).
static uint64_t layer_seed(uint64_t seed, int l) { return seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(l + 1)); }

struct StackLayout {
  size_t h_act, e_act, blk, saved_total;      // floats
  size_t ws_total;
  size_t h_sz, e_sz;
};

static StackLayout stack_layout(const egt_block_desc* d, int layers) {
  StackLayout S{};
  const BlockLayout L = layout(d);
  S.h_sz = al((size_t)d->B * d->N * d->d * d->H);
  S.e_sz = al((size_t)d->B * d->N * d->N * d->De / (d->dtype == EGT_BF16 ? 2 : 1));   // in floats
  size_t o = 0;
  S.h_act = o; o += S.h_sz * (size_t)(layers > 1 ? layers - 1 : 0);
  S.e_act = o; o += S.e_sz * (size_t)(layers > 1 ? layers - 1 : 0);
  S.blk = o; o += L.saved_total * (size_t)layers;
  S.saved_total = o;
  S.ws_total = L.common_total + L.layer_total * (size_t)layers;
  return S;
}

extern "C" size_t egt_stack_saved_bytes(const egt_block_desc* d, int32_t layers) {
  if (block_check(d, false) || layers < 1) return 0;
  return stack_layout(d, layers).saved_total * sizeof(float);
}
extern "C" size_t egt_stack_workspace_bytes(const egt_block_desc* d, int32_t layers) {
  if (block_check(d, false) || layers < 1) return 0;
  return stack_layout(d, layers).ws_total * sizeof(float);
}

static void bind_layer(const egt_block_desc* d, const StackLayout& S, const BlockLayout& L, int l,
                       BlockArgs& a, float* saved, float* ws) {
  float* bs = saved + S.blk + L.saved_total * (size_t)l;
  a.v_att = bs + L.v_att; a.stats = bs + L.stats; a.qkvp = bs + L.qkvp;
  // per-layer workspace: block l's epilogue must not race block l+1's prepared weights, and the
  // deferred reductions / weight gradients need every layer's partials and dQKV rows at the end
  bind_ws(L, a, ws, ws + L.common_total + L.layer_total * (size_t)l, l & 1);
  a.pw = bs + L.pw_sv;
  (void)d;
}

extern "C" int egt_stack_fwd(const egt_block_desc* desc, int32_t layers, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, void* h_out, void* e_out, void* saved,
                             void* workspace, void* stream) {
  if (layers < 1) EGT_FAIL(EGT_E_SHAPE, "layers must be >= 1");
  if (!params || !h || !e || !h_out || !e_out || !saved || !workspace)
    EGT_FAIL(EGT_E_NULL, "params/h/e/h_out/e_out/saved/workspace is NULL");
  if (block_check(desc, true)) return block_check(desc, true);
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  if (layers > 64) EGT_FAIL(EGT_E_SHAPE, "at most 64 layers per stack call");
  const StackLayout S = stack_layout(desc, layers);
  const BlockLayout L = layout(desc);
  float* sv = (float*)saved;
  BlockArgs as[64];
  for (int l = 0; l < layers; ++l) {
    egt_block_desc dl = *desc;
    dl.seed = layer_seed(desc->seed, l);
    BlockArgs& a = as[l];
    int rc = fill_block(&dl, params + l, a);
    if (rc) return rc;
    const float* hin = l == 0 ? (const float*)h : sv + S.h_act + S.h_sz * (size_t)(l - 1);
    const float* ein = l == 0 ? (const float*)e : sv + S.e_act + S.e_sz * (size_t)(l - 1);
    bind_common(&dl, a, hin, ein, key_mask, attn_mask, nullptr, sv, (float*)workspace);
    bind_layer(&dl, S, L, l, a, sv, (float*)workspace);
    a.h_out = l == layers - 1 ? (float*)h_out : sv + S.h_act + S.h_sz * (size_t)l;
    a.e_out = l == layers - 1 ? (float*)e_out : sv + S.e_act + S.e_sz * (size_t)l;
  }
  // edge weights of every layer in one launch; each block's epilogue then finishes the node side
  // (dense_mha + residual) and already produces the next block's packed QKV, so a layer is ONE
  // launch wherever the epilogue covers the geometry
  egt_node_launch_prep(as, layers, (hipStream_t)stream);
  int prev_epi = 0;
  for (int l = 0; l < layers; ++l) {
    BlockArgs& a = as[l];
    a.prep = 0;
    a.epi = 1;
    if (l + 1 < layers) {
      const BlockArgs& nx = as[l + 1];
      a.epi = 2;
      a.nx_nm_g = nx.nm_g; a.nx_nm_b = nx.nm_b; a.nx_Wqkv = nx.Wqkv; a.nx_bqkv = nx.bqkv;
      a.nx_qkvp = nx.qkvp;
    }
    DISPATCH_BDE(desc->De, prev_epi = launch_fwd<DE>(a, (hipStream_t)stream, prev_epi == 2));
  }
  EGT_HIP_LAUNCH_CHECK("egt_stack_fwd");
  return EGT_OK;
}

extern "C" int egt_stack_bwd(const egt_block_desc* desc, int32_t layers, const egt_block_params* params,
                             const void* h, const void* e, const uint8_t* key_mask,
                             const void* attn_mask, const void* saved, const void* d_h_out,
                             const void* d_e_out, void* d_h, void* d_e,
                             const egt_block_params* grads, void* workspace, void* stream) {
  if (layers < 1) EGT_FAIL(EGT_E_SHAPE, "layers must be >= 1");
  if (layers > 64) EGT_FAIL(EGT_E_SHAPE, "at most 64 layers per stack call");
  if (!params || !grads || !h || !e || !saved || !d_h_out || !d_e_out || !d_h || !d_e || !workspace)
    EGT_FAIL(EGT_E_NULL, "params/grads/h/e/saved/d_h_out/d_e_out/d_h/d_e/workspace is NULL");
  if (block_check(desc, true)) return block_check(desc, true);
  if ((desc->flags & EGT_BF_ATTN_MASK) && !attn_mask) EGT_FAIL(EGT_E_NULL, "ATTN_MASK set but attn_mask is NULL");
  if (d_h == d_h_out) EGT_FAIL(EGT_E_FLAGS, "d_h must not alias d_h_out (d_e may alias d_e_out)");
  const StackLayout S = stack_layout(desc, layers);
  const BlockLayout L = layout(desc);
  float* sv = (float*)saved;
  const bool gated = (desc->flags & EGT_BF_GATE) != 0;
  BlockArgs as[64];
  for (int l = layers - 1; l >= 0; --l) {
    egt_block_desc dl = *desc;
    dl.seed = layer_seed(desc->seed, l);
    BlockArgs& a = as[l];
    int rc = fill_block(&dl, params + l, a);
    if (rc) return rc;
    const egt_block_params* g = grads + l;
    {
      const void* const* gp = reinterpret_cast<const void* const*>(g);
      for (int i = 0; i < 14; ++i) {
        if (!gated && (i == 2 || i == 3)) continue;
        if (!gp[i]) EGT_FAIL(EGT_E_NULL, "layer %d gradient pointer #%d is NULL", l, i);
      }
    }
    const float* hin = l == 0 ? (const float*)h : sv + S.h_act + S.h_sz * (size_t)(l - 1);
    const float* ein = l == 0 ? (const float*)e : sv + S.e_act + S.e_sz * (size_t)(l - 1);
    bind_common(&dl, a, hin, ein, key_mask, attn_mask, nullptr, sv, (float*)workspace);
    bind_layer(&dl, S, L, l, a, sv, (float*)workspace);
    // d_e flows in place below the top layer; every layer keeps its own dh (the deferred dWo
    // contraction reads dh' of each layer at the end)
    auto dhbuf = [&](int ll) { return (float*)workspace + L.common_total + L.layer_total * (size_t)ll + L.dhbuf; };
    a.dh_out = l == layers - 1 ? (const float*)d_h_out : (const float*)dhbuf(l + 1);
    a.de_out = l == layers - 1 ? (const float*)d_e_out : (const float*)d_e;
    a.dh = l == 0 ? (float*)d_h : dhbuf(l);
    a.de = (float*)d_e;
    a.g_ne_g = (float*)g->norm_edge_gamma; a.g_ne_b = (float*)g->norm_edge_beta;
    a.g_Wg = (float*)g->attention_gates_kernel; a.g_bg = (float*)g->attention_gates_bias;
    a.g_We = (float*)g->dense_edge_b_kernel; a.g_be = (float*)g->dense_edge_b_bias;
    a.g_nm_g = (float*)g->norm_mha_gamma; a.g_nm_b = (float*)g->norm_mha_beta;
    a.g_Wqkv = (float*)g->dense_qkv_kernel; a.g_bqkv = (float*)g->dense_qkv_bias;
    a.g_Wo = (float*)g->dense_mha_kernel; a.g_bo = (float*)g->dense_mha_bias;
    a.g_Wr = (float*)g->dense_edge_r_kernel; a.g_br = (float*)g->dense_edge_r_bias;
  }
  bool fuse = true;
  for (int l = 0; l < layers; ++l) fuse = fuse && node_fused_ok(as[l]);
  for (int l = layers - 1; l >= 0; --l) {
    as[l].prep = 0;   // the LN-folded edge weights were prepared by the forward and live in `saved`
    DISPATCH_BDE(desc->De, launch_bwd<DE>(as[l], L, (hipStream_t)stream, l == layers - 1, l > 0 ? &as[l - 1] : nullptr,
                                             l + 1 < layers ? &as[l + 1] : nullptr, fuse));
  }
  egt_node_launch_wgrads(as, layers, (hipStream_t)stream);
  egt_node_launch_reduce(as, layers, L.nwg_bwd, L.EP, (hipStream_t)stream);
  EGT_HIP_LAUNCH_CHECK("egt_stack_bwd");
  return EGT_OK;
}
