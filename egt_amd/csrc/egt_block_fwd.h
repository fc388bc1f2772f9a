// Forward pair kernels of the fused attention block (included by egt_block.hip, which holds the dispatch):
// k_block_fwd<De,KVL,ML,FULL,BF> and the four-rows-per-iteration k_block_fwd_r4<De,FULL,NW,BF> for narrow edge channels.
#pragma once
// Cache-policy hints of the streamed tiles (egt_tile.h).  k_block_fwd reads e_l ONCE: non-temporal, so that it does not push e_{l+1} --
// which this launch writes and the next one reads, 134 MB of the 256 MB memory-side cache at the headline batch -- out.  Measured
// (tools/ab.sh, same box): k_block_fwd 60.7 -> 59.0 us; the stores non-temporal instead (nothing retained): 67 us.
#ifndef EGT_NT_FWD_E
#define EGT_NT_FWD_E true
#endif
#ifndef EGT_NT_FWD_ST
#define EGT_NT_FWD_ST false
#endif

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
  const int RG = a.RGF;   // query rows per workgroup (<= 16; launch_fwd: fewer when the launch would leave workgroup slots empty)
  const int lgroups = (N + RG - 1) / RG;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = wg / lgroups, lg = wg % lgroups;
  const int l0 = lg * RG, nlr = min(RG, N - l0);   // the group's rows [l0, l0 + nlr)
  float* tl0 = sm + wave * 2 * G::TILE_FLOATS;  // two tiles per wave (ping-pong)
  float* kvs = sm + 8 * G::TILE_FLOATS;         // [N][KV_LD]   (KVL)
  float* qs = kvs + (KVL ? N * KV_LD : 0);     // [16][QS_LD]  (KVL)
  float* kms = qs + (KVL ? 16 * QS_LD : 0);    // [N]          (KVL)
  const bool gated = (a.flags & EGT_BF_GATE) != 0;
  const bool clip = (a.flags & EGT_BF_CLIP) != 0;

  // Start-up loads in as few memory round trips as possible (round 5: hipcc compiled the staging loops as load -> wait -> LDS store
  // per iteration, eight dependent round trips per thread at N = 64, and the 44 weight loads of a lane as one more behind them):
  // the K / V rows go in batches of eight 16-byte loads per thread, the Q rows and the key-mask adds ride in the first batch, and
  // the lane-constant MFMA operands below are requested BEFORE the first batch is waited for.
  float4 kvb[8], qb = make_float4(0.f, 0.f, 0.f, 0.f);
  float kmb = 0.f;
  const float* kvsrc = a.qkvp + (size_t)b * N * QKVP;
  if (KVL) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = threadIdx.x + 256 * u, row = min(i >> 5, N - 1), f = i & 31;
      kvb[u] = *reinterpret_cast<const float4*>(kvsrc + (size_t)row * QKVP + 64 + f * 4);
    }
    {
      const int row = threadIdx.x >> 4, f = threadIdx.x & 15, l = min(l0 + row, N - 1);
      qb = *reinterpret_cast<const float4*>(kvsrc + (size_t)l * QKVP + f * 4);
    }
    if ((int)threadIdx.x < N) kmb = (a.km && a.km[(size_t)b * N + threadIdx.x] == 0) ? -EGT_NEG : 0.0f;
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
  if (KVL) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = threadIdx.x + 256 * u, row = i >> 5, f = i & 31;
      if (i < N * 32) *reinterpret_cast<float4*>(kvs + row * KV_LD + f * 4) = kvb[u];
    }
    *reinterpret_cast<float4*>(qs + (threadIdx.x >> 4) * QS_LD + (threadIdx.x & 15) * 4) = qb;
    if ((int)threadIdx.x < N) kms[threadIdx.x] = kmb;
    for (int i = 2048 + threadIdx.x; i < N * 32; i += 256) {   // N > 64: the keys past the first batch
      const int row = i >> 5, f = i & 31;
      *reinterpret_cast<float4*>(kvs + row * KV_LD + f * 4) = *reinterpret_cast<const float4*>(kvsrc + (size_t)row * QKVP + 64 + f * 4);
    }
    for (int i = 256 + threadIdx.x; i < N; i += 256) kms[i] = (a.km && a.km[(size_t)b * N + i] == 0) ? -EGT_NEG : 0.0f;
    __syncthreads();
  }

  const int ntile = (N + 15) / 16;
  int nrows = 0;
  for (int li = 0; li < 4; ++li) nrows += (wave + 4 * li < nlr) ? 1 : 0;
  const int total = nrows * ntile;

  // e tiles in flight per wave.  A De <= 16 tile is 0.5 - 1 KB, so narrow tiles travel PFD
  // iterations ahead in a ring of register sets (more bytes in flight per CU).  The ring is indexed
  // statically (loop unrolled by PFD); the iterations that pad the last group re-run the last tile
  // with every global write predicated off, which keeps the body straight-line (exact waits).
  constexpr int PFD = (DE <= 16 && KVL) ? 4 : 1;   // measured: +3.5 % at De = 8, N = 120; without K/V in LDS the extra registers spill
  TileRegs<DE> ring[PFD];
  auto prefetch = [&](TileRegs<DE>& tr, int it) {
    if (PFD > 1) it = min(it, total - 1);
    const int l = l0 + wave + 4 * (it / ntile), m0 = (it % ntile) * 16;
    const size_t pair0 = ((size_t)b * N + l) * N + m0;
    tile_gload<DE, EGT_NT_FWD_E>(tr, e_in + pair0 * DE, lane, FULL ? 16 : min(16, N - m0));
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
    const int l = l0 + wave + 4 * li, m0 = mt * 16, m = m0 + p;
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
      const int itp = it - 1, lp = l0 + wave + 4 * (itp / ntile), m0p = (itp % ntile) * 16;
      tile_from_lds<DE, EGT_NT_FWD_ST>(tl0 + (itp & 1) * G::TILE_FLOATS, e_o + (((size_t)b * N + lp) * N + m0p) * DE,
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
      tile_from_lds<DE, EGT_NT_FWD_ST>(tl, e_o + pair0 * DE, lane, rows_valid);
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
  if (KVL && a.epi) fwd_node_epilogue(a, sm, qs, b, l0, nlr, N, wave, p, q);
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
    fwd_stage_kv<NT>(kvs, src, N);
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
  if (a.epi) fwd_node_epilogue(a, sm + hf * 16 * QS_LD, qs + hf * 16 * QS_LD, b, lg * RW + hf * 16, min(16, N - (lg * RW + hf * 16)), N, wv, p, q);
}

