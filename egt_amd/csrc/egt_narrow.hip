// Narrow edge channels (De = 8, H = 8, d <= 8: BASELINE configs 3 and 4) -- their own pair kernels.
//
// A De = 8 pair row is 32 bytes: the 16-pair MFMA tile of the wide kernels is 512 B, its per-tile
// skeleton and dependent phases (LDS round trip -> LayerNorm over lanes -> MFMA chain -> ...) cost more
// than the arithmetic, and at ~240 VGPRs only two waves per SIMD hide the latency of tiles that small
// (measured: 1.1 TB/s, 0.08 of the roof).  These kernels keep a pair's whole working set in FOUR lanes:
//   lane = p + 16 q:  lane (p, q) owns heads 2q, 2q+1 and edge channels 2q, 2q+1 of ONE pair per step (p = query row in the
//   forward, key in the backward); e arrives by plain 8-byte global loads straight into the lane (no LDS tile, no transposition).
// That is the operand / result layout of v_mfma_f32_16x16x4_f32 (B operand: lane (n = p, k = q); result: lane (n = p) holds
// rows 4q .. 4q+3), so every pair-local channel contraction -- the [gates | E] projections, dense_edge_r, dH_ext, d ehat -- is
// two to four MFMAs with the lane's own registers as B operand and the weights as 2-4 A-operand registers: no exchange, no weight
// table.  LayerNorm sums cross the four 16-lane rows (v_permlane16_swap / v_permlane32_swap).
//   k_narrow_fwd: the lane keeps a whole query row's softmax state (online softmax over the key loop: no cross-lane reduction);
//     the four waves of a workgroup split the key range and merge their (max, sum, A.V) triples once at the end.
//     <= 128 VGPRs, 20 KB of LDS per workgroup: four workgroups per CU.
//   k_narrow_bwd: see the kernel's header.  168 VGPRs, 47 KB of LDS: three workgroups per CU.
// (Round 2 ran both on quad lanes, lane = 4 p + q, with 2-channel partial sums combined by DPP quad_perm adds: 48 weight
// registers, ~70 VALU instructions per step more, the backward at two workgroups per CU -- forward 91 -> 86 us, backward
// 314 -> 208 us at config 3.)
// Same BlockArgs, saved tensors, partial-buffer layouts, mask order / RNG stream and node-side epilogue /
// prologue as the wide kernels: the dispatch in launch_fwd / launch_bwd is the only difference.
// Own translation unit: built with -fno-slp-vectorize (build.py) -- hipcc otherwise pairs scalar adds into v_pk_add_f32,
// which cannot take a DPP operand (the dQ reduction).
#include "egt_common.h"

#include "egt_block.h"
#include "egt_block_dev.h"

#define NRW_DE 8
#ifdef NRW_TIMING   // measurement builds (EGT_NARROW_FLAGS=-DNRW_TIMING): per-wave cycle sums of the kernel's sections, printed at exit
#define NSTMP(i) do { const unsigned tn__ = (unsigned)__builtin_amdgcn_s_memtime(); nacc[i] += tn__ - nlast; nlast = tn__; } while (0)
#else
#define NSTMP(i) do {} while (0)
#endif
#ifdef NRW_TIMING
struct NrwTimer {   // host side of a timing build: one device record of 8 counters per wave, summed after every launch (synchronises)
  const char* kernel; const char* const* names; int nsec;
  unsigned* dev = nullptr; unsigned* host = nullptr; int n = 0; double sum[8] = {}; long launches = 0, waves = 0;
  unsigned* attach(int nwg) {
    if (n < nwg) { if (dev) (void)hipFree(dev); (void)hipMalloc(&dev, (size_t)nwg * 32 * sizeof(unsigned)); free(host); host = (unsigned*)malloc((size_t)nwg * 32 * sizeof(unsigned)); n = nwg; }
    return dev;
  }
  void collect(int nwg, hipStream_t st) {
    (void)hipStreamSynchronize(st);
    if (++launches <= 20) return;
    (void)hipMemcpy(host, dev, (size_t)nwg * 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int w = 0; w < nwg * 4; ++w) { for (int i = 0; i < 8; ++i) sum[i] += host[w * 8 + i]; ++waves; }
  }
  void report() const {
    if (!waves) return;
    double tot = 0; for (int i = 0; i < nsec; ++i) tot += sum[i];
    fprintf(stderr, "[egt] %s section cycles per wave (mean over %ld waves, %ld launches; total %.0f):\n", kernel, waves, launches, tot / waves);
    for (int i = 0; i < nsec; ++i) fprintf(stderr, "    %-30s %10.0f  (%.1f %%)\n", names[i], sum[i] / waves, 100.0 * sum[i] / tot);
  }
};
#endif
#ifndef NRW_ABL          // timing ablations (measurement builds only, tools/build_variant.sh -DNRW_ABL=<bits>):
#define NRW_ABL 0        // 1 no e' stores, 2 no e loads, 4 no K/V chunk traffic, 8 no mask hash, 16 no exp/sigmoid, 32 no epilogue
#endif
#ifndef NRW_KB
#define NRW_KB 4
#endif
//      NRW_KB: keys per K/V chunk and per e prefetch block (one 128-byte line of a pair row)
#define NRW_KV_CHUNK (NRW_KB * 128)   // floats: [key][K 64 | V 64]


// plain v_max_f32 / v_min_f32 (fmaxf's llvm.maxnum canonicalises both operands first: three instructions per max)
__device__ __forceinline__ float nrw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nrw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Matrix-core lanes (lane = p + 16 q): the four lanes of a pair are p, p+16, p+32, p+48 -- sums over a pair's channels / heads
// cross the four 16-lane rows (v_permlane16_swap, v_permlane32_swap: 3 instructions per level)
__device__ __forceinline__ float nrw_sum4rows(float v) { return sum_xor32(sum_xor16(v)); }
// both sums in every lane: 7 instructions for the pair (reduce-scatter over lane bit 5, reduce over bit 4, all-gather)
__device__ __forceinline__ void nrw_sum4rows_pair(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);   // [x.lo | y.lo], [x.hi | y.hi]
  float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);                                             // lanes < 32: x over bit 5; >= 32: y
  s = sum_xor16(s);
  auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  x = __uint_as_float(g[0]); y = __uint_as_float(g[1]);
}

// Wave-uniform base pointer + 32-bit lane offset (in elements): hipcc then uses the SGPR-base addressing mode and
// keeps ONE offset register per lane instead of a 64-bit address per tensor.
template <bool BF> struct NrwLd;
template <> struct NrwLd<false> {
  typedef float2 raw;
  static __device__ __forceinline__ raw load(const void* base, size_t pair, int q) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(base) + pair * NRW_DE + 2 * q);
  }
  static __device__ __forceinline__ float2 cvt(raw r) { return r; }
  static __device__ __forceinline__ void store(void* base, size_t pair, int q, float2 v) {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + pair * NRW_DE + 2 * q) = v;
  }
  static __device__ __forceinline__ raw uload(const void* base, size_t upair, int off) {   // upair: wave-uniform pair index
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(base) + upair * NRW_DE + off);
  }
  static __device__ __forceinline__ void ustore(void* base, size_t upair, int off, float2 v) {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + upair * NRW_DE + off) = v;
  }
};
template <> struct NrwLd<true> {
  typedef uint32_t raw;
  static __device__ __forceinline__ raw load(const void* base, size_t pair, int q) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(base) + pair * NRW_DE + 2 * q);
  }
  static __device__ __forceinline__ float2 cvt(raw r) { return make_float2(__uint_as_float(r << 16), __uint_as_float(r & 0xFFFF0000u)); }
  static __device__ __forceinline__ void store(void* base, size_t pair, int q, float2 v) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(base) + pair * NRW_DE + 2 * q) = f2_to_bf2(v.x, v.y);
  }
  static __device__ __forceinline__ raw uload(const void* base, size_t upair, int off) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(base) + upair * NRW_DE + off);
  }
  static __device__ __forceinline__ void ustore(void* base, size_t upair, int off, float2 v) {
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(base) + upair * NRW_DE + off) = f2_to_bf2(v.x, v.y);
  }
};

// ---- pair-local channel contractions on the matrix core ------------------------------------------------------------------
// D[m][n] = C[m][n] + sum_k A[m][k] B[k][n], K <= 16, lane (p, q): A row m = p, B column n = p, contraction slots (q, i):
//   fp32 edge tensors: K / 4 steps of v_mfma_f32_16x16x4_f32 (slot i = step i);
//   bf16 edge tensors (the dtype BASELINE.json asks for in its config 3 -- NOT a reference semantic: the reference is fp32
//   throughout; the bf16-operand products below are held to SURVEY 8(c)'s bf16 tolerance, margins on record in
//   tests/test_block_gpu.py::test_config3_as_specified_bf16_depth4_margins): ONE
//   v_mfma_f32_16x16x16_bf16 with the operands rounded to bfloat16 (fp32 accumulation) -- 16 issue cycles instead of K / 4 x 32.
typedef short nrw_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nrw_v4s nrw_b4(uint32_t lo, uint32_t hi) { union { nrw_v4s v; uint32_t u[2]; } x; x.u[0] = lo; x.u[1] = hi; return x.v; }
// Which contractions take bf16 MFMAs when the edge tensors are bf16:
//   gradient path (dH_ext, d ehat, the weight-gradient products): yes -- rounding there is unbiased noise on sums;
//   value path (the [gates | E] projections and dense_edge_r, forward AND the backward's recompute): NO by default.  Their
//   rounding (2^-9 of a logit) compounds through the per-layer LayerNorm over 8 channels: at three layers 2 of 110 k
//   outputs left SURVEY 8(c)'s bf16 tolerance by 1.8 x (tools/sweep_de8.py: ungated, N = 39, B = 9, Ly = 3); -DNRW_BF16_VALUES
//   turns them on (config 3: another 6 us per forward, 4 us per backward launch).
#ifdef NRW_F32_MMA_ONLY   // A/B and parity triage: fp32 MFMAs for bf16 edge tensors too
#define NRW_MMA_BF(BF) false
#else
#define NRW_MMA_BF(BF) (BF)
#endif
#if defined(NRW_BF16_VALUES) && !defined(NRW_F32_MMA_ONLY)
#define NRW_MMA_BF_VALUES(BF) (BF)
#else
#define NRW_MMA_BF_VALUES(BF) false
#endif
template <bool BF, int NS> struct NrwOp;                       // NS contraction slots per lane (2 or 4)
template <int NS> struct NrwOp<false, NS> { float v[NS]; };
template <int NS> struct NrwOp<true, NS> { nrw_v4s v; };
template <bool BF> __device__ __forceinline__ NrwOp<BF, 2> nrw_op(float a, float b) {
  NrwOp<BF, 2> o;
  if constexpr (BF) o.v = nrw_b4(f2_to_bf2(a, b), 0u); else { o.v[0] = a; o.v[1] = b; }
  return o;
}
template <bool BF> __device__ __forceinline__ NrwOp<BF, 4> nrw_op(float a, float b, float c, float d) {
  NrwOp<BF, 4> o;
  if constexpr (BF) o.v = nrw_b4(f2_to_bf2(a, b), f2_to_bf2(c, d)); else { o.v[0] = a; o.v[1] = b; o.v[2] = c; o.v[3] = d; }
  return o;
}
__device__ __forceinline__ NrwOp<true, 2> nrw_op_raw(uint32_t bf2) { NrwOp<true, 2> o; o.v = nrw_b4(bf2, 0u); return o; }   // two bf16 values as they sit in memory
template <bool BF, int NS> __device__ __forceinline__ v4f nrw_mm(const NrwOp<BF, NS>& A, const NrwOp<BF, NS>& B, v4f c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A.v, B.v, c, 0, 0, 0);
  else {
#pragma unroll
    for (int i = 0; i < NS; ++i) c = MFMA(A.v[i], B.v[i], c);
    return c;
  }
}

// feature sets with compile-time flags (the run-time flag tests cost scalar branches / selects in every step)
#define NRW_F_GATED 1
#define NRW_F_CLIP 2
#define NRW_F_RUNTIME (-1)

// ------------------------------------------------------------------------------------ forward ---
// Workgroup = (graph b, 16 query rows), NW waves = NW equal parts of the key range (NW = 4; 8 for launches of at most one
// workgroup per CU -- BASELINE config 4 as specified, B = 16: 128 workgroups -- where the extra waves are free).  A step = one
// block of NRW_KB keys: logits of the four keys first, then ONE softmax rescale for the block (flash-style blocking).
// LDS: [NRW_FWD_AREA(NW)] = [NW waves][NRW_KV_CHUNK] K/V chunks + [NW][NRW_KB] key-mask adds during the key loop, the
//      merge buffer [NW - 1][20][64] after it, then the epilogue's staging rows; [16][QS_LD] V_att rows for the epilogue.
#define NRW_FWD_AREA(NW) ((((NW) - 1) * 20 * 64) > ((NW) * NRW_KV_CHUNK + (NW) * NRW_KB) ? (((NW) - 1) * 20 * 64) : ((NW) * NRW_KV_CHUNK + (NW) * NRW_KB))
// Projections and dense_edge_r of a pair: two MFMAs each (4 A-operand registers).
#ifndef NRW_FWDM_OCC
#define NRW_FWDM_OCC 3
#endif
// HR ("half rows", with NW = 8): workgroup = (graph, EIGHT query rows) for launches whose 16-row grid would leave half of the CUs
// without a workgroup (config 4 as specified: B = 16, N = 120 -> 128 workgroups of 16 rows on 256 CUs).  Lanes p < 8 and p >= 8 work on the
// same row p & 7 and on the even / odd key of a step's key pair (a row's 64-byte segment per request instead of 32); their online-softmax
// states are merged across the lane pair (p, p ^ 8) before the waves' key ranges are.
template <bool BF, int FEAT, int NW, bool HR>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? NRW_FWDM_OCC : 1) k_narrow_fwd(BlockArgs a) {
  static_assert(!HR || (NW == 8 && NRW_KB % 2 == 0), "half-row workgroups: eight waves, key blocks of an even size");
  constexpr int RG = HR ? 8 : 16;            // query rows per workgroup
  constexpr int KS = HR ? NRW_KB / 2 : NRW_KB;   // key steps per block
  seed_from_device(a);
#ifdef NRW_TIMING
  unsigned nacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  typedef NrwLd<BF> LD;
  constexpr bool MV = NRW_MMA_BF_VALUES(BF);   // bf16 MFMAs for the value path (projections, dense_edge_r): off by default, see NRW_MMA_BF_VALUES
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;   // matrix-core lanes
  const int N = a.N;
  const int lgroups = (N + RG - 1) / RG;
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int b, lg;
  egt_group_order(wg, a.B, lgroups, N, b, lg, RG);
  const int hp = HR ? (p >> 3) : 0;          // HR: the lane's key of a key pair
  float* kvw = sm + wave * NRW_KV_CHUNK;
  float* kmw = sm + NW * NRW_KV_CHUNK + wave * NRW_KB;
  float* qs = sm + NRW_FWD_AREA(NW);
  const bool gated = FEAT >= 0 ? (FEAT & NRW_F_GATED) != 0 : (a.flags & EGT_BF_GATE) != 0;
  const bool clip = FEAT >= 0 ? (FEAT & NRW_F_CLIP) != 0 : (a.flags & EGT_BF_CLIP) != 0;
  const bool ln_on = (a.flags & EGT_BF_NO_EDGE_LN) == 0;
  const int l = lg * RG + (HR ? (p & 7) : p);
  const bool row_ok = l < N;
  const bool row_own = row_ok && (!HR || p < 8);   // the lane that writes the row's V_att / statistics
  const size_t rowl = (size_t)b * N + min(l, N - 1);

  // ---- lane constants: Q of the row (heads 2q, 2q+1); A operands of the two channel contractions ----
  float Qf[16];
  {
    const float4* qp = reinterpret_cast<const float4*>(a.qkvp + rowl * QKVP + q * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float4 v = qp[u]; Qf[4*u] = v.x; Qf[4*u+1] = v.y; Qf[4*u+2] = v.z; Qf[4*u+3] = v.w; }
  }
  // [gates | E] = Wp'^T.xhat + c: A[m = column p][k = q] at step s = Wp'[channel 2q + s][column p]; result rows 4q .. 4q+3 = the
  // lane's four columns.  dense_edge_r: A[m][k = q] at step j = Wr[head 2q + j][channel of row m]; rows 4q', 4q'+1 of the result
  // carry channels 2q', 2q'+1 (rows 4q'+2, 4q'+3 unused: zero weights).
  const int jm = p & 3, cm = 2 * (p >> 2) + jm;
  float brr[2];
  v4f c2r;
  const NrwOp<MV, 2> pwA = nrw_op<MV>(a.pw[(2 * q) * 16 + p], a.pw[(2 * q + 1) * 16 + p]);
  const NrwOp<MV, 2> wrA = nrw_op<MV>(jm < 2 ? a.Wr[(2 * q) * NRW_DE + cm] : 0.0f, jm < 2 ? a.Wr[(2 * q + 1) * NRW_DE + cm] : 0.0f);
#pragma unroll
  for (int r = 0; r < 4; ++r) c2r[r] = a.pw[16 * 16 + 4 * q + r];
  brr[0] = a.br[2 * q]; brr[1] = a.br[2 * q + 1];

  // ---- the wave's key blocks ----
  const int nblk = (N + NRW_KB - 1) / NRW_KB;
  const int blk0 = (wave * nblk) / NW, blk1 = ((wave + 1) * nblk) / NW;
  float mx[2] = {-3.0e38f, -3.0e38f}, sum[2] = {0.f, 0.f}, O[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) O[k] = 0.f;

  const size_t pair_row = rowl * N;              // (mask-RNG counter)
  const size_t ugraph = (size_t)b * N * N;       // wave-uniform: first pair of the graph
  const int loff = min(l, N - 1) * N * NRW_DE + 2 * q;   // the lane's element offset from the pair (row 0, key m)
  typename LD::raw eb[NRW_KB];
  constexpr int NU = NRW_KB / 2;   // float4s per lane of a K/V chunk
  v4f kvr[NU];   // (a plain vector type: hipcc leaves an array of HIP's float4 struct, captured by the lambdas, in scratch memory)
  float kmr = 0.f;
  auto fetch_e = [&](int blk) __attribute__((always_inline)) {
    const int m0 = blk * NRW_KB;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (HR) {   // keys m0 + 2 kk + hp: wave-uniform pair index of the even key, the odd key through the lane offset (never past the row's last key)
        const int k0 = m0 + 2 * kk;
        eb[kk] = LD::uload(a.e, ugraph + min(k0, N - 1), loff + ((hp != 0 && k0 + 1 < N) ? NRW_DE : 0));
      } else eb[kk] = LD::uload(a.e, (NRW_ABL & 2) ? (size_t)kk : ugraph + min(m0 + kk, N - 1), loff);
    }
  };
  auto fetch_kv = [&](int blk) __attribute__((always_inline)) {   // K / V of the block come from L2: requested half a block ahead
    const int m0 = blk * NRW_KB;
    const int kmax = N - 1 - m0;   // keys of the block past the graph's last one are clamped to it
    const float* kvb = a.qkvp + ((size_t)b * N + m0) * QKVP;   // wave-uniform
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int f = lane + 64 * u, key = min(f >> 5, kmax), piece = f & 31;
      if (!(NRW_ABL & 4) || blk == blk0) kvr[u] = *reinterpret_cast<const v4f*>(kvb + key * QKVP + 64 + piece * 4);
    }
    if (a.km) kmr = (a.km[(size_t)b * N + m0 + min(lane & (NRW_KB - 1), kmax)] == 0) ? -EGT_NEG : 0.0f;
  };
  if (blk0 < blk1) { fetch_e(blk0); fetch_kv(blk0); }
  NSTMP(0);   // lane constants, weight table, first requests

  // one block of keys; nv = number of real keys in it (NRW_KB except in the graph's last block)
  auto block = [&](int blk, int nv) __attribute__((always_inline)) {
    const int m0 = blk * NRW_KB;
    // commit the chunk to the wave's LDS slot (DS operations of a wave retire in order: the previous
    // chunk's reads are behind these writes), take the e registers, then put the next block in flight
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int f = lane + 64 * u;
      *reinterpret_cast<v4f*>(kvw + (f >> 5) * 128 + (f & 31) * 4) = kvr[u];
    }
    if (lane < NRW_KB) kmw[lane] = kmr;
    float2 ev[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ev[kk] = LD::cvt(eb[kk]);
    if (blk + 1 < blk1) fetch_e(blk + 1);
    lds_sync();
    float xl[KS][2], gl[KS][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kl = HR ? 2 * kk + hp : kk;   // the lane's key inside the block
      const int m = m0 + kl;
      // ---- norm_edge: the pair's 8 channels sit in the quad, two per lane (two-pass moments) ----
      float x0 = ev[kk].x, x1 = ev[kk].y;
      const float mu = ln_on ? nrw_sum4rows(x0 + x1) * 0.125f : 0.0f;
      x0 -= mu; x1 -= mu;
      const float var = nrw_sum4rows(fmaf(x0, x0, x1 * x1)) * 0.125f;
      const float rstd = ln_on ? __builtin_amdgcn_rsqf(var + a.ln_eps) : 1.0f;
      x0 *= rstd; x1 *= rstd;
      // ---- [attention_gates | dense_edge_b]: acc[r] = column 4q + r of the pair ----
      const v4f acc = nrw_mm<MV, 2>(pwA, nrw_op<MV>(x0, x1), c2r);
      // ---- scaled QK^T, clip, + E (egt_layers.py:79-86) ----
      float Kf[16];
      {
        const float4* kp = reinterpret_cast<const float4*>(kvw + kl * 128 + q * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float4 v = kp[u]; Kf[4*u] = v.x; Kf[4*u+1] = v.y; Kf[4*u+2] = v.z; Kf[4*u+3] = v.w; }
      }
      float hh[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float dot = Qf[j] * Kf[j];
#pragma unroll
        for (int k = 1; k < 8; ++k) dot = fmaf(Qf[2 * k + j], Kf[2 * k + j], dot);
        float ah = dot * a.scale;
        if (clip) ah = nrw_min(nrw_max(ah, a.clip_lo), a.clip_hi);
        hh[j] = ah + acc[2 * j + 1];
        xl[kk][j] = hh[j];
        gl[kk][j] = acc[2 * j];
      }
      const MaskRegs mr{make_float2(1.f, 1.f), 0};
      if (!(NRW_ABL & 8)) apply_masks<false>(a, kmw[kl], mr, (size_t)(((uint32_t)pair_row + (uint32_t)m) * (uint32_t)BH), q, xl[kk], gl[kk]);   // (counter mod 2^32)
      if (kl >= nv) { xl[kk][0] = -3.0e38f; xl[kk][1] = -3.0e38f; }   // past the graph's last key: probability exactly 0
      // ---- dense_edge_r + res_edge: e' = e + H_hat.Wr + br, the lane's two channels ----
      float2 eo;
      {
        v4f t = {ev[kk].x + brr[0], ev[kk].y + brr[1], 0.f, 0.f};
        t = nrw_mm<MV, 2>(wrA, nrw_op<MV>(hh[0], hh[1]), t);
        eo.x = t[0];
        eo.y = t[1];
      }
      if (row_ok && kl < nv && !(NRW_ABL & 1)) LD::ustore(a.e_out, ugraph + (m - hp), loff + hp * NRW_DE, eo);   // (m - hp: wave-uniform)
      if ((NRW_ABL & 1) && eo.x == 123.456f) LD::ustore(a.e_out, ugraph + (m - hp), loff + hp * NRW_DE, eo);
    }
    if (blk + 1 < blk1) fetch_kv(blk + 1);   // (after the logits phase: its registers are free again)
    // ---- one online-softmax step for the block, x gate, A.V: the row's state never leaves the lane ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float mn = mx[j];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) mn = nrw_max(mn, xl[kk][j]);
      const float alpha = (NRW_ABL & 16) ? (mx[j] - mn) : __expf(mx[j] - mn);
      mx[j] = mn;
      sum[j] *= alpha;
#pragma unroll
      for (int k = 0; k < 8; ++k) O[2 * k + j] *= alpha;
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int kl = HR ? 2 * kk + hp : kk;
      float Vf[16];
      {
        const float4* vp = reinterpret_cast<const float4*>(kvw + kl * 128 + 64 + q * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float4 v = vp[u]; Vf[4*u] = v.x; Vf[4*u+1] = v.y; Vf[4*u+2] = v.z; Vf[4*u+3] = v.w; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float pe = (NRW_ABL & 16) ? (xl[kk][j] - mx[j]) : __expf(xl[kk][j] - mx[j]);
        sum[j] += pe;
        const float av = gated ? pe * ((NRW_ABL & 16) ? gl[kk][j] : egt_sigmoid(gl[kk][j])) : pe;
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(av, Vf[2 * k + j], O[2 * k + j]);
      }
    }
  };
  {
    const int nfull = N / NRW_KB;                 // blocks below nfull hold NRW_KB real keys
    const int bfull = min(blk1, nfull);
    for (int blk = blk0; blk < bfull; ++blk) block(blk, NRW_KB);
    if (blk1 > nfull && blk0 <= nfull) block(nfull, N - nfull * NRW_KB);   // the graph's ragged last block
  }
  if (HR) {   // the two key subsets of a row: lanes (p, p ^ 8) -- both lanes end up with the merged state, lane p < 8 is the one used
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float mo = __shfl_xor(mx[j], 8), so = __shfl_xor(sum[j], 8);
      const float mn = fmaxf(mx[j], mo);
      const float f0 = __expf(mx[j] - mn), f1 = __expf(mo - mn);
      mx[j] = mn;
      sum[j] = fmaf(sum[j], f0, so * f1);
#pragma unroll
      for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(O[2 * k + j], f0, __shfl_xor(O[2 * k + j], 8) * f1);
    }
  }
  NSTMP(1);   // key loop
  // the node-side epilogue's weight fragments / bias / residual rows: requested now, landed by the time the key ranges are merged
  // (eight waves: two waves per SIMD at most, the 69 registers are there; with four waves the kernel runs three workgroups per CU at <= 128)
  const bool epi_on = a.epi && !(NRW_ABL & 32);
  FwdEpiRegs epiR;
  if (NW == 8 && epi_on && wave < 4) fwd_node_epilogue_load(a, epiR, b, lg * RG, N, wave, lane & 15, lane >> 4);
  // ---- merge the key ranges (waves 1 .. NW-1 -> LDS -> wave 0), write V_att / statistics ----
  __syncthreads();   // every wave is done with its K/V chunk: the area becomes the merge buffer
  float* mg = sm;    // [NW - 1][20][64]
  if (wave > 0) {
    float* o = mg + (wave - 1) * 20 * 64 + lane;
    o[0] = mx[0]; o[64] = mx[1]; o[128] = sum[0]; o[192] = sum[1];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[(4 + k) * 64] = O[k];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
      const float* o = mg + w * 20 * 64 + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mo = o[j * 64], so = o[(2 + j) * 64];
        const float mn = fmaxf(mx[j], mo);
        const float f0 = __expf(mx[j] - mn), f1 = __expf(mo - mn);
        mx[j] = mn;
        sum[j] = fmaf(sum[j], f0, so * f1);
#pragma unroll
        for (int k = 0; k < 8; ++k) O[2 * k + j] = fmaf(O[2 * k + j], f0, o[(4 + 2 * k + j) * 64] * f1);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 vo = make_float2(O[2 * k] / sum[0], O[2 * k + 1] / sum[1]);
      if (row_own && k < a.DK) *reinterpret_cast<float2*>(a.v_att + rowl * a.Dh + k * BH + 2 * q) = vo;
      if (a.epi && (!HR || p < 8)) *reinterpret_cast<float2*>(qs + p * QS_LD + k * BH + 2 * q) = vo;
    }
    if (row_own) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* st = a.stats + (rowl * BH + 2 * q + j) * 4;
        st[0] = mx[j];
        st[1] = sum[j];
      }
    }
  }
  NSTMP(2);   // sync + merge of the key quarters
  // node-side epilogue on the 16 rows (its own lane roles: MFMA layout); its staging rows reuse the merge area
  if (epi_on) {
    if (NW == 4) fwd_node_epilogue(a, sm, qs, b, lg * RG, min(RG, N - lg * RG), N, wave, lane & 15, lane >> 4);
    else if (wave < 4) fwd_node_epilogue_finish(a, epiR, sm, qs, b, lg * RG, min(RG, N - lg * RG), N, wave, lane & 15, lane >> 4);
    else fwd_node_epilogue_idle(a);   // the epilogue is four waves' work: the others only meet its barriers
  }
#ifdef NRW_TIMING
  NSTMP(3);   // node-side epilogue
  if (a.dbg && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (wave < 4) a.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = nacc[i];
  }
#endif
}
static_assert(16 * QS_LD <= NRW_FWD_AREA(4), "the epilogue's staging rows fit the merge area");

// a.epi must already hold the epilogue the geometry allows (launch_fwd decides)
#ifdef NRW_TIMING
static const char* const g_nf_names[] = {"constants + first requests", "key loop", "sync + merge", "node-side epilogue"};
static NrwTimer g_nf{"k_narrow_fwd", g_nf_names, 4};
#endif
void egt_narrow_launch_fwd(BlockArgs& a, hipStream_t st) {
  const int grid16 = a.B * ((a.N + 15) / 16);
  static const int nw_forced = getenv("EGT_NRW_FWD_WAVES") ? atoi(getenv("EGT_NRW_FWD_WAVES")) : 0;   // 4 | 8 (tests, A/B)
  static const int hr_forced = getenv("EGT_NRW_FWD_HALF") ? atoi(getenv("EGT_NRW_FWD_HALF")) : -1;    // 0 | 1 (tests, A/B)
  const bool w8 = nw_forced == 8 || (nw_forced != 4 && grid16 <= egt_device_cus() && a.N >= 64);   // one workgroup per CU at most: eight key ranges
  const bool hr = w8 && (hr_forced == 1 || (hr_forced != 0 && 2 * grid16 <= egt_device_cus()));       // ... per two CUs: eight-row workgroups
  const dim3 grid(hr ? a.B * ((a.N + 7) / 8) : grid16);
  const dim3 block(w8 ? 512 : 256);
#ifdef NRW_TIMING
  { static bool reg = false; if (!reg) { reg = true; atexit([] { g_nf.report(); }); } }
  a.dbg = g_nf.attach(grid.x);
#endif
  const size_t lds = ((size_t)(w8 ? NRW_FWD_AREA(8) : NRW_FWD_AREA(4)) + 16 * QS_LD + 80) * 4;
  const int full = NRW_F_GATED | NRW_F_CLIP;
  const int feat = ((a.flags & EGT_BF_GATE) ? NRW_F_GATED : 0) | ((a.flags & EGT_BF_CLIP) ? NRW_F_CLIP : 0);
#define NRW_FWD(BF_, FEAT_) do { \
    if (hr) { EGT_MAX_LDS_ONCE(k_narrow_fwd<BF_, FEAT_, 8, true>); EGT_LAUNCH("k_block_fwd", (k_narrow_fwd<BF_, FEAT_, 8, true>), grid, block, lds, st, a); } \
    else if (w8) { EGT_MAX_LDS_ONCE(k_narrow_fwd<BF_, FEAT_, 8, false>); EGT_LAUNCH("k_block_fwd", (k_narrow_fwd<BF_, FEAT_, 8, false>), grid, block, lds, st, a); } \
    else { EGT_LAUNCH("k_block_fwd", (k_narrow_fwd<BF_, FEAT_, 4, false>), grid, block, lds, st, a); } } while (0)
  if (a.bf16) { if (feat == full) NRW_FWD(true, NRW_F_GATED | NRW_F_CLIP); else NRW_FWD(true, NRW_F_RUNTIME); }
  else { if (feat == full) NRW_FWD(false, NRW_F_GATED | NRW_F_CLIP); else NRW_FWD(false, NRW_F_RUNTIME); }
#undef NRW_FWD
#ifdef NRW_TIMING
  g_nf.collect(grid.x, st);
#endif
}

// ------------------------------------------------------------------- backward, matrix-core lanes ---
// k_narrow_bwd: workgroup = (graph b, 16 query rows), 4 waves; a wave owns key tiles (w, w+4, ... or a balanced contiguous range
// of (tile, row) steps) and walks the rows.  Lane roles:  lane = p + 16 q,  p = key of the tile, q = head pair AND channel pair -- the
// operand / result layout of v_mfma_f32_16x16x4_f32 (B operand: lane (n = p, k = q); result: lane (n = p) holds rows 4q..4q+3).
// The pair-local channel contractions move to the matrix core with the lane's own registers as B operand and NO exchange:
//   [gates | E] = Wp'^T.xhat + c   (2 MFMA, accumulator preloaded with c),   dH_ext = Wr.de' (2),   d ehat = Wp'.dGE (4);
// their weights are 12 A-operand registers (the quad-lane layout of the forward needs 48 registers or 12 LDS reads per step here:
// the round-2 backward in that layout ran at two workgroups per CU, 314 us at config 3 against 226 us for this kernel).
// LayerNorm sums run over the four 16-lane rows (v_permlane16_swap / v_permlane32_swap); dQ of a row is the 16-lane
// transposed reduction of the wide kernels (reduce16_keep_own: p IS the DPP row) -- no dA batch, no LDS.  K and V of the
// lane's key live in LDS ([u][lane][4]: conflict-free 16-byte reads); the three operand tiles of the weight-gradient products
// share ONE buffer (DS operations of a wave execute in order).  ~12 KB of LDS per wave and <= 168 VGPRs: THREE workgroups per CU.
// A key tile shared by two waves (balanced ranges) is parked in its final dkvp slot by the later wave and completed by the
// earlier one (both waves sit on one CU: plain stores, acknowledged before an LDS flag goes up; no LDS park area).
#define NRW_OPW 20                                   // row stride of the operand tile (floats): 16-byte aligned rows, spread banks
#define NRW_M_WAVE (2048 + 16 * NRW_OPW)             // floats per wave: K [4][64][4] | V [4][64][4] | operand tile [16][NRW_OPW]
// NW waves per workgroup: 4, or 8 for a launch with at most one workgroup per CU (BASELINE config 4 as specified: B = 16 per GPU, 240
// workgroups -- with four waves every SIMD holds ONE wave and a 16-pair step costs it 2.8 k cycles against 1.45 k per SIMD when
// three waves interleave; the section timers: 56 % of such a workgroup is its row loop).  Eight waves split the key tiles eight ways;
// the node-side prologue stays four waves' work (the others meet its barriers), whose scratch lies below the tile areas of waves 4-7.
template <bool BF, int FEAT, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 3 : 2) k_narrow_bwd(BlockArgs a) {
  seed_from_device(a);
#ifdef NRW_TIMING
  unsigned nacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  typedef NrwLd<BF> LD;
  constexpr bool MV = NRW_MMA_BF_VALUES(BF), MB = NRW_MMA_BF(BF);   // bf16 MFMAs: value path (projection recompute) / gradient path
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, q = lane >> 4;
  const int N = a.N, TL = a.TL;   // rows per workgroup: 16, or 8 / 4 for launches that would not fill the chip (layout())
  const int wg = a.xcd ? egt_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int b, lr;
  egt_group_order(wg, a.B, a.NLR, N, b, lr, TL);
  const int l_begin = lr * TL, l_end = min(N, l_begin + TL), nl = l_end - l_begin;
  const bool gated = FEAT >= 0 ? (FEAT & NRW_F_GATED) != 0 : (a.flags & EGT_BF_GATE) != 0;
  const bool clip = FEAT >= 0 ? (FEAT & NRW_F_CLIP) != 0 : (a.flags & EGT_BF_CLIP) != 0;
  const bool ln_on = (a.flags & EGT_BF_NO_EDGE_LN) == 0;
  constexpr int AREA = NW * NRW_M_WAVE > BWD_PRO_WS ? NW * NRW_M_WAVE : BWD_PRO_WS;
  static_assert(NW * 528 <= AREA, "edge partial staging must fit the per-wave area");
  float* kt = sm + wave * NRW_M_WAVE;      // [4 u][64 lanes][4]: K[4u .. 4u+3] of (key p, head pair q)
  float* vt = kt + 1024;                   // the same for V
  float* op = vt + 1024;                   // [16 pairs][NRW_OPW]
  float* qd = sm + AREA;                   // [TL][QD_LD]
  volatile int* pflag = reinterpret_cast<volatile int*>(qd + TL * QD_LD);   // [NW]: wave w parked its partial
  if (threadIdx.x < NW) pflag[threadIdx.x] = 0;
  // eight waves (two per SIMD at most: the registers are there): the node-side prologue's own global inputs -- the partial gather
  // included -- are requested by waves 0-3 ahead of the row staging and its barrier instead of behind them
  BwdProRegs proR;
  if (NW == 8 && wave < 4) bwd_node_prologue_load<NRW_DE>(a, proR, b, l_begin);
  bwd_stage_rows<64 * NW, NW == 4 ? 3 : 2>(a, qd, b, l_begin, nl);
  NSTMP(0);   // staging issued
  const int ntile = (N + 15) / 16;
  // Work of a wave: key tiles w, w+NW, ... when the tile count is a multiple of NW (or < NW); otherwise the ntile x nl (tile, row)
  // steps are cut into NW CONTIGUOUS equal ranges and a tile that straddles two ranges is shared by neighbouring waves.
#ifdef NRW_NO_BALANCE
  const bool balance = false;
#else
  const bool balance = ntile >= NW && (ntile % NW) != 0;
#endif
  const int T = ntile * nl;
  const int t0 = balance ? (wave * T) / NW : 0, t1 = balance ? ((wave + 1) * T) / NW : 0;
  const int mt_first = balance ? t0 / nl : wave, mt_last = balance ? (t1 - 1) / nl : ntile - 1, mt_step = balance ? 1 : NW;
  // e / de' run two steps ahead of the (tile, row) sequence, ACROSS tile boundaries: the last two rows of a tile request the first
  // two rows of the wave's next tile (a tile switch then costs the K / V round trip only)
  const size_t ugraph = (size_t)b * N * N;      // wave-uniform pair index of the graph's first pair
  typename LD::raw en[2] = {}, dn[2] = {};
  int cmt = mt_first, cli = balance ? t0 - mt_first * nl : 0;   // request cursor: the (tile, row) step two ahead of the one being worked on
  auto request = [&](int slot) __attribute__((always_inline)) {
    if (cmt <= mt_last) {   // (wave-uniform)
      const int lo = min(cmt * 16 + p, N - 1) * NRW_DE + 2 * q;
      const size_t pr = ugraph + (size_t)(l_begin + cli) * N;
      en[slot] = LD::uload(a.e, pr, lo);
      dn[slot] = LD::uload(a.de_out, pr, lo);
      if (++cli >= ((balance && cmt == mt_last) ? t1 - cmt * nl : nl)) { cli = 0; cmt += mt_step; }
    }
  };
  // K / V of the lane's key of tile mt -> the wave's LDS tiles
  auto fill_kv = [&](int mt) __attribute__((always_inline)) {
    const size_t rowm = (size_t)b * N + min(mt * 16 + p, N - 1);
    const float4* kp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 64 + q * 16);
    const float4* vp = reinterpret_cast<const float4*>(a.qkvp + rowm * QKVP + 128 + q * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // (the previous tile's reads of kt / vt are earlier in program order)
      *reinterpret_cast<float4*>(kt + u * 256 + lane * 4) = kp[u];
      *reinterpret_cast<float4*>(vt + u * 256 + lane * 4) = vp[u];
    }
  };
  // ---- A operands (lane = row m = p of the product, k index q) and the accumulator preload ----
  const int jm = p & 3, hm = 2 * (p >> 2) + jm;   // rows 4q'+0, 4q'+1 of a result carry head / channel 2q'+0, 2q'+1; rows 4q'+2, 4q'+3 are unused
  float c2r[4];
  NrwOp<MV, 2> pwA;
  NrwOp<MB, 2> wrA;
  NrwOp<MB, 4> wdA;
  auto load_weights = [&]() __attribute__((always_inline)) {
    pwA = nrw_op<MV>(a.pw[(2 * q) * 16 + p], a.pw[(2 * q + 1) * 16 + p]);           // [gates | E] column p from channels 2q, 2q+1
    wrA = nrw_op<MB>(jm < 2 ? a.Wr[hm * NRW_DE + 2 * q] : 0.0f,                       // dH_ext of head hm from de' channels 2q, 2q+1
                     jm < 2 ? a.Wr[hm * NRW_DE + 2 * q + 1] : 0.0f);
    wdA = nrw_op<MB>(jm < 2 ? a.pw[hm * 16 + 4 * q] : 0.0f, jm < 2 ? a.pw[hm * 16 + 4 * q + 1] : 0.0f,   // d ehat of channel hm from
                     jm < 2 ? a.pw[hm * 16 + 4 * q + 2] : 0.0f, jm < 2 ? a.pw[hm * 16 + 4 * q + 3] : 0.0f);  // dGE columns 4q .. 4q+3
#pragma unroll
    for (int r = 0; r < 4; ++r) c2r[r] = a.pw[16 * 16 + 4 * q + r];
  };
  // (Eight waves, measured and not kept: waves 4-7 fetching their weights, first K / V tiles and first e / de' rows while waves 0-3 run
  //  the prologue -- its own loads queue behind them: prologue 16.8 k -> 22.0 k cycles, launch 35.7 -> 36.6 us; the two waves of a SIMD
  //  taking turns at s_setprio 1 step by step -- the older wave's 26 k / the younger's 37 k cycles become 29 k / 37 k: the same launch;
  //  waves 4-7 entering the row loop 0.5 k / 1.3 k / 2.6 k cycles late (s_sleep: out of lock step): 35.7 / 35.8 / 36.3 against 36.0 us.)
  if (a.pro) {
    __syncthreads();
    if (NW == 4) bwd_node_prologue<NRW_DE>(a, sm, qd, b, l_begin, wg);
    else if (wave < 4) bwd_node_prologue_finish<NRW_DE>(a, sm, qd, b, l_begin, wg, proR);
    else bwd_node_prologue_idle(a);   // the prologue is four waves' work: the others only meet its barriers
  }
  __syncthreads();   // prologue scratch dead, qd rows complete
  NSTMP(1);   // node-side prologue
  load_weights();
  v4f accT = {0.f, 0.f, 0.f, 0.f}, accR = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f;   // column p of dGE summed over the pairs 4 s + q of every step (the B operands of the T product): ONE register
  const float hcst = p == 8 ? 1.0f : 0.0f;   // columns 8..15 of the [H_hat | 1] operand
  request(0);
  request(1);
  for (int mt = mt_first; mt <= mt_last; mt += mt_step) {
    const int r0 = (balance && mt == mt_first) ? t0 - mt * nl : 0;          // rows [r0, r1) of the workgroup's nl
    const int r1 = (balance && mt == mt_last) ? t1 - mt * nl : nl;
    const int m0 = mt * 16, m = m0 + p;
    const bool kvalid = m < N;
    const int mc = kvalid ? m : N - 1;
    const size_t rowm = (size_t)b * N + mc;
    float dKa[16], dVa[16];
    fill_kv(mt);
#pragma unroll
    for (int i = 0; i < 16; ++i) { dKa[i] = 0.f; dVa[i] = 0.f; }
    const float kadd = (a.km && a.km[rowm] == 0) ? -EGT_NEG : 0.0f;
    const MaskRegs mr{make_float2(1.f, 1.f), 0};
    const uint32_t pcol = (uint32_t)((size_t)b * N * N + mc);   // pair index of (row 0 of the graph, key mc), mod 2^32: the mask-RNG counter
    const int loff = mc * NRW_DE + 2 * q;         // the lane's element offset inside a pair row
    NSTMP(2);   // tile set-up: weights (first tile), K / V -> LDS, first e / de' requests
    for (int li = r0; li < r1; ++li) {
      const int l = l_begin + li;
      const uint32_t pair = pcol + (uint32_t)(l * N);
      float2 ev = LD::cvt(en[0]), dyv = LD::cvt(dn[0]);
      const typename LD::raw dyraw = dn[0];
      en[0] = en[1]; dn[0] = dn[1];
      request(1);
      if (!kvalid) { ev = make_float2(0.f, 0.f); dyv = make_float2(0.f, 0.f); }   // a key past N: zero tile row
      // ---- norm_edge (recompute): the pair's 8 channels sit in lanes p, p+16, p+32, p+48 ----
      float x0 = ev.x, x1 = ev.y;
      const float mu = ln_on ? nrw_sum4rows(x0 + x1) * 0.125f : 0.0f;
      x0 -= mu; x1 -= mu;
      const float var = nrw_sum4rows(fmaf(x0, x0, x1 * x1)) * 0.125f;
      const float rstd = ln_on ? __builtin_amdgcn_rsqf(var + a.ln_eps) : 1.0f;
      x0 *= rstd; x1 *= rstd;
      // ---- projections (acc[r] = column 4q + r) and dH_ext (dhx[j] = head 2q + j) on the matrix core ----
      const v4f acc = nrw_mm<MV, 2>(pwA, nrw_op<MV>(x0, x1), (v4f){c2r[0], c2r[1], c2r[2], c2r[3]});
      v4f dh4;
      if constexpr (MB) dh4 = nrw_mm<true, 2>(wrA, nrw_op_raw(kvalid ? dyraw : 0u), (v4f){0.f, 0.f, 0.f, 0.f});   // de' is bfloat16 in memory already
      else dh4 = nrw_mm<false, 2>(wrA, nrw_op<false>(dyv.x, dyv.y), (v4f){0.f, 0.f, 0.f, 0.f});
      // ---- weight-gradient operand A = [xhat | de'] of the step's 16 pairs (transposed through the operand tile) ----
      float wa[4], wb1[4], wb2[4];
      *reinterpret_cast<float2*>(op + p * NRW_OPW + 2 * q) = make_float2(x0, x1);
      *reinterpret_cast<float2*>(op + p * NRW_OPW + 8 + 2 * q) = dyv;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) wa[s4] = op[(4 * s4 + q) * NRW_OPW + p];
      asm volatile("" ::: "memory");
      // ---- logits, softmax / gate backward ----
      const float* qr = qd + li * QD_LD;
      const float4* qp = reinterpret_cast<const float4*>(qr + q * 16);
      const float4* dp = reinterpret_cast<const float4*>(qr + 64 + q * 16);
      float dots[2], dAd[2];
      {
        float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 v = qp[u], w = dp[u];
          const float4 kf = *reinterpret_cast<const float4*>(kt + u * 256 + lane * 4);
          const float4 vf = *reinterpret_cast<const float4*>(vt + u * 256 + lane * 4);
          d0 = fmaf(v.x, kf.x, d0); d1 = fmaf(v.y, kf.y, d1);
          d0 = fmaf(v.z, kf.z, d0); d1 = fmaf(v.w, kf.w, d1);
          e0 = fmaf(w.x, vf.x, e0); e1 = fmaf(w.y, vf.y, e1);
          e0 = fmaf(w.z, vf.z, e0); e1 = fmaf(w.w, vf.w, e1);
          if (u == 1) {   // two of the four 16-byte pieces of Q / dV_att / K / V in flight at a time (32 registers instead of 64 at the
                          // kernel's register peak); the pins keep the partial dot products on this side of the fence
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(e0), "+v"(e1));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        dots[0] = d0; dots[1] = d1; dAd[0] = e0; dAd[1] = e1;
      }
      const float4 st0 = *reinterpret_cast<const float4*>(qr + 128 + q * 8);
      const float4 st1 = *reinterpret_cast<const float4*>(qr + 128 + q * 8 + 4);
      float dge[4], hh[2], dA[2], at[2];
      {
        float xl[2], gl[2], inr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float araw = dots[j] * a.scale;
          float ah = araw;
          inr[j] = 1.0f;
          if (clip) {
            ah = nrw_min(nrw_max(araw, a.clip_lo), a.clip_hi);
            inr[j] = (ah == araw) ? 1.0f : 0.0f;   // inside [lo, hi]  <=>  the clamp left it alone
          }
          hh[j] = ah + acc[2 * j + 1];
          xl[j] = hh[j];
          gl[j] = acc[2 * j];
        }
        apply_masks<false>(a, kadd, mr, (size_t)(pair * (uint32_t)BH), q, xl, gl);   // (the counter is taken mod 2^32 there)
        if (!kvalid) { xl[0] = xl[1] = -3.0e38f; gl[0] = gl[1] = -3.0e38f; }   // a key past N: S = 0, gate = 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float smax = j ? st1.x : st0.x, rsum = j ? st1.y : st0.y, delta = j ? st1.z : st0.z;
          const float S = __expf(xl[j] - smax) * rsum;
          const float g = gated ? egt_sigmoid(gl[j]) : 1.0f;
          const float dS = dAd[j] * g;
          const float dGl = gated ? dAd[j] * S * g * (1.0f - g) : 0.f;
          const float dH = fmaf(S, dS - delta, dh4[j]);
          dA[j] = dH * inr[j] * a.scale;
          at[j] = S * g;
          dge[2 * j] = dGl;
          dge[2 * j + 1] = dH;
        }
      }
      // ---- B operands of the weight-gradient products: dGE, then [H_hat | 1 | 0] -- the SAME tile, in DS order ----
      *reinterpret_cast<float4*>(op + p * NRW_OPW + 4 * q) = make_float4(dge[0], dge[1], dge[2], dge[3]);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) wb1[s4] = op[(4 * s4 + q) * NRW_OPW + p];
      asm volatile("" ::: "memory");
      ssum += (wb1[0] + wb1[1]) + (wb1[2] + wb1[3]);
      *reinterpret_cast<float2*>(op + p * NRW_OPW + 2 * q) = make_float2(hh[0], hh[1]);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { const float t = op[(4 * s4 + q) * NRW_OPW + p]; wb2[s4] = p < 8 ? t : hcst; }
      asm volatile("" ::: "memory");   // the next step's tile writes stay behind these reads
      // ---- d ehat = Wp'.dGE (channels 2q, 2q+1 in d4[0], d4[1]) ----
      const v4f d4 = nrw_mm<MB, 4>(wdA, nrw_op<MB>(dge[0], dge[1], dge[2], dge[3]), (v4f){0.f, 0.f, 0.f, 0.f});
      // ---- dQ of the row over this tile's 16 keys -> HBM (summed over key tiles by the next prologue / k_node_bwd) ----
      // (K of the lane's key is read from its LDS tile a second time: sixteen registers held across the softmax phase were what
      //  spilled the fp32 instances at three workgroups per CU)
      float dq[16];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 kf = *reinterpret_cast<const float4*>(kt + u * 256 + lane * 4);
        dq[4*u] = kf.x * dA[0]; dq[4*u+1] = kf.y * dA[1]; dq[4*u+2] = kf.z * dA[0]; dq[4*u+3] = kf.w * dA[1];
      }
      a.dqp[(((size_t)b * ntile + mt) * N + l) * 64 + lane] = reduce16_keep_own(dq, p);
      // ---- dK / dV (Q / dV_att of the row re-read: not held across the softmax phase) ----
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = qp[u], w = dp[u];
        dKa[4*u]   = fmaf(dA[0], v.x, dKa[4*u]);   dKa[4*u+1] = fmaf(dA[1], v.y, dKa[4*u+1]);
        dKa[4*u+2] = fmaf(dA[0], v.z, dKa[4*u+2]); dKa[4*u+3] = fmaf(dA[1], v.w, dKa[4*u+3]);
        dVa[4*u]   = fmaf(at[0], w.x, dVa[4*u]);   dVa[4*u+1] = fmaf(at[1], w.y, dVa[4*u+1]);
        dVa[4*u+2] = fmaf(at[0], w.z, dVa[4*u+2]); dVa[4*u+3] = fmaf(at[1], w.w, dVa[4*u+3]);
      }
      // ---- weight gradients over the step's 16 pairs: T += [xhat | de']^T.dGE, R += [xhat | de']^T.[H_hat | 1] ----
      {
        const NrwOp<MB, 4> oa = nrw_op<MB>(wa[0], wa[1], wa[2], wa[3]);
        accT = nrw_mm<MB, 4>(oa, nrw_op<MB>(wb1[0], wb1[1], wb1[2], wb1[3]), accT);
        accR = nrw_mm<MB, 4>(oa, nrw_op<MB>(wb2[0], wb2[1], wb2[2], wb2[3]), accR);
      }
      // ---- LayerNorm backward; de = de' + ... ----
      {
        float m1 = d4[0] + d4[1], m2 = fmaf(d4[0], x0, d4[1] * x1);
        if (ln_on) { nrw_sum4rows_pair(m1, m2); m1 *= 0.125f; m2 *= 0.125f; } else { m1 = 0.f; m2 = 0.f; }
        float2 o;
        o.x = dyv.x + rstd * (d4[0] - m1 - x0 * m2);
        o.y = dyv.y + rstd * (d4[1] - m1 - x1 * m2);
        if (kvalid) LD::ustore(a.de, ugraph + (size_t)l * N, loff, o);
      }
    }
    NSTMP(3);   // row loop
    // ---- the tile's dK / dV: the workgroup's partial slot of key m ----
    float4* ko = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + mc) * 2 + 0) * 4 + q) * 16);
    float4* vo = reinterpret_cast<float4*>(a.dkvp + (((((size_t)b * a.NLR + lr) * N + mc) * 2 + 1) * 4 + q) * 16);
    if (r1 < nl && r0 == 0) {   // the tile's last rows were done by the next wave, long ago: its partial sits in the slot
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
    if (r0 > 0) {   // the tile's first rows belong to the previous wave: what was just stored is this wave's partial, parked in the
                    // slot (same CU: the stores are acknowledged by the L2 before the flag goes up, the slot was never in this L1)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) pflag[wave] = 1;
    }
  }
  NSTMP(4);   // dK / dV stores, parked partials
  // ---- edge-parameter gradient partials of the workgroup: T [16][16] | s [16] | R [16][16] ----
  ssum = nrw_sum4rows(ssum);   // column p: the four lane rows hold the pairs q, 4 + q, 8 + q, 12 + q
  __syncthreads();
  {
    float* ep = sm + wave * 528;
    const int col = p, kq = q;   // MFMA result layout: rows 4 kq + r4, column col
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int row = 4 * kq + r4;
      ep[row * 16 + col] = row < 8 ? accT[r4] : 0.f;                          // T = rows 0..7 of [xhat | de']^T.dGE (channels >= De: 0)
      ep[272 + ((row + 8) & 15) * 16 + col] = row >= 8 ? accR[r4] : 0.f;      // R = rows 8..15 of [xhat | de']^T.[H_hat | 1]
    }
    if (q == 0) ep[256 + p] = ssum;
  }
  __syncthreads();
  float* out = a.epart + (size_t)wg * 528;
  for (int i = threadIdx.x; i < 528; i += 64 * NW) {
    float v = (sm[i] + sm[528 + i]) + (sm[2 * 528 + i] + sm[3 * 528 + i]);
    if (NW == 8) v += (sm[4 * 528 + i] + sm[5 * 528 + i]) + (sm[6 * 528 + i] + sm[7 * 528 + i]);
    out[i] = v;
  }
#ifdef NRW_TIMING
  NSTMP(5);   // workgroup partials (waits for the slowest wave)
  if (a.dbg && lane == 0 && wave < 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = nacc[i];
  }
#endif
}

#ifdef NRW_TIMING
static unsigned* g_nt_dev = nullptr;
static unsigned* g_nt_dev2 = nullptr;
static double g_nt2_sum[8];
static int g_nt_n = 0;
static double g_nt_sum[8];
static long g_nt_launch = 0, g_nt_waves = 0;
static void nrw_timing_report() {
  static const char* nm[] = {"staging issue", "node-side prologue + sync", "tile set-up (K/V, first e)", "row loop", "dK/dV stores + park", "workgroup partials + sync"};
  if (!g_nt_waves) return;
  double tot = 0; for (int i = 0; i < 6; ++i) tot += g_nt_sum[i];
  fprintf(stderr, "[egt] k_narrow_bwd section cycles per wave (mean over %ld waves, %ld launches; total %.0f):\n", g_nt_waves, g_nt_launch, tot / g_nt_waves);
  for (int i = 0; i < 6; ++i) fprintf(stderr, "    %-30s %10.0f  (%.1f %%)\n", nm[i], g_nt_sum[i] / g_nt_waves, 100.0 * g_nt_sum[i] / tot);
  fprintf(stderr, "    prologue (pro = 2 launches, per wave): load issue %.0f | first round trip -> rows in LDS %.0f | sync + LN fwd + 48 MFMA %.0f | dQKV out + sync + LN bwd + col sums %.0f | wo/va + sync %.0f | 16 MFMA + delta partials %.0f | dbo + sync + delta %.0f\n",
          g_nt2_sum[0] / g_nt_waves, g_nt2_sum[1] / g_nt_waves, g_nt2_sum[2] / g_nt_waves, g_nt2_sum[3] / g_nt_waves, g_nt2_sum[4] / g_nt_waves, g_nt2_sum[5] / g_nt_waves, g_nt2_sum[6] / g_nt_waves);
}
#endif

void egt_narrow_launch_bwd(BlockArgs& a, int nwg, hipStream_t st) {
  const int full = NRW_F_GATED | NRW_F_CLIP;
  const int feat = ((a.flags & EGT_BF_GATE) ? NRW_F_GATED : 0) | ((a.flags & EGT_BF_CLIP) ? NRW_F_CLIP : 0);
  constexpr int AREA4 = 4 * NRW_M_WAVE > BWD_PRO_WS ? 4 * NRW_M_WAVE : BWD_PRO_WS, AREA8 = 8 * NRW_M_WAVE;
  static_assert(AREA8 >= BWD_PRO_WS, "tile areas of eight waves cover the prologue's scratch");
  static const int nw_forced = getenv("EGT_NRW_BWD_WAVES") ? atoi(getenv("EGT_NRW_BWD_WAVES")) : 0;   // 4 | 8 (tests, A/B)
  const bool w8 = nw_forced == 8 || (nw_forced != 4 && nwg <= egt_device_cus() && a.N >= 64);   // at most one workgroup per CU: eight waves share its key tiles
  const size_t lds = ((size_t)(w8 ? AREA8 : AREA4) + BWD_TL * QD_LD + 8) * 4;   // four waves: 47 KB, three workgroups per CU
#define NRW_BWD(BF_, FEAT_)                                                                     \
  do {                                                                                            \
    if (w8) { EGT_MAX_LDS_ONCE(k_narrow_bwd<BF_, FEAT_, 8>);                                       \
              EGT_LAUNCH("k_block_bwd", (k_narrow_bwd<BF_, FEAT_, 8>), dim3(nwg), dim3(512), lds, st, a); } \
    else { EGT_MAX_LDS_ONCE(k_narrow_bwd<BF_, FEAT_, 4>);                                          \
           EGT_LAUNCH("k_block_bwd", (k_narrow_bwd<BF_, FEAT_, 4>), dim3(nwg), dim3(256), lds, st, a); } \
  } while (0)
#ifdef NRW_TIMING
  if (g_nt_n < nwg) {
    if (g_nt_dev) (void)hipFree(g_nt_dev);
    (void)hipMalloc(&g_nt_dev, (size_t)nwg * 32 * sizeof(unsigned));
    if (g_nt_dev2) (void)hipFree(g_nt_dev2);
    (void)hipMalloc(&g_nt_dev2, (size_t)nwg * 32 * sizeof(unsigned));
    (void)hipMemset(g_nt_dev2, 0, (size_t)nwg * 32 * sizeof(unsigned));
    if (!g_nt_n) atexit(nrw_timing_report);
    g_nt_n = nwg;
  }
  a.dbg = g_nt_dev; a.dbg2 = g_nt_dev2;
#endif
  if (a.bf16) { if (feat == full) NRW_BWD(true, NRW_F_GATED | NRW_F_CLIP); else NRW_BWD(true, NRW_F_RUNTIME); }
  else { if (feat == full) NRW_BWD(false, NRW_F_GATED | NRW_F_CLIP); else NRW_BWD(false, NRW_F_RUNTIME); }
#undef NRW_BWD
#ifdef NRW_TIMING
  (void)hipStreamSynchronize(st);
  if (++g_nt_launch > 20) {
    static unsigned* h = nullptr; static int hn = 0;
    if (hn < nwg) { free(h); h = (unsigned*)malloc((size_t)nwg * 32 * sizeof(unsigned)); hn = nwg; }
    (void)hipMemcpy(h, g_nt_dev, (size_t)nwg * 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int w = 0; w < nwg * 4; ++w) { for (int i = 0; i < 8; ++i) g_nt_sum[i] += h[w * 8 + i]; ++g_nt_waves; }
    (void)hipMemcpy(h, g_nt_dev2, (size_t)nwg * 32 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int w = 0; w < nwg * 4; ++w) for (int i = 0; i < 8; ++i) g_nt2_sum[i] += h[w * 8 + i];
  }
#endif
}
