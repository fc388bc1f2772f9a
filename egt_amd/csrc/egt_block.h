// Shared declarations of the fused attention-block path (egt_block.hip: pair
// kernels; egt_node.hip: node-side kernels, partial reductions, parameter grads).
#pragma once
#include "egt_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define BH 8        // heads (all reference configs)
#define QKVP 192    // packed floats per node row: [3][4 head-pairs][8 k][2]

struct BlockArgs {
  int B, N, De, DK, Dh;  // DK = per-head dim (<= 8)
  uint32_t flags;
  int bf16;   // edge tensors e / e' / de' / de are bf16 in HBM (arithmetic stays fp32)
  float clip_lo, clip_hi, scale, ln_eps;
  uint32_t rm_thr, s0, s1;
  const uint32_t* sd;   // EGT_BF_SEED_DEVICE: the two words XORed into (s0, s1) at kernel entry (seed_from_device)
  int rng_rm;
  int TL, NLR;  // backward: query rows per workgroup, row-ranges per graph
  int NQP;      // backward: dQ partials per row (key tiles) in dqp [B][NQP][N][64]
  int epi;      // forward epilogue in k_block_fwd: 0 none, 1 dense_mha+res, 2 + next block's norm_mha/dense_qkv
  int RGF;      // k_block_fwd: query rows per workgroup (<= 16)
  int xcd;      // pair kernels: XCD-aware workgroup order
  // backward prologue of k_block_bwd_v4 (Dh = 64): the node-side step between two pair kernels is
  // done per 16-row workgroup inside the lower layer's kernel.  pro = 0: dV_att / delta come from
  // dvp / stats (k_node_bwd ran); 1: chain top, dh' rows = dh_out; 2: dh' rows are computed here
  // from the partials of the layer ABOVE (up_*): dQKV gather, d h_ln = dQKV.Wqkv^T, LN backward.
  int pro;
  const float *up_h, *up_nm_g, *up_Wqkv, *up_dh_out, *up_dqp, *up_dkvp;
  float *up_dqkv_sv, *up_spart;
  float *sbo;    // per-workgroup partials of this layer's dense_mha bias gradient
  int spart_n, sbo_n;   // how many workgroup partials the reduction finds in spart / sbo
  int wpart_n;          // ... and in wpart: the row chunks egt_node_launch_wgrads launched (stored there, read by the reduction)
  int guard;    // backward phase guards (always 0 in production, see k_block_bwd_v4)
  unsigned* dbg; unsigned dbg_t0;   // measurement builds of egt_narrow.hip (-DNRW_TIMING): per-wave section cycle sums
  int prep;     // node kernels: add the edge-weight preparation workgroup
  const float *nx_nm_g, *nx_nm_b, *nx_Wqkv, *nx_bqkv;   // next block (epi == 2)
  // MFMA-fragment-major copies of Wqkv / Wo (prepared by the forward beside pw, kept in `saved`): what the node-side epilogue /
  // prologue of the pair kernels load as lane-linear 16-byte pieces (WFRAG_* below); own layer, layer above (backward), next layer (forward)
  float* wfrag;
  const float *up_wfrag, *nx_wfrag;
  float* nx_qkvp;
  // params
  const float *ne_g, *ne_b, *Wg, *bg, *We, *be, *nm_g, *nm_b, *Wqkv, *bqkv, *Wo, *bo, *Wr, *br;
  // tensors
  const float *h, *e, *M;
  const uint8_t *km, *rm;
  float *h_out, *e_out;
  // saved (forward -> backward)
  float *v_att, *stats, *qkvp;
  // workspace
  float *pw;        // prepared edge weights: Wp[DEP][16], c2[16]
  float *dvp, *dqp, *dkvp, *epart, *ered;
  float *spart, *wpart;   // node-side partials: small column sums per workgroup; dWqkv|dWo per row chunk
  float *dqkv_sv;         // dQKV rows [B*N][3Dh] kept for the deferred weight-gradient kernel
  // backward
  const float *dh_out, *de_out;
  float *dh, *de;
  float *g_ne_g, *g_ne_b, *g_Wg, *g_bg, *g_We, *g_be, *g_nm_g, *g_nm_b, *g_Wqkv, *g_bqkv, *g_Wo,
      *g_bo, *g_Wr, *g_br;
  unsigned* dbg2;   // -DNRW_TIMING: the node-side prologue's phase cycles (last member: the other translation units ignore it)
};

template <int DE>
struct Geo {
  static constexpr int TILES = (DE + 15) / 16;
  static constexpr int DEP = TILES * 16;
  static constexpr int NSLOT = DE / 4;          // 16-byte slots per pair row
  static constexpr int TILE_FLOATS = 16 * DE;   // one 16-pair tile
  static constexpr int NF4 = 4 * DE;            // float4s per tile
  static constexpr int EP = DEP * 16 + 16 + DEP * 16;  // edge partial: T, s, R
};

// row order of the 16 projection columns: i = 4*q + r ->
//   r=0: gate head 2q, r=1: edge-bias head 2q, r=2: gate head 2q+1, r=3: edge-bias head 2q+1
__host__ __device__ inline int col_is_gate(int i) { return ((i & 1) == 0); }
__host__ __device__ inline int col_head(int i) { return 2 * (i >> 2) + ((i >> 1) & 1); }


#define BWD_TL 16   // backward: query rows per workgroup
int egt_device_cus();   // compute units of the current device (egt_block.hip)
#define NODE_RC 32  // node rows per workgroup in the node kernels
// Fragment-major weights of one layer (floats; every array is [wave][step][64 lanes] float4, zero where the padded 64-wide row has no
// channel): the 16-byte pieces a lane feeds to its MFMAs, so that a load instruction reads 1 KB of consecutive memory instead of
// 64 pieces at a 768-byte stride (measured: -1.2 us per backward launch, -0.6 us per forward launch at the headline batch).
//   BWQ [4][12][64]: Wqkv[16 w + p][48 q + 4 s ..]      B operand of d h_ln = dQKV . Wqkv^T   (backward prologue, layer above)
//   BWO [4][4][64] : Wo[16 w + p][16 q + 4 s ..]        B operand of dV_att = dh' . Wo^T      (backward prologue)
//   FWO [4][4][64] : Wo[4 (4 s4 + i) + q][16 w + p]     B operand of h' = V_att . Wo          (forward epilogue)
//   FWQ [4][3][4][64]: Wqkv[4 (4 s4 + i) + q][col(16 (w + 4 j) + p)]   next layer's QKV       (forward epilogue)
#define WFRAG_BWQ 0
#define WFRAG_BWO 12288
#define WFRAG_FWO 16384
#define WFRAG_FWQ 20480
#define WFRAG_FLOATS 32768

// De = 8 VALU pair kernels (egt_narrow.hip)
void egt_narrow_launch_fwd(BlockArgs& a, hipStream_t st);
void egt_narrow_launch_bwd(BlockArgs& a, int nwg, hipStream_t st);   // same a.pro / partial-buffer contract as k_block_bwd_v4r


// launchers implemented in egt_node.hip
void egt_node_launch_pre(BlockArgs& a, hipStream_t st);
void egt_node_launch_post(BlockArgs& a, hipStream_t st);
void egt_node_launch_bwd(BlockArgs& a, const BlockArgs* dv_layer, bool do_pre, hipStream_t st);
void egt_node_launch_wgrads(BlockArgs* as, int n, hipStream_t st);   // n <= 64 layers
int egt_node_wgrad_chunks(int rows, int layers);   // row chunks (= partial slots in wpart) of a k_node_wgrads launch over `layers` layers
void egt_node_launch_prep(BlockArgs* as, int n, hipStream_t st);   // n <= 64 layers
bool egt_node_launch_pre_stack(BlockArgs* as, int n, hipStream_t st);   // layer 0's k_node_pre + the preparation of all n layers in one launch
void egt_node_launch_reduce(BlockArgs* as, int n, int nwg_bwd, int EP, hipStream_t st);
