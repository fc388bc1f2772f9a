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

#include "egt_block_fwd.h"   // k_block_fwd, k_block_fwd_r4
#include "egt_block_bwd.h"   // k_block_bwd_v4, k_block_bwd_v5, k_block_bwd_v4r

// ================================================================ host glue ====


// Run-time switches (read ONCE per process: the launch path does no getenv()) -- the complete list, see README.md:
//   EGT_NO_NARROW / EGT_NO_NARROW_FWD / EGT_NO_NARROW_BWD: De = 8 falls back from the De = 8 pair kernels (egt_narrow.hip) to the
//     MFMA-tile kernels (tests exercise both);  EGT_BWD_MATMUL=bf16x3: the backward's channel contractions as 3-term bf16 split
//     products (opt-in; default exact fp32);  EGT_BWD_TL / EGT_FWD_ROWS: query rows per backward / forward workgroup (tests / sweeps).
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
  size_t v_att, stats, qkvp, pw_sv, wfrag_sv, saved_total;   // pw_sv: LN-folded edge weights, wfrag_sv: fragment-major Wqkv / Wo -- prepared by the forward, reused by the backward
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
//    (pattern500k_n120, per launch: B = 16: 59.0 us at 16 rows, 43.3 at 8, 47.9 at 6; B = 32: 63.0 / 59.5 / 65.5) -- round 6, with
//    the eight-wave k_narrow_bwd for such launches: the shortest groups of >= 8 rows that keep the launch at ONE workgroup per CU
//    (N = 120: B = 16 -> 8 rows, B = 24 -> 12 rows: 51.5 -> 42.5 us, B = 32 -> 15 rows: 54.0 -> 48.4 us against 8 rows);
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
    // ... and 4 rows when even four-row groups leave every CU at most ONE workgroup (B = 16 per GPU at the headline shapes -- a
    // global batch of 128 over 8 GPUs: 64 sixteen-row workgroups on 512 slots -> 256 four-row ones: 23.8 k -> 30.1 k graphs/s with the
    // forward's same rule; at B = 24 / 32 four rows measure the same as / below eight: profiles/r06_pair_abl2.txt)
    const int slots = 2 * egt_device_cus();
    int g = slots / (d->B > 0 ? d->B : 1);
    const int cap = d->B * ((d->N + 3) / 4) <= slots / 2 ? (d->N + 3) / 4 : (d->N + 7) / 8;
    if (g > cap) g = cap;
    if (g < groups) g = groups;
    return (d->N + g - 1) / g;
  }
  if (d->B * groups <= egt_device_cus()) {
    // a launch of at most one 16-row workgroup per CU (it runs the eight-wave k_narrow_bwd): the shortest groups of >= 8 rows that still give
    // every CU at most ONE workgroup -- B = 16: N = 120 -> 8 rows (240 workgroups), N = 188 -> 12 rows (256; with 8 rows the launch was
    // 368 four-wave workgroups: 71.4 us against 58.4, `pattern500k_n188` 9.4 k -> 10.9 k graphs/s)
    if (d->N <= 8) return BWD_TL;
    for (int tl = 8; tl < BWD_TL; ++tl)
      if (d->B * ((d->N + tl - 1) / tl) <= egt_device_cus()) return tl;
    return BWD_TL;
  }
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
  L.wfrag_sv = o; o += al(Dh <= 64 ? WFRAG_FLOATS : 0);
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
  L.wpart = o; o += al((size_t)egt_node_wgrad_chunks((int)rows, 1) * (Dh * 3 * Dh + Dh * Dh));
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

// Which backward pair kernel launch_bwd takes for `d` when no mask TENSOR is passed (the in-kernel random mask is not one) and
// the node side is fused: the families of DESIGN.md section 4 by name.  For tests and bench lines; NULL when `d` is not covered.
extern "C" const char* egt_block_bwd_kernel(const egt_block_desc* d) {
  if (block_check(d, false)) return nullptr;
  const BlockLayout L = layout(d);
  const bool ml = (d->flags & EGT_BF_ATTN_MASK) != 0, bf = d->dtype == EGT_BF16;
  if (d->De == 8 && !ml && !block_env().no_narrow_bwd) return "k_narrow_bwd";
  if (d->De <= 16 && !ml) return "k_block_bwd_v4r";
  if (d->De >= 32 && !ml && !bf) return "k_block_bwd_v5";
  return "k_block_bwd_v4";
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
  a.wfrag = a.Dh <= 64 ? saved + L.wfrag_sv : nullptr;
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
// Dh = 8 DK <= 64 as a zero-padded 64-wide row.  (Wo / Wqkv reach those kernels through the fragment-major copies the preparation
// writes with scalar loads -- WFRAG_*, egt_block.h -- so parameter views that are not 16-byte aligned are covered as well.)
static bool node_fused_ok(const BlockArgs& a) {
  return a.Dh == BH * a.DK && a.DK >= 1 && a.DK <= 8 && a.wfrag != nullptr;
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
  // k_block_fwd: 16 query rows per workgroup (four per wave); a launch that leaves workgroup slots empty (two per CU) takes
  // more, shorter groups -- whole multiples of four rows, so the waves stay balanced -- while they fit one round
  // (ZINC-100K, B = 128, N = 37: 3 x 16 rows = 384 workgroups -> 4 x 12 rows = 512: the longest wave walks 3 rows instead of 4)
  a.RGF = 16;
  {
    static const int forced = getenv("EGT_FWD_ROWS") ? atoi(getenv("EGT_FWD_ROWS")) : 0;   // 4 .. 16 (tests)
    const int slots = 2 * egt_device_cus();
    int g = slots / (a.B > 0 ? a.B : 1);
    // (at least 8 rows per group -- 4 when even four-row groups leave every CU at most one workgroup: bwd_rows_per_wg has the numbers)
    const int cap = a.B * ((a.N + 3) / 4) <= slots / 2 ? (a.N + 3) / 4 : (a.N + 7) / 8;
    if (g > cap) g = cap;
    if (g > lgroups) a.RGF = (((a.N + g - 1) / g + 3) / 4) * 4;
    if (forced >= 4 && forced <= 16) a.RGF = forced;
  }
  const dim3 grid(a.B * ((a.N + a.RGF - 1) / a.RGF)), block(256);
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
// Returns the number of edge-parameter partials (workgroups) the pair kernel wrote into a.epart.
static int launch_bwd(BlockArgs& a, const BlockLayout& L, hipStream_t st, bool top, BlockArgs* below, BlockArgs* above, bool fuse) {
  using GG = Geo<DE>;
  int nep = L.nwg_bwd;
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
      a.up_h = above->h; a.up_nm_g = above->nm_g; a.up_Wqkv = above->Wqkv; a.up_wfrag = above->wfrag; a.up_dh_out = above->dh_out;
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
        if (!ml && !a.bf16) {   // LDS-DMA staged e tiles (k_block_bwd_v5), ragged N included
          const bool x3 = block_env().bwd_mm == EGT_MM_BF16X3;
          const size_t lds_v5 = ((size_t)V5_AREA(DE) + (size_t)BWD_TL * QD_LD + 3 * ((GG::TILES + 1) / 2) * 512 + 4) * 4;   // (+ 4: the parked-partial flags)
#define V5_LAUNCH(MM_, RAG_)                                                                                 \
  do {                                                                                                       \
    EGT_MAX_LDS_ONCE(k_block_bwd_v5<DE, MM_, RAG_>);                                                         \
    EGT_LAUNCH("k_block_bwd", (k_block_bwd_v5<DE, MM_, RAG_>), dim3(L.nwg_bwd), dim3(256), lds_v5, st, a);   \
  } while (0)
          if (full) { if (x3) V5_LAUNCH(EGT_MM_BF16X3, false); else V5_LAUNCH(0, false); }
          else { if (x3) V5_LAUNCH(EGT_MM_BF16X3, true); else V5_LAUNCH(0, true); }
#undef V5_LAUNCH
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
  return nep;
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
  int nep = L.nwg_bwd;
  DISPATCH_BDE(desc->De, nep = launch_bwd<DE>(a, L, (hipStream_t)stream, true, nullptr, nullptr, node_fused_ok(a)));
  egt_node_launch_wgrads(&a, 1, (hipStream_t)stream);
  egt_node_launch_reduce(&a, 1, nep, L.EP, (hipStream_t)stream);  // partial sums + edge param grads
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
  a.wfrag = a.Dh <= 64 ? bs + L.wfrag_sv : nullptr;
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
  // (the preparation rides along with layer 0's k_node_pre as extra workgroups: one launch less in front of every step)
  for (int l = 0; l < layers; ++l) as[l].prep = 0;
  int prev_epi = 0;
  if (egt_node_launch_pre_stack(as, layers, (hipStream_t)stream)) prev_epi = 2;   // layer 0's packed QKV rows exist
  else egt_node_launch_prep(as, layers, (hipStream_t)stream);
  for (int l = 0; l < layers; ++l) {
    BlockArgs& a = as[l];
    a.epi = 1;
    if (l + 1 < layers) {
      const BlockArgs& nx = as[l + 1];
      a.epi = 2;
      a.nx_nm_g = nx.nm_g; a.nx_nm_b = nx.nm_b; a.nx_Wqkv = nx.Wqkv; a.nx_wfrag = nx.wfrag; a.nx_bqkv = nx.bqkv;
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
  int nep = L.nwg_bwd;
  for (int l = layers - 1; l >= 0; --l) {
    as[l].prep = 0;   // the LN-folded edge weights were prepared by the forward and live in `saved`
    DISPATCH_BDE(desc->De, nep = launch_bwd<DE>(as[l], L, (hipStream_t)stream, l == layers - 1, l > 0 ? &as[l - 1] : nullptr,
                                                   l + 1 < layers ? &as[l + 1] : nullptr, fuse));
  }
  egt_node_launch_wgrads(as, layers, (hipStream_t)stream);
  egt_node_launch_reduce(as, layers, nep, L.EP, (hipStream_t)stream);
  EGT_HIP_LAUNCH_CHECK("egt_stack_bwd");
  return EGT_OK;
}
