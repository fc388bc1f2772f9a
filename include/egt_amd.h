/*
 * egt_amd.h — C-ABI of the MI355X-native EGT edge-augmented attention path.
 *
 * The reference (shamim-hussain/egt) has NO FFI for this path: its seam is a
 * pure-Python plugin registry (lib/base/track_layers/base.py:43-60) through
 * which lib/models/graph_xformer_model_base.py:117-131 instantiates
 * lib/models/egt_layers.py:4 `EGT`.  This header is the boundary a native
 * replacement of that layer binds (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer valid on `stream` (a hipStream_t passed
 *     as void*; NULL = the null stream); the caller allocates everything; the
 *     library never allocates, frees or keeps user memory and never syncs.
 *   - tensors are dense row-major fp32 in the reference's layouts:
 *       qkv   [B,N,3*d*H]   channel c = s*d*H + k*H + h   (egt_layers.py:73-76)
 *       E,G,M,h_hat,a_tild [B,N,N,H]  (h innermost)       (egt_layers.py:63-65)
 *       v_att [B,N,d*H]     channel k*H + h               (egt_layers.py:139-141)
 *       key_mask [B,N] uint8, 1 = real node               (egt_layers.py:91-94)
 *       h [B,N,Dh], e [B,N,N,De]        (graph_xformer_model_base.py:192-223)
 *     Dense kernels are Keras-layout [in,out].
 *   - return value: EGT_OK or a negative EGT_E_* code; egt_last_error_string()
 *     describes the last failure on the calling thread.  No C++ exception
 *     crosses this boundary.
 *   - kernels are stateless and re-entrant; safe from several host threads on
 *     different streams.
 */
#ifndef EGT_AMD_H_
#define EGT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGT_ABI_VERSION 4

/* error codes */
#define EGT_OK 0
#define EGT_E_NULL (-1)      /* required pointer is NULL                        */
#define EGT_E_SHAPE (-2)     /* bad / unsupported shape (the reference's assert,
                                egt_layers.py:70, maps here)                    */
#define EGT_E_DTYPE (-3)     /* unsupported dtype                               */
#define EGT_E_FLAGS (-4)     /* inconsistent flags (egt_layers.py:20-24)        */
#define EGT_E_HIP (-5)       /* a HIP runtime call failed                       */
#define EGT_E_WORKSPACE (-6) /* workspace too small                             */
#define EGT_E_RCCL (-7)      /* RCCL missing or a collective call failed        */

/* dtype */
#define EGT_F32 0
#define EGT_BF16 1 /* fused block/stack only: the EDGE tensors (e, e', d e', d e and the stack's
                      saved e_l) are bfloat16 in HBM; node tensors, parameters, gradients of
                      parameters and all arithmetic stay fp32 */

/* egt_attn_desc.flags — the operator attributes of EGT.__init__
 * (egt_layers.py:5-16) */
#define EGT_F_EDGE_INPUT 0x001u    /* edge_input: E is added to the logits      */
#define EGT_F_GATE_INPUT 0x002u    /* gate_input: call_gated, else call_ungated */
#define EGT_F_ATTN_MASK 0x004u     /* attn_mask: M present                      */
#define EGT_F_SCALE_DEGREE 0x008u  /* scale_degree                              */
#define EGT_F_SCALER_LINEAR 0x010u /* scaler_type == 'linear' (else 'log')      */
#define EGT_F_TRAINING 0x020u      /* training: random mask / dropout active    */
#define EGT_F_CLIP 0x040u          /* clip_logits_value is not None             */

typedef struct egt_attn_desc {
  int32_t B, N, H, d;        /* graphs, padded nodes, heads, per-head dot dim   */
  int32_t dtype;             /* EGT_F32                                         */
  uint32_t flags;            /* EGT_F_*                                         */
  float clip_lo, clip_hi;    /* clip_logits_value                               */
  float random_mask_prob;    /* egt_layers.py:103                               */
  float attn_dropout;        /* egt_layers.py:116                               */
  int32_t num_virtual_nodes; /* egt_layers.py:131                               */
  int32_t reserved;          /* 0, or EGT_ATTN_WS_* bits (MFMA path only)       */
  uint64_t seed;             /* counter-hash seed for the in-kernel mask RNG    */
} egt_attn_desc;
/* egt_attn_desc.reserved, MFMA path: the caller gives egt_attn_mfma_fwd a workspace of
 * egt_attn_mfma_workspace_bytes() and hands THE SAME, untouched workspace to egt_attn_mfma_bwd:
 * the forward then packs the q/k/v operand copies of both directions once and the backward packs
 * only dV_att.  Without the bit each call packs what it needs into its own workspace. */
#define EGT_ATTN_WS_SHARED 0x1

const char* egt_last_error_string(void);
int egt_abi_version(void);

/* ---- inner op: EGT.call_gated / call_ungated ---------------------------------
 * Replaces lib/models/egt_layers.py:57-143 (gated) and :145-213 (ungated):
 *   (V_att, H_hat, A_tild) = EGT([QKV, E?, G?, M?], mask)
 * key_mask may be NULL (mask=None).  rand_mask / drop_keep ([B,N,N,H] uint8;
 * rand_mask 1 = key masked, drop_keep 1 = kept) are optional INJECTED samples of
 * the two stochastic ops (egt_layers.py:103-108,116-117) used by parity tests;
 * when NULL and (flags & TRAINING) the kernel draws them from the counter hash
 * on (seed,b,l,m,h) (egt_mask_sample exposes the same stream).
 * a_tild may be NULL (only lib/models/analysis.py reads it).
 * rowstats [B,N,H,4] fp32 (softmax max, softmax sum, gate degree, reserved) is
 * written for egt_attn_bwd. */
int egt_attn_fwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                 const uint8_t* rand_mask, const uint8_t* drop_keep, void* v_att,
                 void* h_hat, void* a_tild, void* rowstats, void* stream);

/* Backward of the above — what TF autodiff derives for egt_layers.py:57-143
 * (triggered by model.fit, lib/training/training_base.py:294).
 * d_h_ext (grad w.r.t. output 2, H_hat) may be NULL.  d_E / d_G may be NULL
 * when the corresponding input is absent.  v_att and rowstats are the forward's
 * outputs.  workspace: egt_attn_bwd_workspace_bytes() bytes. */
size_t egt_attn_bwd_workspace_bytes(const egt_attn_desc* desc);
int egt_attn_bwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                 const void* G, const uint8_t* key_mask, const void* attn_mask,
                 const uint8_t* rand_mask, const uint8_t* drop_keep,
                 const void* v_att, const void* rowstats, const void* d_v_att,
                 const void* d_h_ext, void* d_qkv, void* d_E, void* d_G,
                 void* workspace, void* stream);

/* MFMA-tiled (flash-style) variant of the inner op for the large-head geometry
 * (H = 8, d in {16,32,64}; BASELINE config 5): QK^T and A.V on
 * v_mfma_f32_16x16x4_f32, probabilities never leave registers.  Same outputs and the
 * same rowstats as egt_attn_fwd (so egt_attn_bwd pairs with either); not covered:
 * attention dropout, degree scalers, the A_tild output.  egt_attn_mfma_supported
 * returns 1 when `desc` (and the A_tild request) is covered. */
int egt_attn_mfma_supported(const egt_attn_desc* desc, int need_a_tild);
/* Workspace of the MFMA path: head-major operand copies of Q/K/V/dV_att (+ per-row constants and
 * the dA tiles in the backward).  *_fwd_* is what egt_attn_mfma_fwd needs on its own; the other
 * covers both directions (and is what EGT_ATTN_WS_SHARED asks for). */
size_t egt_attn_mfma_fwd_workspace_bytes(const egt_attn_desc* desc);
size_t egt_attn_mfma_workspace_bytes(const egt_attn_desc* desc);
int egt_attn_mfma_fwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                      const void* G, const uint8_t* key_mask, const void* attn_mask,
                      const uint8_t* rand_mask, void* v_att, void* h_hat, void* rowstats,
                      void* workspace, void* stream);
/* Backward on MFMA tiles (launches: pack, dK/dV/dE/dG per 32-key block, dQ per query
 * block).  rowstats is read and its 4th slot written; workspace: egt_attn_mfma_workspace_bytes. */
int egt_attn_mfma_bwd(const egt_attn_desc* desc, const void* qkv, const void* E,
                      const void* G, const uint8_t* key_mask, const void* attn_mask,
                      const uint8_t* rand_mask, const void* v_att, void* rowstats,
                      const void* d_v_att, const void* d_h_ext, void* d_qkv, void* d_E,
                      void* d_G, void* workspace, void* stream);

/* The in-kernel sample streams, materialised ([B,N,N,H] uint8), for bit-exact
 * checks against oracle/rng_ref.py.  which: 0 = random mask (1 = masked),
 * 1 = dropout keep (1 = kept). */
int egt_mask_sample(int which, uint64_t seed, float prob, int32_t B, int32_t N,
                    int32_t H, uint8_t* out, void* stream);

/* EGT_BF_SEED_DEVICE companion (see egt_block_desc): words[i] += increment (mod 2^64) for i < count, as ONE launch
 * on `stream` — capturable, so a hipGraph that contains it in front of the forward draws a fresh random mask on
 * every replay.  The torch host advances its words by 0xD1B54A32D192ED03 per call (EGT.next_seed's step). */
int egt_seed_advance(uint64_t* words, int32_t count, uint64_t increment, void* stream);

/* ---- edge-channel projections around the inner op ----------------------------
 * rows = B*N*N edge rows of width De; H must be 8. */
#define EGT_EP_LAYERNORM 0x1u /* norm_edge before the projections (residual /
                                 constrained); absent for 'bias'               */
#define EGT_EP_GATES 0x2u     /* attention_gates present (gate_attention)       */
#define EGT_ACT_NONE 0
#define EGT_ACT_LRELU 1 /* 'lreluN': alpha = N/10 (graph_xformer_model_base.py:150-156) */
#define EGT_ACT_RELU 2
#define EGT_ACT_ELU 3

typedef struct egt_edge_desc {
  int64_t rows;     /* B*N*N                                                   */
  int32_t De, H;    /* edge_width, num_heads (8)                               */
  int32_t dtype;    /* EGT_F32                                                 */
  uint32_t flags;   /* EGT_EP_*                                                */
  int32_t act;      /* EGT_ACT_* for dense_edge_b                              */
  float act_alpha;  /* leaky-relu slope                                        */
  float ln_eps;     /* Keras LayerNormalization default 1e-3                   */
  int32_t reserved;
} egt_edge_desc;

/* G = LN(e)·Wg + bg ; E = act(LN(e)·We + be)
 * Replaces graph_xformer_model_base.py:195 (norm_edge), :201-204
 * (attention_gates), :149-162 (edge_channel_contrib / dense_edge_b).
 * Wg/We [De,H], bg/be [H]; G_out may be NULL without EGT_EP_GATES. */
int egt_edge_proj_fwd(const egt_edge_desc* desc, const void* e,
                      const void* ln_gamma, const void* ln_beta, const void* Wg,
                      const void* bg, const void* We, const void* be,
                      void* G_out, void* E_out, void* stream);

/* Backward: d_e is WRITTEN (not accumulated).  E_out = forward output (only
 * read when act != NONE).  Parameter grads are written.  workspace:
 * egt_edge_proj_bwd_workspace_bytes(). */
size_t egt_edge_proj_bwd_workspace_bytes(const egt_edge_desc* desc);
int egt_edge_proj_bwd(const egt_edge_desc* desc, const void* e,
                      const void* ln_gamma, const void* ln_beta, const void* Wg,
                      const void* We, const void* E_out, const void* d_G,
                      const void* d_E, void* d_e, void* d_ln_gamma,
                      void* d_ln_beta, void* d_Wg, void* d_bg, void* d_We,
                      void* d_be, void* workspace, void* stream);
/* Same, with d_e = d_e_base + (projection backward): d_e_base [rows,De] is the gradient e already
 * carries from the residual branch e' = e + ... (graph_xformer_model_base.py:218), so the sum the
 * autodiff of the reference forms with a separate add costs no extra pass.  d_e_base may be NULL
 * (then identical to egt_edge_proj_bwd) and may alias d_e. */
int egt_edge_proj_bwd_acc(const egt_edge_desc* desc, const void* e,
                          const void* ln_gamma, const void* ln_beta, const void* Wg,
                          const void* We, const void* E_out, const void* d_G,
                          const void* d_E, const void* d_e_base, void* d_e,
                          void* d_ln_gamma, void* d_ln_beta, void* d_Wg, void* d_bg,
                          void* d_We, void* d_be, void* workspace, void* stream);

/* e' = e + H_hat·Wr + br   (dense_edge_r + res_edge,
 * graph_xformer_model_base.py:214-218).  Wr [H,De], br [De]. */
int egt_edge_update_fwd(const egt_edge_desc* desc, const void* e,
                        const void* h_hat, const void* Wr, const void* br,
                        void* e_out, void* stream);

/* Backward: d_h_hat = d_e_out·Wrᵀ (written); d_Wr, d_br written.  The residual
 * branch (d_e += d_e_out) is the caller's add. */
size_t egt_edge_update_bwd_workspace_bytes(const egt_edge_desc* desc);
int egt_edge_update_bwd(const egt_edge_desc* desc, const void* d_e_out,
                        const void* h_hat, const void* Wr, void* d_h_hat,
                        void* d_Wr, void* d_br, void* workspace, void* stream);

/* ---- outer op: the fused attention block (the data-parallel hot path) ---------
 *   (h', e') = edge_update_residual(h, e) with mha_block inside, pre-norm
 * Replaces graph_xformer_model_base.py:192-223 + :106-145 for
 * edge_channel_type 'residual' / 'constrained' (add_n_norm=False, no dropout,
 * scale_degree=False): norm_edge -> attention_gates / dense_edge_b -> norm_mha ->
 * dense_qkv -> EGT -> dense_mha + res_mha ; dense_edge_r + res_edge.
 * One launch streams e once: LN, the two De->H projections, QK^T, clip, +E,
 * masks, softmax x sigmoid gate, A.V and e' = e + H_hat.Wr + br never leave the
 * CU (E, G, H_hat, A_tild are not materialised).  Everything else (other
 * variants, dropout, degree scalers, d > 8) composes the kernels above.
 * Parameter pointers carry the reference's Keras layer names. */
#define EGT_BF_GATE 0x1u      /* gate_attention                                  */
#define EGT_BF_ATTN_MASK 0x2u /* 'constrained': attn_mask [B,N,N,H] fp32 present */
#define EGT_BF_TRAINING 0x4u  /* random attention mask active                    */
#define EGT_BF_CLIP 0x8u      /* clip_logits_value is not None                   */
#define EGT_BF_NO_EDGE_LN 0x10u /* 'bias' edge channels (EGT-simple, :173-190): gates / edge bias from
                                  the RAW e.  The caller passes norm_edge gamma = 1, beta = 0 and
                                  dense_edge_r kernel = bias = 0 (then e' = e); their gradient
                                  outputs are scratch */
#define EGT_BF_SEED_DEVICE 0x20u /* hipGraph-safe random mask: the kernels draw from the stream seeded with
                                  desc->seed ^ *desc->seed_device (a uint64 in DEVICE memory, read by the
                                  kernel when it runs, not by the host at launch).  A captured step replays
                                  with a fresh sample by advancing that word on the device; forward and
                                  backward of one step must see the same value */

typedef struct egt_block_desc {
  int32_t B, N, H, d, De;   /* model_width Dh = d*H                             */
  int32_t dtype;            /* EGT_F32                                          */
  uint32_t flags;           /* EGT_BF_*                                         */
  float clip_lo, clip_hi;
  float random_mask_prob;
  float ln_eps;             /* 1e-3                                             */
  int32_t reserved;
  uint64_t seed;
  const void* seed_device;  /* EGT_BF_SEED_DEVICE: const uint64_t* on the device, else ignored (NULL) */
} egt_block_desc;

typedef struct egt_block_params {
  const void* norm_edge_gamma;        /* [De]      norm_edge_XX/gamma            */
  const void* norm_edge_beta;         /* [De]                                    */
  const void* attention_gates_kernel; /* [De,H]    attention_gates_XX            */
  const void* attention_gates_bias;   /* [H]                                     */
  const void* dense_edge_b_kernel;    /* [De,H]    dense_edge_b_XX               */
  const void* dense_edge_b_bias;      /* [H]                                     */
  const void* norm_mha_gamma;         /* [Dh]      norm_mha_XX                   */
  const void* norm_mha_beta;          /* [Dh]                                    */
  const void* dense_qkv_kernel;       /* [Dh,3Dh]  dense_qkv_XX                  */
  const void* dense_qkv_bias;         /* [3Dh]                                   */
  const void* dense_mha_kernel;       /* [Dh,Dh]   dense_mha_XX                  */
  const void* dense_mha_bias;         /* [Dh]                                    */
  const void* dense_edge_r_kernel;    /* [H,De]    dense_edge_r_XX               */
  const void* dense_edge_r_bias;      /* [De]                                    */
} egt_block_params;

/* 1 when the fused kernels cover `desc`, else 0 (caller composes instead). */
int egt_block_supported(const egt_block_desc* desc);
/* Name of the backward pair-kernel family the dispatch takes for `desc` when no mask tensor is passed (
 * "k_block_bwd_v5", "k_block_bwd_v4", "k_block_bwd_v4r", "k_narrow_bwd": DESIGN.md section 4); NULL when `desc` is not covered.
 * Static string; for tests and bench lines (the launch profiler reports every family as "k_block_bwd"). */
const char* egt_block_bwd_kernel(const egt_block_desc* desc);
/* bytes of the forward->backward buffer (V_att, softmax row statistics, packed
 * Q/K/V, the LN-folded edge weights and the MFMA-fragment-major copies of Wqkv / Wo the forward prepares: the backward
 * must be given the `saved` buffer of ITS forward call, made with the same parameter values) and of the scratch
 * workspace (max of forward and backward needs). */
size_t egt_block_saved_bytes(const egt_block_desc* desc);
size_t egt_block_workspace_bytes(const egt_block_desc* desc);

/* rand_mask: optional injected sample ([B,N,N,H] uint8, 1 = masked) as in
 * egt_attn_fwd; NULL => in-kernel counter hash when TRAINING and prob > 0. */
int egt_block_fwd(const egt_block_desc* desc, const egt_block_params* params,
                  const void* h, const void* e, const uint8_t* key_mask,
                  const void* attn_mask, const uint8_t* rand_mask, void* h_out,
                  void* e_out, void* saved, void* workspace, void* stream);

/* Backward.  h, e are the block's INPUTS (nothing else of size [B,N,N,*] is
 * kept: LN, projections, logits and softmax are recomputed from e).  d_e may
 * alias d_e_out; d_h must NOT alias d_h_out (the deferred dense_mha weight
 * gradient reads d_h_out after d_h has been written).  Every pointer of `grads`
 * (same layout as the params, but writable) is written. */
int egt_block_bwd(const egt_block_desc* desc, const egt_block_params* params,
                  const void* h, const void* e, const uint8_t* key_mask,
                  const void* attn_mask, const uint8_t* rand_mask,
                  const void* saved, const void* d_h_out, const void* d_e_out,
                  void* d_h, void* d_e, const egt_block_params* grads,
                  void* workspace, void* stream);

/* ---- the model_height loop over attention blocks -------------------------------
 * Replaces the `for ii in range(model_height): h, e = edge_update(tag, h, e)` half
 * of graph_xformer_model_base.py:336-339 when the blocks are applied back to back
 * (the measurement configuration; the reference interleaves ffn_block, for which
 * egt_block_fwd/bwd is the per-layer drop-in).  params / grads are arrays of
 * `layers` structs.  saved keeps the layer activations h_l, e_l (l = 1..layers-1)
 * and each layer's egt_block_saved buffer.  Forward: one launch per layer where the
 * node-side epilogue covers the geometry (Dh = 64), else three.  Backward: d_e carries
 * the edge gradient down the stack in place, every layer's dh and dQKV rows are kept
 * in the workspace, and the GEMM-shaped weight gradients plus all partial sums of ALL
 * layers are finished by three launches at the end (d_h must not alias d_h_out; at
 * most 64 layers per call).  Layer l draws its random attention mask from
 * the counter hash seeded with desc->seed ^ 0x9E3779B97F4A7C15*(l+1). */
size_t egt_stack_saved_bytes(const egt_block_desc* desc, int32_t layers);
size_t egt_stack_workspace_bytes(const egt_block_desc* desc, int32_t layers);
int egt_stack_fwd(const egt_block_desc* desc, int32_t layers,
                  const egt_block_params* params, const void* h, const void* e,
                  const uint8_t* key_mask, const void* attn_mask, void* h_out,
                  void* e_out, void* saved, void* workspace, void* stream);
int egt_stack_bwd(const egt_block_desc* desc, int32_t layers,
                  const egt_block_params* params, const void* h, const void* e,
                  const uint8_t* key_mask, const void* attn_mask, const void* saved,
                  const void* d_h_out, const void* d_e_out, void* d_h, void* d_e,
                  const egt_block_params* grads, void* workspace, void* stream);

/* ---- the fused pair operator at large heads (BASELINE config 5: N = 512, d = 64, De = 32) ------------------
 *   (V_att, e') = norm_edge -> attention_gates / dense_edge_b -> EGT([QKV,E,G],mask) -> dense_edge_r + res_edge
 * Replaces graph_xformer_model_base.py:195-218 (edge_update_residual around mha_block's EGT call, :117-131) with
 * lib/models/egt_layers.py:57-143 inside, for gated 'residual' edge channels: ONE pair kernel per direction streams
 * e (and de' in the backward) once; E, G, H_hat, dE, dG, dH_ext exist only in LDS.  At this head width the
 * node-side Dense layers of mha_block (norm_mha, dense_qkv :109-113, dense_mha :136) are [B N, 512] GEMMs and stay
 * with the caller (library GEMMs): the operator takes QKV [B,N,3 d H] (channel s*dH + k*H + h, egt_layers.py:73-76)
 * and returns V_att [B,N,d H].  egt_block_desc / egt_block_params are reused (node-side pointers are ignored);
 * desc->reserved & EGT_ATTN_WS_SHARED: the caller hands the forward's untouched workspace to the backward, called
 * with the same parameter values (the q / k / v operand copies and the LN-folded weight table are then made once).  In-kernel random mask as in egt_block_fwd. */
int egt_pair_supported(const egt_block_desc* desc);
size_t egt_pair_workspace_bytes(const egt_block_desc* desc);
int egt_pair_fwd(const egt_block_desc* desc, const egt_block_params* params, const void* qkv,
                 const void* e, const uint8_t* key_mask, void* v_att, void* e_out, void* rowstats,
                 void* workspace, void* stream);
/* rowstats [B,N,H,4] is the forward's output (read; slot 3 written).  d_e may alias d_e_out.  Every edge-side
 * pointer of `grads` is written (norm_edge_*, attention_gates_*, dense_edge_b_*, dense_edge_r_*). */
int egt_pair_bwd(const egt_block_desc* desc, const egt_block_params* params, const void* qkv,
                 const void* e, const uint8_t* key_mask, const void* v_att, void* rowstats,
                 const void* d_v_att, const void* d_e_out, void* d_qkv, void* d_e,
                 const egt_block_params* grads, void* workspace, void* stream);

/* ---- channel FFN (the step after the attention block in every layer) ----
 * Replaces ffnlr1 / ffnact / ffnlr2 of lib/models/graph_xformer_model_base.py:230-258 as
 * ffn_block applies them (:309-324; pre-norm, no cross-talk, ffn_multiplier = 2):
 *   y = x + Dense_2( act( Dense_1( LayerNorm(x) ) ) )
 * on `rows` rows of `width` channels: the edge channels [B*N*N, De] or the node channels
 * [B*N, Dh].  Keras layouts: kernels are [in,out]; LayerNormalization epsilon = ln_eps.
 * Fused MFMA kernels for width in {16,32,48,64}, fp32, activation EGT_ACT_ELU (config default) or
 * EGT_ACT_RELU; egt_ffn_supported says whether a desc is covered. */
/* how the matrix products are evaluated (tensors stay fp32 in HBM, accumulation is always fp32):
 *   EGT_MM_F32     exact fp32 products (v_mfma_f32_16x16x4_f32)
 *   EGT_MM_BF16X3  every operand split into two bfloat16 terms, a.b = a_hi.b_hi + a_lo.b_hi + a_hi.b_lo on the
 *                  bf16 matrix pipe: per-product error <= 2^-16 (fp32: 2^-24); stays inside the fp32 parity
 *                  tolerances of the test-suite at 3/16 of the fp32 MFMA cost
 *   EGT_MM_BF16    plain bfloat16 products (hi terms only): tolerance rtol 2e-2 (SURVEY 8(c) bf16 figure) */
#define EGT_MM_F32 0
#define EGT_MM_BF16X3 1
#define EGT_MM_BF16 2

typedef struct egt_ffn_desc {
  int64_t rows;
  int32_t width;
  int32_t dtype;      /* EGT_F32 */
  int32_t activation; /* EGT_ACT_* (config.activation) */
  float ln_eps;       /* 1e-3 */
  int32_t matmul;     /* EGT_MM_* */
  int32_t flags;      /* EGT_FFN_* (0 for plain calls) */
} egt_ffn_desc;
/* egt_ffn_bwd only: `workspace` is the buffer an egt_ffn_fwd call with the same rows / width / activation / matmul and
 * the same parameter VALUES has used on this stream or an earlier-ordered one; the prepared operands in it
 * (LayerNorm-folded weight slabs) are reused and the backward skips its preparation launch. */
#define EGT_FFN_WS_PREPARED 1

typedef struct egt_ffn_params {
  const void* norm_gamma;  /* [W]     norm_fnn_{node,edge}_XX */
  const void* norm_beta;   /* [W]                              */
  const void* lr1_kernel;  /* [W,2W]  fnn_lr1_XX               */
  const void* lr1_bias;    /* [2W]                             */
  const void* lr2_kernel;  /* [2W,W]  fnn_lr2_XX               */
  const void* lr2_bias;    /* [W]                              */
} egt_ffn_params;

int egt_ffn_supported(const egt_ffn_desc* desc);
size_t egt_ffn_workspace_bytes(const egt_ffn_desc* desc);
int egt_ffn_fwd(const egt_ffn_desc* desc, const egt_ffn_params* params, const void* x, void* y,
                void* workspace, void* stream);
/* x is the layer INPUT (the hidden activations are recomputed); dx may alias dy; every pointer
 * of `grads` (same layout as the params, writable) is written. */
int egt_ffn_bwd(const egt_ffn_desc* desc, const egt_ffn_params* params, const void* x,
                const void* dy, void* dx, const egt_ffn_params* grads, void* workspace,
                void* stream);

/* ---- mask producers (integer / boolean work: bit-exact) ---------
 * The masks the path consumes are produced by the model around it; these entry points replace
 *   Neg1MaskedEmbedding.compute_mask   lib/base/xformer_layers/masking.py:35-43   (x + 1) != 0
 *   keras.layers.Masking(mask_value)   lib/models/cifar10/dc.py:69               any(x != v, -1)
 *   VirtualNodeEmbedding.compute_mask  lib/base/graph_layers/virtual_nodes.py:47-50
 *   AdjMatModel.get_edge_mask          lib/models/graph_model_base.py:131-142     tile(adj, H)
 *   VNModel.get_edge_mask              lib/models/graph_model_base.py:248-268     ones for VN rows/cols
 * features [B,N] int32 (padding value -1)  ->  mask [B, num_virtual_nodes + N] uint8 (1 = real node) */
int egt_node_mask_from_features(const int32_t* features, int32_t B, int32_t N, int32_t num_virtual_nodes,
                                uint8_t* mask, void* stream);
/* features [B,N,width] fp32 -> mask [B, num_virtual_nodes + N] uint8: some feature != mask_value */
int egt_node_mask_from_float_features(const float* features, int32_t B, int32_t N, int32_t width,
                                      float mask_value, int32_t num_virtual_nodes, uint8_t* mask,
                                      void* stream);
/* adj [B,N,N] fp32 -> M [B, nv+N, nv+N, H] fp32 (nv = num_virtual_nodes): the attention mask of
 * edge_channel_type 'constrained' as the inner op / fused block take it */
int egt_constrained_edge_mask(const float* adj, int32_t B, int32_t N, int32_t H, int32_t num_virtual_nodes,
                              float* M, void* stream);

/* ---- edge-channel input embedding (the model code that PRODUCES e) -----------
 *   e0 = Neg1MaskedEmbedding(num_edge_features + 1, De)(feature_matrix)        lib/models/zinc/dc.py:70-73
 *      + Dense(upto_hop -> De)(stack_hops(graph_matrix))                        lib/models/graph_model_base.py:97-127
 * feature_matrix [B,N,N] int32 (-1 = no edge / padding), graph_matrix [B,N,N] fp32, fm_table [V,De]
 * (V = num_edge_features + 1), adj_kernel [upto_hop, De] (Keras layout), adj_bias [De].  `hops`
 * ([upto_hop + num_float_features,B,N,N] fp32 -- plane-major --, egt_edge_embed_hops_bytes) receives the hop matrices; the backward
 * reads them again.  Gradients: d_fm_table, d_adj_kernel, d_adj_bias (no gradient flows to the inputs). */
typedef struct egt_embed_desc {
  int32_t B, N, De;
  int32_t upto_hop;           /* 1..16 */
  int32_t clip_hops;          /* clip every hop product to [0,1] (graph_model_base.py:114-115) */
  int32_t num_edge_features;  /* embedding rows - 1 (0: no integer feature matrix; pass a zero one-row table) */
  int32_t dtype;              /* EGT_F32 */
  int32_t num_float_features; /* 0..4 real-valued edge features per pair: keras Masking(mask_value) + Dense
                                 (lib/models/cifar10/dc.py:70-73); the rows of that Dense kernel are appended to
                                 adj_kernel ([upto_hop + num_float_features, De]) and its bias added to adj_bias */
  float mask_value;           /* Masking: a pair whose features all equal mask_value contributes 0 */
  int32_t reserved;
} egt_embed_desc;
int egt_edge_embed_supported(const egt_embed_desc* desc);
size_t egt_edge_embed_hops_bytes(const egt_embed_desc* desc);
size_t egt_edge_embed_workspace_bytes(const egt_embed_desc* desc);
int egt_edge_embed_fwd(const egt_embed_desc* desc, const int32_t* feature_matrix, const void* graph_matrix,
                       const void* float_features /* [B,N,N,num_float_features] or NULL */, const void* fm_table,
                       const void* adj_kernel, const void* adj_bias, void* hops, void* e_out, void* stream);
int egt_edge_embed_bwd(const egt_embed_desc* desc, const int32_t* feature_matrix, const void* hops,
                       const void* d_e, void* d_fm_table, void* d_adj_kernel, void* d_adj_bias,
                       void* workspace, void* stream);

/* ---- batch data parallelism: the gradient all-reduce on RCCL -------------------
 * Replaces what tf.distribute.MirroredStrategy does for the reference
 * (lib/training/training_base.py:230-247): one synchronous all-reduce of every
 * parameter gradient per step.  One process per GPU and ONE communicator per process
 * (the library's only global state).  Bootstrap: rank 0 calls egt_dp_unique_id and
 * hands the 128 bytes to every rank out of band (the launcher's TCP store); every
 * rank then calls egt_dp_init with its HIP device current.  egt_dp_allreduce reduces
 * `count` fp32 values in place on `stream` (the stream the backward ran on; no host
 * synchronisation) -- SUM, or the mean over ranks when average != 0.  RCCL is resolved
 * at run time (the copy already mapped into the process, else librccl.so.1); its
 * absence is EGT_E_RCCL, not a load failure of this library. */
#define EGT_DP_ID_BYTES 128
int egt_dp_unique_id(void* id_out /* host, EGT_DP_ID_BYTES */);
int egt_dp_init(const void* id /* host, EGT_DP_ID_BYTES */, int32_t world, int32_t rank);
int egt_dp_allreduce(void* buf, size_t count, int32_t average, void* stream);
int egt_dp_world(void); /* 0 before init */
int egt_dp_rank(void);  /* -1 before init */
int egt_dp_finalize(void);

/* ---- per-kernel timing (measurement only) ------------------------------------
 * egt_prof_enable(1) makes every launch site bracket its kernel with hipEvents
 * on the launch stream (2 = reset counters and enable, 0 = off).  After the
 * caller has synchronised, egt_prof_read(name, ...) returns the launch count
 * and summed milliseconds of kernel `name`; egt_prof_names lists the names. */
int egt_prof_enable(int on);
int egt_prof_filter(const char* kernel_name); /* time only this kernel (NULL/"" = all) */
int egt_prof_stride(int every);               /* time every `every`-th launch of each timed kernel (default 1 = all): an event pair
                                                * costs the stream a few microseconds, a sample keeps a throughput measurement honest */
int egt_prof_read(const char* name, int64_t* count, double* total_ms);
/* Launches captured into a hipGraph while the profile is enabled carry their hipEvents as event-record nodes of
 * that graph; every replay re-records them.  After a replay has completed, egt_prof_collect_graph() adds its elapsed times to
 * the kernels' counts / sums (reset_counts != 0: zero them first) and returns the number of event pairs read. */
int egt_prof_collect_graph(int reset_counts);
int egt_prof_forget_graphs(void);             /* drop the event pairs of captured graphs from the profile (the graphs may go on replaying) */
int egt_prof_names(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* EGT_AMD_H_ */
