// Edge-channel input embedding of the EGT model -- the producer of the e tensor the
// attention path consumes:   e0 = Embedding(fmat + 1) + Dense(stack_hops(adj))
//   stack_hops   lib/models/graph_model_base.py:101-119  hops[...,0] = A, hops[...,k] = clip(A . hops[...,k-1], 0, 1)
//   adj_emb      :125-126                                Dense(upto_hop -> De)
//   fm_emb       lib/models/zinc/dc.py:70-73             Neg1MaskedEmbedding(num_edge_features + 1, De)
//   edge_emb_add lib/models/graph_xformer_model_base.py:401-404
// Once per batch (not per layer).  The hop products are batched N x N x N fp32 contractions on MFMA
// tiles (one wave per 16 x 16 output tile, operands straight from L2: 1 GFLOP per batch at config 2);
// the embedding itself is a write-bound streaming kernel (256 B per pair at De = 64); its backward is a
// read-bound contraction over the pair axis with deterministic per-workgroup partials.
#include "egt_common.h"

typedef float v4f_e __attribute__((ext_vector_type(4)));

// hop-major storage hops[k][b][l][m] (every plane is a contiguous [B,N,N] matrix stack: the B operand of the next
// product and the embedding kernels read it with unit stride).
// hops[k] = clip( adj . hops[k-1] ) per graph; grid = B * T2 * T2 waves (T2 = ceil(N/32)), one 32 x 32 output block = 2 x 2
// MFMA tiles per wave: every A / B fragment feeds two products, half the L2 -> L1 operand traffic of one tile per wave
// (N = 150: 57 us per hop with 16 x 16 tiles and scalar loads -> 42 us with 16-byte A loads -> this form).
__global__ void __launch_bounds__(64) k_hop_step(const float* __restrict__ adj, float* __restrict__ hops, int B, int N,
                                                 int K, int k, int clip) {
  const int T2 = (N + 31) / 32;
  const int tile = blockIdx.x % (T2 * T2), b = blockIdx.x / (T2 * T2);
  const int l0 = (tile / T2) * 32, m0 = (tile % T2) * 32;
  const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
  const size_t plane = (size_t)B * N * N;
  const float* A = adj + (size_t)b * N * N;
  const float* P = hops + (size_t)(k - 1) * plane + (size_t)b * N * N;
  v4f_e acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = (v4f_e){0.f, 0.f, 0.f, 0.f};
  const float* Ar[2] = {A + (size_t)min(l0 + i, N - 1) * N, A + (size_t)min(l0 + 16 + i, N - 1) * N};
  const int mb[2] = {min(m0 + i, N - 1), min(m0 + 16 + i, N - 1)};
  // 16 contraction steps per iteration: lane (i, kk) takes A[row][j0 + 4 kk .. + 3] as ONE 16-byte load (contraction index
  // j = j0 + 4 kk + s in MFMA step s, the same on the B side).  (Exact for 0/1 adjacencies in any summation order.)
  for (int j0 = 0; j0 < N; j0 += 16) {
    const int jb = j0 + 4 * kk;
    float av[2][4], bv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (jb + 3 < N) {
        const float4 t = *reinterpret_cast<const float4*>(Ar[u] + jb);   // (4-byte alignment is enough for global loads)
        av[u][0] = t.x; av[u][1] = t.y; av[u][2] = t.z; av[u][3] = t.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) av[u][s] = jb + s < N ? Ar[u][jb + s] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[u][s] = jb + s < N ? P[(size_t)(jb + s) * N + mb[u]] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bv[v][s], acc[u][v], 0, 0, 0);
  }
  // D: row 4 * (lane >> 4) + r, column lane & 15
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = l0 + 16 * u + 4 * kk + r, m = m0 + 16 * v + i;
        float x = acc[u][v][r];
        if (clip) x = fminf(fmaxf(x, 0.f), 1.f);
        if (l < N && m < N) hops[(size_t)k * plane + ((size_t)b * N + l) * N + m] = x;
      }
}

// ALL hop planes in one launch.  hops[k][:, c] depends only on A and hops[k-1][:, c]: a workgroup that owns a block of
// columns of one graph keeps the whole adjacency (zero padded [R16][R16 + 4]) and its column block of the current plane
// ([R16][16 CT + 4]) in LDS and walks k = 1 .. KH-1 without another global read: per hop every wave contracts its (row tile,
// column tile) pairs over all of A's columns (A: one ds_read_b128 = 4 contraction steps, B: 4 ds_read_b32; the pitches make
// both conflict-free), then -- barrier -- clips, stores the plane to HBM and puts it back into LDS as the next B operand.
// (One launch per hop, operands from L2, 3 waves per SIMD in flight: 36 us per hop at N = 150, 15 launches per batch.)
#define HOP_NW 8     // waves per workgroup
#define HOP_MAXT 8   // (row tile, column tile) pairs per wave
__global__ void __launch_bounds__(HOP_NW * 64) k_hop_chain(const float* __restrict__ adj, float* __restrict__ hops, int B, int N,
                                                           int KH, int clip, int NB, int CT) {
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  const int RT = (N + 15) / 16, R16 = RT * 16, PA = R16 + 4, PH = 16 * CT + 4;
  float* As = hsm;
  float* Hs = As + R16 * PA;
  const int b = blockIdx.x / NB, c0 = (blockIdx.x % NB) * CT * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
  const size_t plane = (size_t)B * N * N;
  const float* A = adj + (size_t)b * N * N;
  float* H0 = hops + (size_t)b * N * N;
  for (int i = threadIdx.x; i < R16 * PA; i += HOP_NW * 64) {
    const int r = i / PA, c = i % PA;
    As[i] = (r < N && c < N) ? A[(size_t)r * N + c] : 0.f;
  }
  for (int i = threadIdx.x; i < R16 * PH; i += HOP_NW * 64) {
    const int r = i / PH, c = c0 + i % PH;
    const bool in = r < N && c < N && (i % PH) < 16 * CT;
    const float v = in ? A[(size_t)r * N + c] : 0.f;
    Hs[i] = v;
    if (in) H0[(size_t)r * N + c] = v;   // plane 0
  }
  __syncthreads();
  const int ntile = RT * CT;
  for (int k = 1; k < KH; ++k) {
    v4f_e acc[HOP_MAXT];
#pragma unroll
    for (int ti = 0; ti < HOP_MAXT; ++ti) {
      acc[ti] = (v4f_e){0.f, 0.f, 0.f, 0.f};
      const int t = wave + HOP_NW * ti;
      if (t < ntile) {
        const float* ap = As + (16 * (t / CT) + m) * PA + 4 * kq;
        const float* bp = Hs + (4 * kq) * PH + 16 * (t % CT) + m;
        v4f_e c = acc[ti];
        for (int g = 0; g < RT; ++g) {
          const v4f_e a4 = *reinterpret_cast<const v4f_e*>(ap + 16 * g);
          const float* bg = bp + 16 * g * PH;
          const float b0 = bg[0], b1 = bg[PH], b2 = bg[2 * PH], b3 = bg[3 * PH];
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[0], b0, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[1], b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[2], b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[3], b3, c, 0, 0, 0);
        }
        acc[ti] = c;
      }
    }
    __syncthreads();   // every wave has read the old plane
    float* Hk = H0 + (size_t)k * plane;
#pragma unroll
    for (int ti = 0; ti < HOP_MAXT; ++ti) {
      const int t = wave + HOP_NW * ti;
      if (t < ntile) {
        const int rt = t / CT, ct = t % CT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // D: row 4 kq + r, column m
          float x = acc[ti][r];
          if (clip) x = fminf(fmaxf(x, 0.f), 1.f);
          const int row = 16 * rt + 4 * kq + r, col = c0 + 16 * ct + m;
          Hs[row * PH + 16 * ct + m] = x;
          if (row < N && col < N) Hk[(size_t)row * N + col] = x;
        }
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_hop_first(const float* __restrict__ adj, float* __restrict__ hops, long pairs, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < pairs) hops[i] = adj[i];   // plane 0
}

// real-valued edge features (lib/models/cifar10/dc.py:70-73: keras Masking(mask_value) + Dense(edge_emb)): Masking zeroes
// a pair whose features ALL equal mask_value; the Dense then is F more "planes" behind the hop planes with the rows of its
// kernel appended to adj_emb's -- plane[K + f][pair] = masked ? 0 : x[pair, f]
__global__ void __launch_bounds__(256) k_feature_planes(const float* __restrict__ x, float* __restrict__ planes, long pairs, int F,
                                                       float mask_value) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pairs) return;
  bool keep = false;
  for (int f = 0; f < F; ++f) keep |= x[i * F + f] != mask_value;
  for (int f = 0; f < F; ++f) planes[(size_t)f * pairs + i] = keep ? x[i * F + f] : 0.f;
}

// e0[pair, c] = fm_table[fmat[pair] + 1, c] + bias[c] + sum_k hops[pair, k] * W[k, c]
// thread = (pair, 4 channels); W / bias / table staged in LDS
template <int KMAX>
__global__ void __launch_bounds__(256) k_edge_embed_fwd(const int32_t* __restrict__ fmat, const float* __restrict__ hops,
                                                        const float* __restrict__ table, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ e, long pairs,
                                                        int De, int K, int V) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ws = sm;                 // [K][De]
  float* Ts = Ws + K * De;        // [V][De] (+ bias folded in)
  for (int i = threadIdx.x; i < K * De; i += 256) Ws[i] = W[i];
  for (int i = threadIdx.x; i < V * De; i += 256) Ts[i] = table[i] + bias[i % De];
  __syncthreads();
  const int C4 = De / 4;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long pair = gid / C4;
  const int c4 = (int)(gid % C4);
  if (pair >= pairs) return;
  int f = fmat[pair] + 1;
  f = min(max(f, 0), V - 1);
  float4 acc = *reinterpret_cast<const float4*>(Ts + f * De + 4 * c4);
  const float* hp = hops + pair;
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const float hv = hp[(size_t)k * pairs];
    const float4 w = *reinterpret_cast<const float4*>(Ws + k * De + 4 * c4);
    acc.x = fmaf(hv, w.x, acc.x); acc.y = fmaf(hv, w.y, acc.y);
    acc.z = fmaf(hv, w.z, acc.z); acc.w = fmaf(hv, w.w, acc.w);
  }
  *reinterpret_cast<float4*>(e + pair * De + 4 * c4) = acc;
}

// backward: dW[k,c] = sum_pairs hops[pair,k] de[pair,c];  dtable[v,c] = sum_{fmat+1 == v} de[pair,c];  dbias = sum de
// block = PL pair lanes x C4P channel quads (C4P = De/4 rounded up to a power of two, PL = 256 / C4P: narrow edge
// channels get more pair lanes instead of idle threads); each thread owns 4 channels and walks its pairs; the
// (K + V) x 4 accumulators per thread are reduced over the pair lanes in LDS; one partial per workgroup
#define EMB_PPB 2048   // pairs per workgroup
template <int K_, int V_>
__global__ void __launch_bounds__(256) k_edge_embed_bwd(const int32_t* __restrict__ fmat, const float* __restrict__ hops,
                                                        const float* __restrict__ de, float* __restrict__ part, long pairs,
                                                        int De, int C4P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [PL pair lanes][(K+V)][De]
  const int C4 = De / 4;
  const int PL = 256 / C4P;
  const int c4 = threadIdx.x % C4P, pl = threadIdx.x / C4P;
  float4 aw[K_], at[V_];
#pragma unroll
  for (int k = 0; k < K_; ++k) aw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int v = 0; v < V_; ++v) at[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long p0 = (long)blockIdx.x * EMB_PPB;
  const long p1 = min(pairs, p0 + EMB_PPB);
  if (c4 < C4) {
    for (long pair = p0 + pl; pair < p1; pair += PL) {
      const float4 d = *reinterpret_cast<const float4*>(de + pair * De + 4 * c4);
      const float* hp = hops + pair;
      int f = fmat[pair] + 1;
      f = min(max(f, 0), V_ - 1);
#pragma unroll
      for (int k = 0; k < K_; ++k) {
        const float hv = hp[(size_t)k * pairs];
        aw[k].x = fmaf(hv, d.x, aw[k].x); aw[k].y = fmaf(hv, d.y, aw[k].y);
        aw[k].z = fmaf(hv, d.z, aw[k].z); aw[k].w = fmaf(hv, d.w, aw[k].w);
      }
#pragma unroll
      for (int v = 0; v < V_; ++v) {   // branch-free one-hot accumulate (static register indexing)
        const float s = (f == v) ? 1.0f : 0.0f;
        at[v].x = fmaf(s, d.x, at[v].x); at[v].y = fmaf(s, d.y, at[v].y);
        at[v].z = fmaf(s, d.z, at[v].z); at[v].w = fmaf(s, d.w, at[v].w);
      }
    }
#pragma unroll
    for (int k = 0; k < K_; ++k) *reinterpret_cast<float4*>(sm + ((size_t)pl * (K_ + V_) + k) * De + 4 * c4) = aw[k];
#pragma unroll
    for (int v = 0; v < V_; ++v) *reinterpret_cast<float4*>(sm + ((size_t)pl * (K_ + V_) + K_ + v) * De + 4 * c4) = at[v];
  }
  __syncthreads();
  const int R = (K_ + V_) * De;
  float* out = part + (size_t)blockIdx.x * R;
  for (int i = threadIdx.x; i < R; i += 256) {
    float s = 0.f;
    for (int j = 0; j < PL; ++j) s += sm[(size_t)j * R + i];   // fixed order: bit-reproducible
    out[i] = s;
  }
}

// deterministic reduction of the workgroup partials + split into the three gradients.
// block = 64 outputs x 4 partial groups (group g sums partials j = g, g+4, ... in order; the four group sums
// are added in a fixed order: bit-reproducible); the bias gradient is the column sum of the table gradient
// (every pair selects exactly one table row)
__global__ void __launch_bounds__(256) k_edge_embed_bwd_reduce(const float* __restrict__ part, int nparts, int K, int V, int De,
                                                               float* __restrict__ dW, float* __restrict__ dtable,
                                                               float* __restrict__ dbias) {
  __shared__ float red[4][64];
  const int R = (K + V) * De;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < R)
    for (int j = g; j < nparts; j += 4) s += part[(size_t)j * R + i];
  red[g][o] = s;
  __syncthreads();
  if (g == 0 && i < R) {
    const float v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    if (i < K * De) dW[i] = v; else dtable[i - K * De] = v;
  }
}
__global__ void __launch_bounds__(64) k_edge_embed_bias_grad(const float* __restrict__ dtable, int V, int De, float* __restrict__ dbias) {
  const int c = threadIdx.x;
  if (c >= De) return;
  float b = 0.f;
  for (int v = 0; v < V; ++v) b += dtable[v * De + c];
  dbias[c] = b;
}

static int embed_check(const egt_embed_desc* d) {
  if (!d) EGT_FAIL(EGT_E_NULL, "desc is NULL");
  if (d->B <= 0 || d->N <= 0) EGT_FAIL(EGT_E_SHAPE, "B and N must be positive");
  if (d->De < 4 || d->De > 64 || d->De % 4) EGT_FAIL(EGT_E_SHAPE, "edge embedding covers edge_width in 4..64, multiple of 4 (got %d)", d->De);
  if (d->upto_hop < 1 || d->upto_hop > 16) EGT_FAIL(EGT_E_SHAPE, "upto_hop must be in 1..16 (got %d)", d->upto_hop);
  if (d->num_float_features < 0 || d->num_float_features > 4) EGT_FAIL(EGT_E_SHAPE, "num_float_features must be in 0..4 (got %d)", d->num_float_features);
  if (d->num_edge_features < 0 || d->num_edge_features > 7) EGT_FAIL(EGT_E_SHAPE, "num_edge_features must be in 0..7 (got %d)", d->num_edge_features);
  if (d->dtype != EGT_F32) EGT_FAIL(EGT_E_DTYPE, "edge embedding is fp32");
  return EGT_OK;
}
extern "C" int egt_edge_embed_supported(const egt_embed_desc* d) {
  if (!d) return 0;
  return d->B > 0 && d->N > 0 && d->De >= 4 && d->De <= 64 && d->De % 4 == 0 && d->upto_hop >= 1 && d->upto_hop <= 16 &&
         d->num_float_features >= 0 && d->num_float_features <= 4 &&
         d->num_edge_features >= 0 && d->num_edge_features <= 7 && d->dtype == EGT_F32;
}
extern "C" size_t egt_edge_embed_hops_bytes(const egt_embed_desc* d) {
  if (!egt_edge_embed_supported(d)) return 0;
  return (size_t)d->B * d->N * d->N * (d->upto_hop + d->num_float_features) * sizeof(float);
}
static int embed_nparts(const egt_embed_desc* d) {
  const long pairs = (long)d->B * d->N * d->N;
  return (int)((pairs + EMB_PPB - 1) / EMB_PPB);
}
extern "C" size_t egt_edge_embed_workspace_bytes(const egt_embed_desc* d) {
  if (!egt_edge_embed_supported(d)) return 0;
  return (size_t)embed_nparts(d) * (d->upto_hop + d->num_float_features + d->num_edge_features + 1) * d->De * sizeof(float);
}

extern "C" int egt_edge_embed_fwd(const egt_embed_desc* d, const int32_t* feature_matrix, const void* graph_matrix,
                                  const void* float_features, const void* fm_table, const void* adj_kernel,
                                  const void* adj_bias, void* hops, void* e_out, void* stream) {
  int rc = embed_check(d);
  if (rc) return rc;
  if (!feature_matrix || !graph_matrix || !fm_table || !adj_kernel || !adj_bias || !hops || !e_out)
    EGT_FAIL(EGT_E_NULL, "feature_matrix/graph_matrix/fm_table/adj_kernel/adj_bias/hops/e_out is NULL");
  hipStream_t st = (hipStream_t)stream;
  const long pairs = (long)d->B * d->N * d->N;
  if (d->num_float_features > 0 && !float_features) EGT_FAIL(EGT_E_NULL, "num_float_features set but float_features is NULL");
  const int KH = d->upto_hop, K = KH + d->num_float_features, V = d->num_edge_features + 1, T2 = (d->N + 31) / 32;
  // one launch for all planes when the adjacency and one column block fit in LDS (N <= ~180), else a launch per hop
  int CT = 0;
  {
    static const bool no_chain = getenv("EGT_NO_HOP_CHAIN") != nullptr;
    const int RT = (d->N + 15) / 16, R16 = RT * 16;
    int ct = (int)((long)RT * d->B / 256);   // >= 256 workgroups when the batch allows it
    ct = ct < 1 ? 1 : (ct > RT ? RT : ct);
    while (ct > 1 && RT * ct > HOP_NW * HOP_MAXT) --ct;
    auto lds_of = [&](int c) { return (size_t)(R16 * (R16 + 4) + R16 * (16 * c + 4)) * sizeof(float); };
    while (ct > 1 && lds_of(ct) > 160 * 1024) --ct;
    if (!no_chain && KH > 1 && RT * ct <= HOP_NW * HOP_MAXT && lds_of(ct) <= 160 * 1024) CT = ct;
    if (CT) {
      const int NB = (RT + CT - 1) / CT;
      EGT_MAX_LDS_ONCE(k_hop_chain);
      EGT_LAUNCH("k_hop_chain", k_hop_chain, dim3((unsigned)(d->B * NB)), dim3(HOP_NW * 64), lds_of(CT), st,
                 (const float*)graph_matrix, (float*)hops, d->B, d->N, KH, d->clip_hops ? 1 : 0, NB, CT);
    }
  }
  if (!CT) {
    EGT_LAUNCH("k_hop_first", k_hop_first, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, (const float*)graph_matrix,
               (float*)hops, pairs, K);
    for (int k = 1; k < KH; ++k)
      EGT_LAUNCH("k_hop_step", k_hop_step, dim3((unsigned)(d->B * T2 * T2)), dim3(64), 0, st, (const float*)graph_matrix,
                 (float*)hops, d->B, d->N, K, k, d->clip_hops ? 1 : 0);
  }
  if (d->num_float_features > 0)
    EGT_LAUNCH("k_feature_planes", k_feature_planes, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st,
               (const float*)float_features, (float*)hops + (size_t)KH * pairs, pairs, d->num_float_features, d->mask_value);
  const long threads = pairs * (d->De / 4);
  const size_t lds = (size_t)(K + V) * d->De * sizeof(float);
  EGT_LAUNCH("k_edge_embed_fwd", k_edge_embed_fwd<16>, dim3((unsigned)((threads + 255) / 256)), dim3(256), lds, st,
             feature_matrix, (const float*)hops, (const float*)fm_table, (const float*)adj_kernel, (const float*)adj_bias,
             (float*)e_out, pairs, d->De, K, V);
  EGT_HIP_LAUNCH_CHECK("egt_edge_embed_fwd");
  return EGT_OK;
}

template <int K_>
static void launch_embed_bwd(const egt_embed_desc* d, const int32_t* fmat, const float* hops, const float* de, float* part,
                             hipStream_t st) {
  const long pairs = (long)d->B * d->N * d->N;
  const int nparts = embed_nparts(d);
  int c4p = 1;
  while (c4p < d->De / 4) c4p *= 2;
  while ((size_t)(256 / c4p) * (K_ + 8) * d->De * sizeof(float) > 150 * 1024) c4p *= 2;   // LDS reduction image must fit
#define EB(V_)                                                                                                      \
  do {                                                                                                              \
    const size_t lds = (size_t)(256 / c4p) * (K_ + V_) * d->De * sizeof(float);                                     \
    EGT_MAX_LDS_ONCE(k_edge_embed_bwd<K_, V_>); \
    EGT_LAUNCH("k_edge_embed_bwd", (k_edge_embed_bwd<K_, V_>), dim3(nparts), dim3(256), lds, st, fmat, hops, de, part, pairs, d->De, c4p); \
  } while (0)
  switch (d->num_edge_features + 1) {
    case 1: EB(1); break; case 2: EB(2); break; case 3: EB(3); break; case 4: EB(4); break;
    case 5: EB(5); break; case 6: EB(6); break; case 7: EB(7); break; default: EB(8); break;
  }
#undef EB
}

extern "C" int egt_edge_embed_bwd(const egt_embed_desc* d, const int32_t* feature_matrix, const void* hops,
                                  const void* d_e, void* d_fm_table, void* d_adj_kernel, void* d_adj_bias,
                                  void* workspace, void* stream) {
  int rc = embed_check(d);
  if (rc) return rc;
  if (!feature_matrix || !hops || !d_e || !d_fm_table || !d_adj_kernel || !d_adj_bias || !workspace)
    EGT_FAIL(EGT_E_NULL, "feature_matrix/hops/d_e/d_fm_table/d_adj_kernel/d_adj_bias/workspace is NULL");
  hipStream_t st = (hipStream_t)stream;
  const float* h = (const float*)hops;
  const float* de = (const float*)d_e;
  float* part = (float*)workspace;
  const int KT = d->upto_hop + d->num_float_features;   // hop planes + real-valued feature planes
  switch (KT) {
#define C(K_) case K_: launch_embed_bwd<K_>(d, feature_matrix, h, de, part, st); break;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
#undef C
  }
  const int R = (KT + d->num_edge_features + 1) * d->De;
  EGT_LAUNCH("k_edge_embed_bwd_reduce", k_edge_embed_bwd_reduce, dim3((R + 63) / 64), dim3(256), 0, st, (const float*)part,
             embed_nparts(d), KT, d->num_edge_features + 1, d->De, (float*)d_adj_kernel, (float*)d_fm_table,
             (float*)d_adj_bias);
  EGT_LAUNCH("k_edge_embed_bwd_reduce", k_edge_embed_bias_grad, dim3(1), dim3(64), 0, st, (const float*)d_fm_table,
             d->num_edge_features + 1, d->De, (float*)d_adj_bias);
  EGT_HIP_LAUNCH_CHECK("egt_edge_embed_bwd");
  return EGT_OK;
}
