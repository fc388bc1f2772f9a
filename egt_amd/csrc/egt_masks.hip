// Mask producers of the EGT attention path -- integer / boolean work, bit-exact.
//   node mask   [B,N] uint8 : Neg1MaskedEmbedding.compute_mask  (lib/base/xformer_layers/masking.py:35-43:
//                             Embedding(mask_zero=True) on inputs+1, i.e. (x + 1) != 0  <=>  x != -1)
//                             keras.layers.Masking(mask_value)   (lib/models/cifar10/dc.py:69:
//                             any(x != mask_value, axis=-1))
//                             VirtualNodeEmbedding.compute_mask  (lib/base/graph_layers/virtual_nodes.py:47-50:
//                             num_virtual_nodes leading True entries)
//   attention mask M [B,N,N,H] float : AdjMatModel.get_edge_mask (lib/models/graph_model_base.py:131-142:
//                             tile(adj[...,None], [1,1,1,H])), VNModel.get_edge_mask (:248-268: rows / columns
//                             of the virtual nodes are all ones)
// HBM-bound streaming kernels: one coalesced read, one coalesced write; the tile kernel writes
// 32-byte head vectors as two float4 stores.
#include "egt_common.h"

// ---- node mask from integer features: out[b, nvn + i] = (x[b,i] + 1) != 0 ; out[b, < nvn] = 1
__global__ void __launch_bounds__(256) k_mask_nodes_i32(const int32_t* __restrict__ x, int B, int N, int nvn,
                                                       uint8_t* __restrict__ out) {
  const int NO = N + nvn;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * NO) return;
  const int b = (int)(i / NO), c = (int)(i % NO);
  out[i] = c < nvn ? (uint8_t)1 : (uint8_t)((x[(long)b * N + (c - nvn)] + 1) != 0);
}

// ---- node mask from float feature rows (keras Masking): any(x[row, :] != mask_value); a wave
// handles 64 / LPR rows, LPR lanes per row (LPR = power of two <= 64 covering `width` in strides)
__global__ void __launch_bounds__(256) k_mask_nodes_f32(const float* __restrict__ x, long rows, int width,
                                                       float mask_value, int N, int nvn,
                                                       uint8_t* __restrict__ out) {
  // one thread per row for narrow rows is uncoalesced; use 8 lanes per row (feature rows are 1..~10 wide
  // in the reference's datasets) -- any wider row is walked in strides of 8
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = t >> 3;
  const int sub = (int)(t & 7);
  int hit = 0;
  if (row < rows)
    for (int c = sub; c < width; c += 8) hit |= (x[row * width + c] != mask_value) ? 1 : 0;
  hit |= __shfl_xor(hit, 1, 64);
  hit |= __shfl_xor(hit, 2, 64);
  hit |= __shfl_xor(hit, 4, 64);
  if (row < rows && sub == 0) {
    const long b = row / N, i = row % N;
    out[b * (N + nvn) + nvn + i] = (uint8_t)hit;
  }
  if (nvn > 0 && t < (rows / N) * nvn) {   // the virtual nodes' leading True entries
    const long b = t / nvn, v = t % nvn;
    out[b * (N + nvn) + v] = 1;
  }
}

// ---- M[b, l, m, 0..H) = (l < nvn || m < nvn) ? 1 : adj[b, l - nvn, m - nvn]   (H = 8: two float4)
template <int H>
__global__ void __launch_bounds__(256) k_mask_constrained(const float* __restrict__ adj, int B, int N, int nvn,
                                                         float* __restrict__ M) {
  const int NO = N + nvn;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // pair index in the OUTPUT geometry
  if (i >= (long)B * NO * NO) return;
  const int m = (int)(i % NO);
  const long r = i / NO;
  const int l = (int)(r % NO), b = (int)(r / NO);
  const float v = (l < nvn || m < nvn) ? 1.0f : adj[((long)b * N + (l - nvn)) * N + (m - nvn)];
  float* o = M + i * H;
  if (H % 4 == 0) {
#pragma unroll
    for (int k = 0; k < H / 4; ++k) reinterpret_cast<float4*>(o)[k] = make_float4(v, v, v, v);
  } else {
#pragma unroll
    for (int k = 0; k < H; ++k) o[k] = v;
  }
}
__global__ void __launch_bounds__(256) k_mask_constrained_anyh(const float* __restrict__ adj, int B, int N, int nvn,
                                                              int H, float* __restrict__ M) {
  const int NO = N + nvn;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // element index of M
  if (i >= (long)B * NO * NO * H) return;
  const long pr = i / H;
  const int m = (int)(pr % NO);
  const long r = pr / NO;
  const int l = (int)(r % NO), b = (int)(r / NO);
  M[i] = (l < nvn || m < nvn) ? 1.0f : adj[((long)b * N + (l - nvn)) * N + (m - nvn)];
}

extern "C" int egt_node_mask_from_features(const int32_t* features, int32_t B, int32_t N, int32_t num_virtual_nodes,
                                           uint8_t* mask, void* stream) {
  if (!features || !mask) EGT_FAIL(EGT_E_NULL, "features/mask is NULL");
  if (B <= 0 || N <= 0 || num_virtual_nodes < 0) EGT_FAIL(EGT_E_SHAPE, "B, N must be positive, num_virtual_nodes >= 0");
  const long n = (long)B * (N + num_virtual_nodes);
  EGT_LAUNCH("k_mask_nodes", k_mask_nodes_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             features, B, N, num_virtual_nodes, mask);
  EGT_HIP_LAUNCH_CHECK("egt_node_mask_from_features");
  return EGT_OK;
}

extern "C" int egt_node_mask_from_float_features(const float* features, int32_t B, int32_t N, int32_t width,
                                                 float mask_value, int32_t num_virtual_nodes, uint8_t* mask,
                                                 void* stream) {
  if (!features || !mask) EGT_FAIL(EGT_E_NULL, "features/mask is NULL");
  if (B <= 0 || N <= 0 || width <= 0 || num_virtual_nodes < 0) EGT_FAIL(EGT_E_SHAPE, "B, N, width must be positive");
  const long rows = (long)B * N;
  const long threads = rows * 8 > (long)B * num_virtual_nodes ? rows * 8 : (long)B * num_virtual_nodes;
  EGT_LAUNCH("k_mask_nodes", k_mask_nodes_f32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
             (hipStream_t)stream, features, rows, width, mask_value, N, num_virtual_nodes, mask);
  EGT_HIP_LAUNCH_CHECK("egt_node_mask_from_float_features");
  return EGT_OK;
}

extern "C" int egt_constrained_edge_mask(const float* adj, int32_t B, int32_t N, int32_t H, int32_t num_virtual_nodes,
                                         float* M, void* stream) {
  if (!adj || !M) EGT_FAIL(EGT_E_NULL, "adj/M is NULL");
  if (B <= 0 || N <= 0 || H <= 0 || num_virtual_nodes < 0) EGT_FAIL(EGT_E_SHAPE, "B, N, H must be positive");
  const long NO = N + num_virtual_nodes;
  const long pairs = (long)B * NO * NO;
  if (pairs * H > 0x7FFFFFFF00ll) EGT_FAIL(EGT_E_SHAPE, "mask too large");
  if (H == 8) {
    EGT_LAUNCH("k_mask_constrained", k_mask_constrained<8>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, adj, B, N, num_virtual_nodes, M);
  } else {
    const long n = pairs * H;
    EGT_LAUNCH("k_mask_constrained", k_mask_constrained_anyh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, adj, B, N, num_virtual_nodes, H, M);
  }
  EGT_HIP_LAUNCH_CHECK("egt_constrained_edge_mask");
  return EGT_OK;
}
