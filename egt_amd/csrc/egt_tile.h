// Pair-tile helpers shared by the fused kernels (egt_block.hip, egt_ffn.hip): coalesced
// 16-pair tile transfers HBM <-> registers <-> XOR-swizzled LDS tile, MFMA fragment access
// (lane (p = lane&15, q = lane>>4) owns row p, channels 16t + 4q + {0..3}), LayerNorm of the
// lane's fragments.
#pragma once
#include "egt_block.h"

// XOR swizzle of the 16-byte slots of a De = 64 tile row: slot' = slot ^ swz(row).
// (A GF(2)-linear map that is conflict-free for every access pattern of the pair kernels was measured as a null in
// round 3: profiles/r03_mfma_valu_issue.md section 8.)
template <int DE>
__device__ __forceinline__ int swz(int row) {
  return DE == 64 ? (row & 15) : 0;
}

// LDS hand-offs inside one wavefront: DS operations of a wave complete in order,
// so draining lgkmcnt (never vmcnt: global loads/stores stay in flight) plus a
// scheduling barrier is all the ordering the tile round-trips need.
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------ tile helpers -----
template <int DE>
struct TileRegs { float4 v[(Geo<DE>::NF4 + 63) / 64]; };

// Cache-policy hints of the streamed edge tiles (round 5).  The 134 MB tensor one launch of a stack hands to the next (e_{l+1} in the
// forward, de in the backward) fits the 256 MB memory-side cache; what a launch reads ONCE (its e_l input) is loaded non-temporal so
// that it does not push the hand-over tensor out (k_block_fwd 60.7 -> 59.0 us, k_block_bwd_v5 100.9 -> 97.7 us; non-temporal STORES
// of the hand-over tensor instead: 67 us -- the reuse is real).
typedef float egt_nt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 egt_ld4_nt(const float* p) {
  const egt_nt_v4f t = __builtin_nontemporal_load(reinterpret_cast<const egt_nt_v4f*>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void egt_st4_nt(float* p, float4 v) {
  __builtin_nontemporal_store((egt_nt_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<egt_nt_v4f*>(p));
}

// issue the coalesced 16-byte loads of one 16-pair tile; rows >= rows_valid are
// redirected to row 0 (always valid) so every load is unconditional.  NT: non-temporal loads (read-once tensors)
template <int DE, bool NT = false>
__device__ __forceinline__ void tile_gload(TileRegs<DE>& r, const float* src, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int f = i * 64 + lane;
    if (G::NF4 < 64) f &= (G::NF4 - 1);
    const int row = f / G::NSLOT;
    const int fc = row < rows_valid ? f : f - row * G::NSLOT;
    if (NT) r.v[i] = egt_ld4_nt(src + (size_t)fc * 4);
    else r.v[i] = *reinterpret_cast<const float4*>(src + (size_t)fc * 4);
  }
}
template <int DE>
__device__ __forceinline__ void tile_lds_put(float* tl, const TileRegs<DE>& r, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = i * 64 + lane;
    if (G::NF4 >= 64 || f < G::NF4) {
      const int row = f / G::NSLOT, slot = f % G::NSLOT;
      const bool ok = row < rows_valid;
      float4 v = r.v[i];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2)) = v;
    }
  }
}
template <int DE, bool NT = false>
__device__ __forceinline__ void tile_from_lds(const float* tl, float* dst, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = i * 64 + lane;
    if (G::NF4 >= 64 || f < G::NF4) {
      const int row = f / G::NSLOT, slot = f % G::NSLOT;
      if (row < rows_valid) {
        const float4 v = *reinterpret_cast<const float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2));
        if (NT) egt_st4_nt(dst + (size_t)f * 4, v);
        else *reinterpret_cast<float4*>(dst + (size_t)f * 4) = v;
      }
    }
  }
}

// ---- bf16 edge tensors (desc.dtype == EGT_BF16): same tiles, 8-byte global accesses; the
// arithmetic, the LDS tiles and everything node-side stay fp32 ----
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
// fp32 -> bf16 pairs on v_cvt_pk_bf16_f32 (gfx950: round to nearest even, NaN stays NaN): one
// instruction per two elements instead of the ~14 integer ops of a software rounding
typedef __bf16 egt_bf2 __attribute__((ext_vector_type(2)));
typedef float egt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2_to_bf2(float lo, float hi) {
  const egt_bf2 b = __builtin_convertvector((egt_f2){lo, hi}, egt_bf2);
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ uint2 f4_to_bf4(float4 v) { return make_uint2(f2_to_bf2(v.x, v.y), f2_to_bf2(v.z, v.w)); }
template <int DE, bool NT = false>
__device__ __forceinline__ void tile_gload(TileRegs<DE>& r, const uint16_t* src, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int f = i * 64 + lane;
    if (G::NF4 < 64) f &= (G::NF4 - 1);
    const int row = f / G::NSLOT;
    const int fc = row < rows_valid ? f : f - row * G::NSLOT;
    r.v[i] = bf4_to_f4(*reinterpret_cast<const uint2*>(src + (size_t)fc * 4));
  }
}
template <int DE, bool NT = false>
__device__ __forceinline__ void tile_from_lds(const float* tl, uint16_t* dst, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = i * 64 + lane;
    if (G::NF4 >= 64 || f < G::NF4) {
      const int row = f / G::NSLOT, slot = f % G::NSLOT;
      if (row < rows_valid)
        *reinterpret_cast<uint2*>(dst + (size_t)f * 4) =
            f4_to_bf4(*reinterpret_cast<const float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2)));
    }
  }
}
template <bool BF> struct EdgeT { typedef float type; };
template <> struct EdgeT<true> { typedef uint16_t type; };

template <int DE>
__device__ __forceinline__ void tile_lds_get(const float* tl, TileRegs<DE>& r, int lane) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int f = i * 64 + lane;
    if (G::NF4 < 64) f &= (G::NF4 - 1);
    const int row = f / G::NSLOT, slot = f % G::NSLOT;
    r.v[i] = *reinterpret_cast<const float4*>(tl + row * DE + ((slot ^ swz<DE>(row)) << 2));
  }
}
template <int DE>
__device__ __forceinline__ void tile_gstore(const TileRegs<DE>& r, float* dst, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = (G::NF4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = i * 64 + lane;
    if ((G::NF4 >= 64 || f < G::NF4) && f / G::NSLOT < rows_valid)
      *reinterpret_cast<float4*>(dst + (size_t)f * 4) = r.v[i];
  }
}

// fragment (p,q): channels 16*t + 4*q + {0..3}
template <int DE>
__device__ __forceinline__ float4 frag_read(const float* tl, int p, int q, int t) {
  if (16 * t + 4 * q < DE)
    return *reinterpret_cast<const float4*>(tl + p * DE + (((4 * t + q) ^ swz<DE>(p)) << 2));
  return make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int DE>
__device__ __forceinline__ void frag_write(float* tl, int p, int q, int t, float4 v) {
  if (16 * t + 4 * q < DE)
    *reinterpret_cast<float4*>(tl + p * DE + (((4 * t + q) ^ swz<DE>(p)) << 2)) = v;
}
template <int DE>
__device__ __forceinline__ float elem_read(const float* tl, int row, int c) {
  return tl[row * DE + ((((c >> 2) ^ swz<DE>(row)) << 2) | (c & 3))];
}

__device__ __forceinline__ float sum_over_q(float v) {  // lanes p, p+16, p+32, p+48
  return sum_xor32(sum_xor16(v));
}

// LayerNorm of the lane's fragments (two-pass moments like tf.nn.moments); returns rstd
template <int DE>
__device__ __forceinline__ float ln_frags(float4 (&x)[Geo<DE>::TILES], int q, float eps, bool on = true) {
  // on == false ('bias' edge channels: projections of the RAW e, graph_xformer_model_base.py:173-190):
  // mean 0 / rstd 1 are selected instead of the statistics -- branch-free, x passes through
  using G = Geo<DE>;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) s += (x[t].x + x[t].y) + (x[t].z + x[t].w);
  const float mu = on ? sum_over_q(s) * (1.0f / DE) : 0.0f;
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) {
    if (16 * t + 4 * q < DE) {
      x[t].x -= mu; x[t].y -= mu; x[t].z -= mu; x[t].w -= mu;
      v = fmaf(x[t].x, x[t].x, v); v = fmaf(x[t].y, x[t].y, v);
      v = fmaf(x[t].z, x[t].z, v); v = fmaf(x[t].w, x[t].w, v);
    }
  }
  const float rstd = on ? rsqrtf(sum_over_q(v) * (1.0f / DE) + eps) : 1.0f;
#pragma unroll
  for (int t = 0; t < G::TILES; ++t) { x[t].x *= rstd; x[t].y *= rstd; x[t].z *= rstd; x[t].w *= rstd; }
  return rstd;
}

// ---- split-bf16 products (EGT_MM_BF16X3): an fp32 operand v is carried as hi + lo bfloat16 terms
// (hi = bf16(v), lo = bf16(v - hi)); a . b = a_hi.b_hi + a_lo.b_hi + a_hi.b_lo on v_mfma_f32_16x16x32_bf16
// with fp32 accumulation: per-product error <= 2^-16 at 3/16 of the fp32 MFMA cost ----
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
union Bf8 { bf16x8_t v; uint32_t u[4]; uint4 q; };
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);   // v_cvt_pk_bf16_f32 (RNE)
  return *reinterpret_cast<const uint32_t*>(&r);
}
// 8 fp32 values -> hi / lo bf16x8 (lo only when SPLIT)
template <bool SPLIT>
__device__ __forceinline__ void split8(const float (&v)[8], Bf8& hi, Bf8& lo) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t h = pk_bf16(v[2 * k], v[2 * k + 1]);
    hi.u[k] = h;
    if (SPLIT) {
      const float f0 = __uint_as_float(h << 16), f1 = __uint_as_float(h & 0xFFFF0000u);
      lo.u[k] = pk_bf16(v[2 * k] - f0, v[2 * k + 1] - f1);
    } else {
      lo.u[k] = 0u;
    }
  }
}

// acc += sum_s A[blk0 + s] . B[s] with the 3-term split (or the hi term only); slab blocks are [part][lane] uint4
template <int NS, bool SPLIT>
__device__ __forceinline__ v4f bf_gemm(const float* slab, int blk0, int lane, const Bf8 (&bh)[NS], const Bf8 (&bl)[NS], v4f acc) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    Bf8 ah, al;
    ah.q = *reinterpret_cast<const uint4*>(slab + ((size_t)((blk0 + s) * 2 + 0) * 64 + lane) * 4);
    acc = MFMA_BF(ah.v, bh[s].v, acc);
    if (SPLIT) {
      al.q = *reinterpret_cast<const uint4*>(slab + ((size_t)((blk0 + s) * 2 + 1) * 64 + lane) * 4);
      acc = MFMA_BF(al.v, bh[s].v, acc);
      acc = MFMA_BF(ah.v, bl[s].v, acc);
    }
  }
  return acc;
}
// the lane's 4 NT values (channel tiles 0 .. NT-1) -> ceil(NT / 2) split B operands
template <int NT, bool SPLIT>
__device__ __forceinline__ void split_tiles(const v4f (&v)[NT], Bf8 (&hi)[(NT + 1) / 2], Bf8 (&lo)[(NT + 1) / 2]) {
#pragma unroll
  for (int s = 0; s < (NT + 1) / 2; ++s) {
    const v4f a0 = v[2 * s];
    const v4f a1 = (2 * s + 1 < NT) ? v[(2 * s + 1 < NT) ? 2 * s + 1 : 0] : (v4f){0.f, 0.f, 0.f, 0.f};
    const float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    split8<SPLIT>(f, hi[s], lo[s]);
  }
}

