// LDS-DMA staging of 16-pair tiles (HBM -> LDS without a VGPR destination) and the transposed-element read of a swizzled
// tile (k_block_bwd_v5, egt_block.hip).
#pragma once
#include "egt_tile.h"

template <int DE>
__device__ __forceinline__ unsigned dma_lane_offset(int lane) {
  // byte offset (inside the tile) of the 16-byte piece lane `lane` fetches for chunk 0
  if (DE == 64) return (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ swz<64>(lane >> 4)) << 4));
  return (unsigned)(lane * 16);
}
// one 16-pair tile HBM -> LDS; `lds` = byte address of the tile in LDS (wave-uniform), `src` = the
// tile's first byte in HBM (wave-uniform), off0 = dma_lane_offset.  Chunk i (rows 4i .. 4i+3 at
// De = 64) differs from chunk 0 by +1024 i bytes and, for the swizzle, by flipping slot bits 2..3
// with i: one XOR with 1088 i (the offsets of chunk 0 are < 1024, so the add is an OR is an XOR).
// NT: the non-temporal form of the request (a tensor this launch reads once: egt_tile.h, cache-policy hints)
#define EGT_TILE_DMA_BODY(MOD) \
  constexpr int NI = Geo<DE>::NF4 / 64; \
  static_assert(Geo<DE>::NF4 % 64 == 0 && NI >= 1 && NI <= 4, "whole 1 KiB chunks"); \
  unsigned keep, t; \
  constexpr unsigned X = DE == 64 ? 1088u : 1024u; \
  constexpr unsigned X1 = X, X2 = 2 * X, X3 = 3 * X; \
  if (NI == 4) \
    asm volatile( \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2" MOD "\n\t" \
        "v_xor_b32 %1, %5, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "v_xor_b32 %1, %6, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "v_xor_b32 %1, %7, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "s_mov_b32 m0, %0" \
        : "=&s"(keep), "=&v"(t) : "s"(src), "v"(off0), "s"(lds), "i"(X1), "i"(X2), "i"(X3) : "memory", "scc"); \
  else if (NI == 3) \
    asm volatile( \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2" MOD "\n\t" \
        "v_xor_b32 %1, %5, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "v_xor_b32 %1, %6, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "s_mov_b32 m0, %0" \
        : "=&s"(keep), "=&v"(t) : "s"(src), "v"(off0), "s"(lds), "i"(X1), "i"(X2) : "memory", "scc"); \
  else if (NI == 2) \
    asm volatile( \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %2" MOD "\n\t" \
        "v_xor_b32 %1, %5, %3\n\ts_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\t" \
        "s_mov_b32 m0, %0" \
        : "=&s"(keep), "=&v"(t) : "s"(src), "v"(off0), "s"(lds), "i"(X1) : "memory", "scc"); \
  else \
    asm volatile( \
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1" MOD "\n\t" \
        "s_mov_b32 m0, %0" \
        : "=&s"(keep) : "s"(src), "v"(off0), "s"(lds) : "memory"); \
  (void)t;
template <int DE, bool NT = false>
__device__ __forceinline__ void tile_dma(unsigned lds, const float* src, unsigned off0) {
  if (NT) { EGT_TILE_DMA_BODY(" nt") } else { EGT_TILE_DMA_BODY("") }
}
#undef EGT_TILE_DMA_BODY
// The same tile when fewer than 16 of its pair rows exist (the last key tile of a ragged N): rows past `rows_valid` would
// lie in the NEXT query row of the tensor (past its end for the last one), so their pieces are fetched from the last valid row
// instead -- finite stand-ins that the kernel's key-validity selects keep out of every result.  Same NI instructions as
// tile_dma (the callers' counted vmcnt waits do not change).
template <int DE>
__device__ __forceinline__ void tile_dma_ragged(unsigned lds, const float* src, int lane, int rows_valid) {
  using G = Geo<DE>;
  constexpr int NI = G::NF4 / 64;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int f = 64 * i + lane, row = f / G::NSLOT, slot = f % G::NSLOT;
    const int rc = min(row, rows_valid - 1);
    const unsigned off = (unsigned)((rc * G::NSLOT + (DE == 64 ? (slot ^ swz<64>(row)) : slot)) << 4);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(src), "v"(off), "s"(lds + 1024u * i) : "memory");
  }
}
// wait until at most N of the wave's vector-memory operations are outstanding (they retire in order)
template <int N_>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N_) : "memory"); }

// transposed element (row q + 4s, channel 16t + p) of a swizzled De = 64 tile: the XOR swizzle splits into a
// per-lane part and a part that depends only on (s, t), so ONE address register + compile-time offsets
// replace 32 per-(s,t) address registers:  floats = [64 q + 4 ((p >> 2) ^ q) + (p & 3)] + 256 s + 16 (t ^ s)
template <int DE>
__device__ __forceinline__ float elem_read_st(const float* lane_base, const float* tl, int p, int q, int s, int t) {
  if (DE == 64) return lane_base[256 * s + 16 * (t ^ s)];
  return elem_read<DE>(tl, q + 4 * s, 16 * t + p);
}

// Pull one 16-pair tile (contiguous, <= 4 KiB) into L2 without a VGPR destination: ONE LDS-DMA instruction, lane i fetching
// the dword at byte 64 i (every 128-byte line of the tile is touched), landing in a dump area of LDS that nobody reads.
__device__ __forceinline__ void tile_prefetch(unsigned lds_dump, const float* src, unsigned lane_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(src), "v"(lane_off), "s"(lds_dump) : "memory");
}

