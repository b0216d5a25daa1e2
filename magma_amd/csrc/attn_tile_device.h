// attn_tile_device.h -- LDS tile images, LDS-DMA loaders and fragment bursts shared by the flash-attention
// forward and backward kernels (head dim 256, KV / query tiles of 32, 8 waves per workgroup).
#pragma once
#include "common.h"

constexpr int DH = 256;
// LDS images, conflict-free for the ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS):
//   row tiles [32][256]: 512-B rows, 16-B chunk c of row r stored at position c ^ row_swz(r)
//   T tiles  [256][32]:  64-B rows,  16-B chunk c of row r stored at position c ^ t_swz(r)
// LDS-DMA writes 64 lanes x 16 B linearly, so the swizzles are applied to the SOURCE addresses.
MG_DEV int row_swz(int row) { return (row & 3) | ((row >> 3) << 2); }
MG_DEV int t_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }   // {0,2,3,1}[(row>>2)&3]
constexpr int ROW_TILE = 32 * DH * 2;     // 16 KiB
constexpr int T_TILE = DH * 32 * 2;       // 16 KiB

// One tile = 16 blocks of 1 KiB; wave w (of NW) moves blocks w*16/NW .. +16/NW-1.
// rows [r0, r0+32) of a row-major [*][256] array (row stride in elements), rows clamped to rmax-1
template <int NW = 8>
MG_DEV void dma_rows(char* tile, const mg_bf16* base, int64_t row_stride, int r0, int rmax, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int blk = wave * (16 / NW) + i;
    const int row = blk * 2 + (lane >> 5);
    const int c = (lane & 31) ^ row_swz(row);
    glds16a(base + (int64_t)min(r0 + row, rmax - 1) * row_stride + c * 8, tile + blk * 1024);
  }
}
// positions [c0, c0+32) (c0 % 32 == 0) of a transposed operand in the column-tiled layout [tile][256][32]: the
// 16-KiB tile is contiguous, a wave instruction reads 1 KiB of it (16 rows x 64 B), the swizzle permutes 16-byte
// chunks inside each 64-byte row only
template <int NW = 8>
MG_DEV void dma_cols(char* tile, const mg_bf16* base_t, int ld, int c0, int wave, int lane) {
  (void)ld;
  const mg_bf16* src = base_t + (int64_t)(c0 >> 5) * (DH * 32);
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int blk = wave * (16 / NW) + i;
    const int row = blk * 16 + (lane >> 2);
    const int c = (lane & 3) ^ t_swz(row);
    glds16a(src + row * 32 + c * 8, tile + blk * 1024);
  }
}

// the same two with the destination tile given as a 32-bit LDS address (see lds_u32)
template <int NW = 8>
MG_DEV void dma_rows(uint32_t tile, const mg_bf16* base, int64_t row_stride, int r0, int rmax, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int blk = wave * (16 / NW) + i;
    const int row = blk * 2 + (lane >> 5);
    const int c = (lane & 31) ^ row_swz(row);
    glds16au(base + (int64_t)min(r0 + row, rmax - 1) * row_stride + c * 8, tile + blk * 1024);
  }
}
template <int NW = 8>
MG_DEV void dma_cols(uint32_t tile, const mg_bf16* base_t, int ld, int c0, int wave, int lane) {
  (void)ld;
  const mg_bf16* src = base_t + (int64_t)(c0 >> 5) * (DH * 32);
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int blk = wave * (16 / NW) + i;
    const int row = blk * 16 + (lane >> 2);
    const int c = (lane & 3) ^ t_swz(row);
    glds16au(src + row * 32 + c * 8, tile + blk * 1024);
  }
}

// fragment bursts: 8 x ds_read_b128 of one operand, and the 8 MFMAs of a 16x16 tile over d = 256
#define MG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef MG_ATTN_MFMA_PRIO
#define MG_ATTN_MFMA_PRIO 0   // wave priority inside MFMA bursts: raising it measured no gain (fwd 1.03 vs 1.04 ms, bwd 3.30 vs 3.25 ms)
#endif
MG_DEV void rd_row8(bf16x8 (&f)[8], const char* row, int lq, int sw) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) f[ks] = *(const bf16x8*)(row + (((ks * 4 + lq) ^ sw) << 4));
}
MG_DEV void rd_t8(bf16x8 (&f)[8], const char* tp) {
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) f[dt] = *(const bf16x8*)(tp + dt * 1024);
}
MG_DEV f32x4 mma8(const bf16x8 (&a)[8], const bf16x8 (&b)[8]) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_setprio(MG_ATTN_MFMA_PRIO);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[ks], c, 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  return c;
}
