// attention_bwd.hip -- causal flash-attention backward, head dim 256 (gfx950).
//
// Three kernels that recompute P from (q, k, lse) -- no S x S tensor in HBM:
//
//   attn_bwd_dq_kernel    one workgroup per 128 queries (8 waves x 16), loops over
//                         KV tiles of 32.  Works in the transposed frame of the
//                         forward kernel (lane&15 = query):
//                           S^T  = K Q^T        dP^T = V dO^T
//                           dS^T = P^T o (dP^T - D) / 16
//                           dQ^T += K^T dS^T
//   attn_bwd_dkdv_kernel  one workgroup per 128 keys (8 waves x 16, lane&15 = key),
//                         loops over query tiles of 32 from the diagonal down:
//                           S  = Q K^T          dP = dO V^T
//                           dV^T += dO^T P      dK^T += Q^T dS
//                         (two instantiations: <false> = dV, <true> = dK)
//
// MFMA contracts over 8 consecutive elements per lane, so every operand is
// needed with its contraction index contiguous: K,V,Q,dO row-major [s][256] for
// the d-contractions and K^T,Q^T,dO^T [256][s] for the s-contractions (made by
// mg_head_transpose_bf16).  The same row permutation as in the forward kernel
// turns accumulator registers directly into the next product's operand.
//
// Pipeline.  A tile step is only 32-48 MFMAs per wave, far shorter than an HBM/L2
// round trip, so the tiles stream through a ring of LDS stages filled by LDS-DMA
// (global_load_lds, no register staging): the loads of tile t+2 (t+3 for dV) are
// issued while tile t is multiplied, each wave waits for ITS pieces with a counted
// s_waitcnt vmcnt(N) and ONE barrier per tile both publishes tile t and retires the
// stage that tile t+2 overwrites.  Past the last tile the ring re-loads the last tile
// (in bounds, never read) so the wait counts stay constant.
#include "common.h"
#include "attn_tile_device.h"
#include "attn_bwd_device.h"
#include <stdlib.h>

namespace {

// acc[dt] = columns dt*16 + lq*4 .. +3 of row s (sequence position) of head (b, h).  A lane's 4 columns are 8 bytes; stored
// like that the epilogue is 16 dwordx2 stores per lane and store-ISSUE-bound (MI355X_MICROARCH.md, attention epilogue store
// tail).  v_permlane16_swap trades halves between the lane groups lq and lq^1 (same row): even groups end up with 8
// consecutive columns of tile dt, odd groups with 8 of tile dt+1 -- 8 dwordx4 stores per lane instead.
MG_DEV void store_grad_row(const GradOut& g, const f32x4 (&acc)[16], int b, int h, int H, int S, int s, int lq) {
  uint32_t wlo[16], whi[16];
  const bool rot = g.merged && g.which < 2;
  const int half_rot = g.rot_dim >> 1;
#pragma unroll
  for (int dt = 0; dt < 16; ++dt) {
    float x0 = acc[dt][0], x1 = acc[dt][1], x2 = acc[dt][2], x3 = acc[dt][3];
    if (rot && dt * 16 + lq * 4 < g.rot_dim) {       // rot_dim % 8 == 0: a lane's 4 columns are inside or outside
      const int pi = (int)((int64_t)s * half_rot) + dt * 8 + lq * 2;
      const float s0 = g.sin_t[pi], c0 = g.cos_t[pi], s1 = g.sin_t[pi + 1], c1 = g.cos_t[pi + 1];
      const float y0 = x0 * c0 + x1 * s0, y1 = x1 * c0 - x0 * s0;
      const float y2 = x2 * c1 + x3 * s1, y3 = x3 * c1 - x2 * s1;
      x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    }
    wlo[dt] = pack2bf(x0, x1);
    whi[dt] = pack2bf(x2, x3);
  }
  mg_bf16* row = grad_row_ptr(g, b, h, H, S, s);
  const int odd = lq & 1;
#pragma unroll
  for (int dt = 0; dt < 16; dt += 2) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(wlo[dt], wlo[dt + 1], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(whi[dt], whi[dt + 1], false, false);
    const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
    *(u32x4*)(row + (dt + odd) * 16 + (lq - odd) * 4) = w;
  }
}

constexpr int DQ_STAGE = 2 * ROW_TILE + T_TILE;             // K rows | V rows | K^T
constexpr int DQ_STAGES = 3;
constexpr int DK_STAGE = 2 * ROW_TILE + T_TILE + LD_TILE;   // Q rows | dO rows | Q^T | lse,D
constexpr int DK_STAGES = 3;
constexpr int DV_STAGE = ROW_TILE + T_TILE + LD_TILE;       // Q rows | dO^T | lse,D
constexpr int DV_STAGES = 4;

// ld2[b,h,s] = {-16 lse, -D}, D = rowsum(dO o O): the two per-query statistics, negated and (the first) in raw-score units,
// i.e. the initial values of the score accumulators of the 32-row kernels (S - 16 lse, dP - D: attention_bwd32.hip);
// one wave per (b, s, h) row of 256
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ O,
                                                            const float* __restrict__ lse, float* __restrict__ ld2,
                                                            int B, int H, int S, int64_t ld_o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // over B*S*H, (b,s,h) order = memory order of [M, H*256]
  if (row >= (int64_t)B * S * H) return;
  const u32x2 a = *(const u32x2*)(dO + row * DH + lane * 4);
  const u32x2 o = *(const u32x2*)(O + (row / H) * ld_o + (row % H) * DH + lane * 4);   // O rows may sit in a wider buffer ([ctx | t])
  float s = bflo(a[0]) * bflo(o[0]) + bfhi(a[0]) * bfhi(o[0]) + bflo(a[1]) * bflo(o[1]) + bfhi(a[1]) * bfhi(o[1]);
  s = wave_sum(s);
  if (lane == 0) {
    const int h = (int)(row % H);
    const int64_t bs = row / H;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    const int64_t i = ((int64_t)b * H + h) * S + sidx;
    ld2[i * 2] = -16.0f * lse[i];
    ld2[i * 2 + 1] = -s;
  }
}

// The same statistics computed while dO is transposed: one workgroup per (b, h, 32 positions) reads the dO and O tiles
// once, writes dO^T in the column-tiled layout of mg_head_transpose_bf16 (zero padded) and ld2 -- the prep pass and the
// dO transpose of the two-pass form in one.  grid (ceil(S/32), B*H), 256 threads; a row is 32 consecutive lanes.
__global__ __launch_bounds__(256) void attn_bwd_prep_t_kernel(const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ O,
                                                              const float* __restrict__ lse, float* __restrict__ ld2,
                                                              mg_bf16* __restrict__ dOt, int ld_t, int B, int H, int S, int64_t ld_o) {
  __shared__ __attribute__((aligned(16))) mg_bf16 tile[32 * DH];
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int s0 = blockIdx.x * 32;
  const int64_t dmodel = (int64_t)H * DH;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31, sidx = s0 + row;
    u32x4 a = (u32x4){0u, 0u, 0u, 0u};
    float dot = 0.f;
    if (sidx < S) {
      const int64_t off = ((int64_t)b * S + sidx) * dmodel + h * DH + c * 8;
      a = *(const u32x4*)(dO + off);
      const u32x4 o = *(const u32x4*)(O + ((int64_t)b * S + sidx) * ld_o + h * DH + c * 8);
#pragma unroll
      for (int w = 0; w < 4; ++w) dot += bflo(a[w]) * bflo(o[w]) + bfhi(a[w]) * bfhi(o[w]);
    }
    *(u32x4*)(tile + row * DH + c * 8) = a;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 32);
    if (c == 0 && sidx < S) {
      const int64_t i = ((int64_t)bh * S + sidx);
      ld2[i * 2] = -16.0f * lse[i];
      ld2[i * 2 + 1] = -dot;
    }
  }
  __syncthreads();
  mg_bf16* d = dOt + (((int64_t)bh * (ld_t >> 5) + blockIdx.x) * DH + tid) * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      o[w] = (uint32_t)tile[(g * 8 + w * 2) * DH + tid] | ((uint32_t)tile[(g * 8 + w * 2 + 1) * DH + tid] << 16);
    *(u32x4*)(d + g * 8) = o;
  }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ kt, const mg_bf16* __restrict__ dO, const float* __restrict__ ld2,
    const GradOut gout, int B, int H, int S, int ld_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  // all query blocks of one (b,h) run on ONE XCD, so its K / V / K^T stream is fetched from HBM once and
  // re-read from that XCD's L2 by the other blocks
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;            // longest (latest) query blocks first
  const int qrow = qt0 + wave * 16 + li, qrow_c = min(qrow, S - 1);
  const mg_bf16* kb = k + (int64_t)bh * S * DH;
  const mg_bf16* vb = v + (int64_t)bh * S * DH;
  const mg_bf16* ktb = kt + (int64_t)bh * DH * ld_t;
  const int dmodel = H * DH;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  auto issue = [&](int t, int buf) {
    const int c0 = min(t, ntiles - 1) * 32;
    const uint32_t st = smem_u + (uint32_t)(buf * DQ_STAGE);
    dma_rows(st, kb, DH, c0, S, wave, lane);
    dma_rows(st + ROW_TILE, vb, DH, c0, S, wave, lane);
    dma_cols(st + 2 * ROW_TILE, ktb, ld_t, c0, wave, lane);
  };
  issue(0, 0);
  issue(1, 1);

  bf16x8 qf[8], dof[8];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + lq * 8;
    const mg_bf16* dp = dO + (int64_t)(b * S + qrow_c) * dmodel + h * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 32); dof[ks] = *(const bf16x8*)(dp + ks * 32); }
  }
  const float sc2 = 0.0625f * 1.4426950408889634f;
  const float lse2 = -(ld2[((int64_t)bh * S + qrow_c) * 2] * sc2);     // lse log2 e
  const float Dq = -ld2[((int64_t)bh * S + qrow_c) * 2 + 1];
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int my_last = qt0 + wave * 16 + 15;     // key tiles past this wave's last query are fully masked
  const int tsw = t_swz(li);
  const int krow0 = (li >> 2) * 8 + (li & 3);   // row permutation: accumulator registers -> 8 consecutive keys
  const int sw0 = row_swz(krow0);

  int sc = 0;
  MG_USE8(qf); MG_USE8(dof);          // retire the ordinary loads in hipcc's scoreboard (see attention.hip)
  asm volatile("" ::"v"(lse2), "v"(Dq));
  for (int t = 0; t < ntiles; ++t) {
    MG_WAIT_VMCNT(6);                 // this wave's pieces of tile t have landed (tile t+1 may be in flight)
    MG_BARRIER_KEEP_DMA();            // tile t complete; everyone is done with tile t-1
    issue(t + 2, sc == 0 ? 2 : sc - 1);
    const int kv0 = t * 32;
    if (kv0 <= my_last) {
      const char* k_lds = smem + sc * DQ_STAGE;
      const char* v_lds = k_lds + ROW_TILE;
      const char* kt_lds = k_lds + 2 * ROW_TILE;
      // Fragment reads are issued in batches of 8 (one ds_read burst, counted lgkmcnt at first use) one
      // batch AHEAD of the MFMAs that consume them; sched_barrier pins that order.
      const char* kp = k_lds + krow0 * 512;
      const char* vp = v_lds + krow0 * 512;
      const char* tp = kt_lds + li * 64 + ((lq ^ tsw) << 4);
      bf16x8 fa[8], fb[8];
      f32x4 st[2], dp[2];
      rd_row8(fa, kp, lq, sw0);                    // K  keys tt=0
      rd_row8(fb, vp, lq, sw0);                    // V  keys tt=0
      MG_SCHED_FENCE();
      st[0] = mma8(fa, qf);
      MG_SCHED_FENCE();
      rd_row8(fa, kp + 4 * 512, lq, sw0);          // K  keys tt=1 (row + 4: same swizzle)
      MG_SCHED_FENCE();
      dp[0] = mma8(fb, dof);
      MG_SCHED_FENCE();
      rd_row8(fb, vp + 4 * 512, lq, sw0);          // V  keys tt=1
      MG_SCHED_FENCE();
      st[1] = mma8(fa, qf);
      MG_SCHED_FENCE();
      rd_t8(fa, tp);                               // K^T d-tiles 0..7
      MG_SCHED_FENCE();
      dp[1] = mma8(fb, dof);
      MG_SCHED_FENCE();
      // only this wave's diagonal tiles (and the ragged last tile) need the mask: -1e30 -> exp2(-huge) = 0
      if (kv0 + 31 > qt0 + wave * 16 || kv0 + 32 > S) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = kv0 + lq * 8 + j;
          st[j >> 2][j & 3] = (key > qrow || key >= S) ? -1e30f : st[j >> 2][j & 3];
        }
      }
      float ds[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[j >> 2][j & 3], sc2, -lse2));   // raw v_exp_f32
        ds[j] = p * (dp[j >> 2][j & 3] - Dq) * 0.0625f;
      }
      u32x4 dw;
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[j] = pack2bf(ds[2 * j], ds[2 * j + 1]);
      const bf16x8 dsf = __builtin_bit_cast(bf16x8, dw);
      MG_SCHED_FENCE();
      rd_t8(fb, tp + 8 * 1024);                    // K^T d-tiles 8..15 land under the first 8 MFMAs
      MG_SCHED_FENCE();
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[dt], dsf, acc[dt], 0, 0, 0);
      MG_SCHED_FENCE();
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[dt], dsf, acc[8 + dt], 0, 0, 0);
    }
    sc = sc == DQ_STAGES - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);                   // drain the ring's trailing loads before the wave retires
  if (qrow < S) store_grad_row(gout, acc, b, h, H, S, qrow, lq);
}

// ---------------------------------------------------------------------------
// DK = false: dV only (needs S -> P and dO^T);  DK = true: dK only (needs S, dP, dS and Q^T).
// One accumulator set (64 VGPRs) per instantiation instead of two keeps both at two waves per SIMD
// with room for a 3- / 4-deep LDS ring; the price is recomputing S once more.
template <bool DK>
__global__ __launch_bounds__(512) void attn_bwd_dkdv_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ qt, const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ dOt,
    const float* __restrict__ ld2, const GradOut gout, int B, int H, int S, int ld_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = DK ? DK_STAGE : DV_STAGE;
  constexpr int NST = DK ? DK_STAGES : DV_STAGES;
  constexpr int T_OFF = (DK ? 2 : 1) * ROW_TILE;
  constexpr int LD_OFF = T_OFF + T_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int nblk = (S + 127) >> 7;              // one (b,h) per XCD at a time, see attn_bwd_dq_kernel
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int k0 = (wg - bh * nblk) * 128;       // earliest key blocks (most query tiles) first
  const int key = k0 + wave * 16 + li, key_c = min(key, S - 1);
  const int dmodel = H * DH;
  const mg_bf16* qb = q + (int64_t)bh * S * DH;
  const mg_bf16* tb = (DK ? qt : dOt) + (int64_t)bh * DH * ld_t;
  const mg_bf16* dob = dO + (int64_t)b * S * dmodel + h * DH;   // row stride dmodel
  const float* ldb = ld2 + (int64_t)bh * S * 2;

  const int t_begin = k0 >> 5;                 // first query tile that can see key k0
  const int t_end = (S + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  auto issue = [&](int t, int buf) {
    const int q0 = min(t, t_end - 1) * 32;
    const uint32_t st = smem_u + (uint32_t)(buf * STAGE);
    dma_rows(st, qb, DH, q0, S, wave, lane);
    if constexpr (DK) dma_rows(st + ROW_TILE, dob, dmodel, q0, S, wave, lane);
    dma_cols(st + T_OFF, tb, ld_t, q0, wave, lane);
    glds4au(ldb + (int64_t)min(q0 + (lane >> 1), S - 1) * 2 + (lane & 1), st + LD_OFF);   // every wave writes the same 256 B
  };
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) issue(t_begin + i, i);

  bf16x8 kf[8], vf[DK ? 8 : 1];
  {
    const mg_bf16* kp = k + ((int64_t)bh * S + key_c) * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const bf16x8*)(kp + ks * 32);
    if (DK) {
      const mg_bf16* vp = v + ((int64_t)bh * S + key_c) * DH + lq * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) vf[ks] = *(const bf16x8*)(vp + ks * 32);
    }
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float sc2 = 0.0625f * 1.4426950408889634f;
  const int my_first = k0 + wave * 16;         // query tiles that end before this wave's first key are fully masked
  const int tsw = t_swz(li);
  const int qr0 = (li >> 2) * 8 + (li & 3);    // row permutation: accumulator registers -> 8 consecutive queries
  const int sw0 = row_swz(qr0);
  const uint32_t ls_addr = (uint32_t)(uintptr_t)(mg_lptr_t)(smem + LD_OFF + lq * 64);   // LDS byte address, stage 0

  int sc = 0;
  MG_USE8(kf);                        // retire the ordinary loads in hipcc's scoreboard (see attention.hip)
  if constexpr (DK) MG_USE8(vf);
  for (int t = t_begin; t < t_end; ++t) {
    if constexpr (DK) { MG_WAIT_VMCNT(7); } else { MG_WAIT_VMCNT(10); }   // (NST-2) later tiles may be in flight
    MG_BARRIER_KEEP_DMA();
    issue(t + NST - 1, sc == 0 ? NST - 1 : sc - 1);
    const int q0 = t * 32;
    if (q0 + 31 >= my_first) {
      const char* q_lds = smem + sc * STAGE;
      const char* do_lds = q_lds + ROW_TILE;      // DK only
      const char* t_lds = q_lds + T_OFF;
      const char* qp = q_lds + qr0 * 512;
      const char* dop = do_lds + qr0 * 512;
      const char* tp = t_lds + li * 64 + ((lq ^ tsw) << 4);
      bf16x8 fa[8], fb[8];
      f32x4 s[2], dp[2];
      dp[0] = dp[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (DK) {
        rd_row8(fa, qp, lq, sw0);                  // Q  queries tt=0
        rd_row8(fb, dop, lq, sw0);                 // dO queries tt=0
        MG_SCHED_FENCE();
        s[0] = mma8(fa, kf);
        MG_SCHED_FENCE();
        rd_row8(fa, qp + 4 * 512, lq, sw0);        // Q  tt=1
        MG_SCHED_FENCE();
        dp[0] = mma8(fb, vf);
        MG_SCHED_FENCE();
        rd_row8(fb, dop + 4 * 512, lq, sw0);       // dO tt=1
        MG_SCHED_FENCE();
        s[1] = mma8(fa, kf);
        MG_SCHED_FENCE();
        rd_t8(fa, tp);                             // Q^T d-tiles 0..7
        MG_SCHED_FENCE();
        dp[1] = mma8(fb, vf);
        MG_SCHED_FENCE();
      } else {
        rd_row8(fa, qp, lq, sw0);
        rd_row8(fb, qp + 4 * 512, lq, sw0);
        MG_SCHED_FENCE();
        s[0] = mma8(fa, kf);
        MG_SCHED_FENCE();
        rd_t8(fa, tp);                             // dO^T d-tiles 0..7
        MG_SCHED_FENCE();
        s[1] = mma8(fb, kf);
        MG_SCHED_FENCE();
      }
      // {-16 lse, -D} of this lane's 8 queries: 64 contiguous bytes.  Read by hand: for a compiler-visible
      // ds_read of the DMA-filled statistics hipcc inserts s_waitcnt vmcnt(0) and drains the ring.
      f32x4 l[4];
      asm volatile(
          "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
          "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
          : "v"(ls_addr + (uint32_t)(sc * STAGE))
          : "memory");
      // lane holds queries q0 + lq*8 + j (j = tt*4 + r) for its key
      // only the tiles that straddle this wave's keys (and the ragged last tile) need the mask
      if (q0 < my_first + 15 || q0 + 32 > S) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int qg = q0 + lq * 8 + j;
          s[j >> 2][j & 3] = (key > qg || qg >= S) ? -1e30f : s[j >> 2][j & 3];
        }
      }
      float val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float lse2 = -(l[j >> 1][(j & 1) * 2] * sc2), Dq = -l[j >> 1][(j & 1) * 2 + 1];
        const float p = __builtin_amdgcn_exp2f(fmaf(s[j >> 2][j & 3], sc2, -lse2));   // raw v_exp_f32; masked -> 0
        val[j] = DK ? p * (dp[j >> 2][j & 3] - Dq) * 0.0625f : p;
      }
      u32x4 pw;
#pragma unroll
      for (int j = 0; j < 4; ++j) pw[j] = pack2bf(val[2 * j], val[2 * j + 1]);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      MG_SCHED_FENCE();
      rd_t8(fb, tp + 8 * 1024);                    // d-tiles 8..15 land under the first 8 MFMAs
      MG_SCHED_FENCE();
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[dt], pf, acc[dt], 0, 0, 0);
      MG_SCHED_FENCE();
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) acc[8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[dt], pf, acc[8 + dt], 0, 0, 0);
    }
    sc = sc == NST - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);
  if (key < S) store_grad_row(gout, acc, b, h, H, S, key, lq);
}


}  // namespace

namespace {
int attn_bwd_launch(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt, const mg_bf16* kt,
                    const mg_bf16* dO, const mg_bf16* dOt, const mg_bf16* O, const float* lse, float* D,
                    const GradOut& gq, const GradOut& gk, const GradOut& gv, int32_t B, int32_t H, int32_t S, int32_t ld_t,
                    int64_t ld_o, bool make_dOt, hipStream_t s, const char* who) {
  if (ld_o < (int64_t)H * DH || (ld_o & 7)) MG_FAIL(MG_ERR_SHAPE, "%s: ld_o must be a multiple of 8 and >= H * 256", who);
  if (B <= 0 || H <= 0 || S <= 0 || (ld_t & 31) || ld_t < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "%s: ld_t must be a multiple of 32 and >= S", who);
  const void* ptrs[] = {q, k, v, qt, kt, dO, dOt, O, lse, D};
  for (const void* p : ptrs) {
    if (!p) MG_FAIL(MG_ERR_SHAPE, "%s: null pointer", who);
    if (!MG_ALIGNED16(p)) MG_FAIL(MG_ERR_ALIGN, "%s: pointers must be 16-byte aligned", who);
  }
  if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dq_kernel, DQ_STAGES * DQ_STAGE, who)) return rc;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv_kernel<true>, DK_STAGES * DK_STAGE, who)) return rc;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv_kernel<false>, DV_STAGES * DV_STAGE, who)) return rc;
  const int64_t rows = (int64_t)B * S * H;
  const dim3 grid((unsigned)(((S + 127) / 128) * B * H));
  const char* env_v = getenv("MAGMA_ATTN_BWD");        // read per call: tests and A/B scripts switch it in-process
  const int variant = env_v ? atoi(env_v) : 8;          // round 6: the tr-read kernels (attention_tr.hip) also behind this signature; qt / kt / dOt are then unused
  if (make_dOt && variant < 5)
    hipLaunchKernelGGL(attn_bwd_prep_t_kernel, dim3((S + 31) / 32, B * H), dim3(256), 0, s, dO, O, lse, D, (mg_bf16*)dOt, ld_t, B, H, S, ld_o);
  else
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dO, O, lse, D, B, H, S, ld_o);
  // MAGMA_ATTN_BWD: 0 = the three 16-row-wave kernels of rounds 1-4 (dQ | dV | dK: S computed three times, dP twice);
  // 1 / 2 = dQ as before + dK and dV in ONE 32-key-wave kernel (attention_bwd32.hip; 1: all LDS-DMA pieces of a tile at the
  // top of the step, 2: spread between the MFMA phases); 3 / 4 = the 32-query-wave dQ kernel as well (3: DMA at the top, 4: spread;
  // the dK/dV kernel then in its spread form).
  // 5 / 6 = the dK/dV kernel WITHOUT transposed images (attention_bwd32_tr.hip: ds_read_b64_tr_b16 from the row images; 5: two LDS
  // stages, 6: three); dO^T is then not made.
  // 7 / 8 = the dQ kernel without K^T as well (7: three LDS stages, 8: four), dK/dV as in 6.
  const AttnRows xr{q, k, v, (int64_t)H * S * DH, (int64_t)S * DH, DH};     // [B,H,S,256]
  if (variant >= 7) {
    if (int rc = attn_bwd_dq32_tr_launch(xr, dO, D, gq, B, H, S, variant == 8 ? 4 : 3, s, who)) return rc;
  } else if (variant >= 3) {
    if (int rc = attn_bwd_dq32_launch(q, k, v, kt, dO, D, gq, B, H, S, ld_t, variant, s, who)) return rc;
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(512), DQ_STAGES * DQ_STAGE, s, q, k, v, kt, dO, D, gq, B, H, S, ld_t);
  }
  if (variant >= 5) {
    MG_CHECK_LAUNCH();
    return attn_bwd_dkdv32_tr_launch(xr, dO, D, gk, gv, B, H, S, variant >= 6 ? 3 : 2, s, who);
  }
  if (variant >= 1) {
    MG_CHECK_LAUNCH();
    return attn_bwd_dkdv32_launch(q, k, v, qt, dO, dOt, D, gk, gv, B, H, S, ld_t, variant, s, who);
  }
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, grid, dim3(512), DV_STAGES * DV_STAGE, s, q, k, v, qt, dO, dOt, D, gv, B, H, S, ld_t);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, grid, dim3(512), DK_STAGES * DK_STAGE, s, q, k, v, qt, dO, dOt, D, gk, B, H, S, ld_t);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
}  // namespace

// q,k,v [B,H,S,256]; kt,qt,dOt column-tiled transposed [B,H,ld_t/32,256,32] (mg_head_transpose_bf16; zero padded);
// dO [B*S, H*256]; O [B*S, >= H*256] with row stride ld_o (elements); lse [B,H,S]; D [B,H,S,2] workspace; dq,dk,dv [B,H,S,256]
extern "C" int mg_attn_bwd_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                                const mg_bf16* kt, const mg_bf16* dO, const mg_bf16* dOt, const mg_bf16* O,
                                const float* lse, float* D, mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, int32_t B,
                                int32_t H, int32_t S, int32_t ld_t, int64_t ld_o, void* stream) {
  if (!dq || !dk || !dv) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_bf16: null pointer");
  if (!MG_ALIGNED16(dq) || !MG_ALIGNED16(dk) || !MG_ALIGNED16(dv)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_bwd_bf16: pointers must be 16-byte aligned");
  const GradOut gq{dq, nullptr, nullptr, nullptr, 0, 0}, gk{dk, nullptr, nullptr, nullptr, 1, 0}, gv{dv, nullptr, nullptr, nullptr, 2, 0};
  return attn_bwd_launch(q, k, v, qt, kt, dO, dOt, O, lse, D, gq, gk, gv, B, H, S, ld_t, ld_o, false, (hipStream_t)stream, "mg_attn_bwd_bf16");
}

// Same backward, written straight into the gradient of the fused qkv projection: dqkv [B*S, 3*H*256] = [dq | dk | dv]
// per row, with the inverse GPT-J rotary (angles of position s, tables sin_t / cos_t [>= S, rot_dim/2]) applied to the
// first rot_dim columns of every dq and dk head -- mg_attn_bwd_bf16 followed by mg_rotary_merge_bwd_bf16 in one pass.
// dOt is a WORKSPACE here ([B,H,ld_t/32,256,32]): the first launch transposes dO into it while it computes D.
extern "C" int mg_attn_bwd_merged_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                                       const mg_bf16* kt, const mg_bf16* dO, mg_bf16* dOt, const mg_bf16* O,
                                       const float* lse, float* D, mg_bf16* dqkv, int32_t rot_dim, const float* sin_t,
                                       const float* cos_t, int32_t B, int32_t H, int32_t S, int32_t ld_t, int64_t ld_o, void* stream) {
  if (!dqkv || !MG_ALIGNED16(dqkv)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_bwd_merged_bf16: dqkv must be a 16-byte aligned pointer");
  if (rot_dim < 0 || rot_dim > 256 || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_merged_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (rot_dim && (!sin_t || !cos_t)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_merged_bf16: rotary tables missing");
  const GradOut gq{nullptr, dqkv, sin_t, cos_t, 0, rot_dim}, gk{nullptr, dqkv, sin_t, cos_t, 1, rot_dim}, gv{nullptr, dqkv, sin_t, cos_t, 2, 0};
  return attn_bwd_launch(q, k, v, qt, kt, dO, dOt, O, lse, D, gq, gk, gv, B, H, S, ld_t, ld_o, true, (hipStream_t)stream, "mg_attn_bwd_merged_bf16");
}
