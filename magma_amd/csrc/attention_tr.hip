// attention_tr.hip -- the 32-row-wave attention kernels (forward, dQ, dK/dV) WITHOUT transposed operand images (gfx950, round 6).
//
// attention_bwd32.hip feeds the s-contractions (dV^T += dO^T P, dK^T += Q^T dS; dQ^T += K^T dS^T) from transposed copies of the
// operands: Q^T / dO^T / K^T exist in HBM (written by rotary_split_kernel<true> and attn_bwd_prep_t_kernel), stream through L2 next
// to the row images and take half of the LDS-DMA pieces a wave issues per tile step (17, of ~76 issue cycles each; the merged
// dK/dV kernel ran at an L2 hit rate of 0.27 with 4.4 x fabric re-fetch, profiles/r05_attention_pmc_summary.txt).  Here the
// s-contraction fragments are read from the ROW images with ds_read_b64_tr_b16 (two per 32x32x16 A fragment): the hardware
// transpose delivers, to lane c of a 16-lane group, column c of a [4 rows][16 columns] block whose rows the group's lanes address
// four to a row.  One row image therefore serves both contractions:
//
//   image   [32 rows][256 d] bf16, 512-byte rows, 16-byte chunk c of row r at position c ^ tr_swz(r),
//           tr_swz(r) = ((r & 3) << 2) | ((r >> 2) & 3)
//   (R)     d-contraction fragment (ds_read_b128): lane (l31, hi) reads chunk 2 ks + hi of row perm32(l31).  A ds_read_b128 lane
//           group covers rows whose r >> 2 is {0,3,6,5} or {2,1,4,7} (and r & 3 = 0..3): 16 distinct tr_swz values = 16 bank quads.
//   (T)     s-contraction fragment (2 x ds_read_b64_tr_b16): lane (i = lane & 15, g16 = (lane >> 4) & 1, hi) reads 8 bytes of row
//           ks 16 + hi 8 + 4 j + (i >> 2) at column db 32 + g16 16 + (i & 3) 4: a 32-lane half touches four rows x 64 bytes, and
//           the four rows (r & 3 = 0..3) sit in four different 64-byte bank spans.
//   Both measured conflict-free (SQ_LDS_BANK_CONFLICT = 0) -- and the round-5 row_swz 4-way conflicted for (T) --
//   in profiles/r06_tr_bank_probe.txt (tools/probes/tr_bank_probe.hip).
//
// Per tile step a wave issues 9 LDS-DMA pieces instead of 17 and the stage is 32.5 KiB instead of 64.5; the HBM/L2 footprint of
// a head halves.  EVERY LDS fragment read of the tile loop is an asm statement and EVERY lgkmcnt wait is written by hand (the
// tr-read has no builtin hipcc's wait-count pass would know): the counts are stated at each wait.
#include "attn_bwd_device.h"
#include "attn32_device.h"
#include <stdlib.h>

namespace {

MG_DEV int tr_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

#define MG_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

// (R) four fragments ks = g*4 .. g*4+3 of the row this lane feeds; `addr` = LDS byte address of the image + row_base32(R, tr_swz(R), hi)
MG_DEV void rd_row4_asm(bf16x8 (&f)[4], uint32_t addr, int g) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = addr ^ (uint32_t)((g * 4 + i) << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(f[i]) : "v"(a));
  }
}
// (T) lane constant: row hi 8 + (i >> 2), chunk ((i >> 2) << 2 | (g16 ^ hi) << 1 | (i >> 1) & 1), byte (i & 1) 8.  The fragment
// (db, ks, j) is at  (base ^ ((db & 3) << 6 | j << 4)) + ((db >> 2) << 8 | j << 11 | ks << 13)  -- XOR where the lane constant has
// bits of its own (tr_swz of the row: its r & 3 part meets db, its (r >> 2) & 3 = hi << 1 | j part meets j), an immediate elsewhere.
MG_DEV uint32_t tr_lane_base(int lane) {
  const int i = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5;
  return (uint32_t)((hi * 8 + (i >> 2)) * 512 + (((((i >> 2) & 3) << 2) | ((g16 ^ hi) << 1) | ((i >> 1) & 1)) << 4) + (i & 1) * 8);
}
template <int OFF>
MG_DEV void rd_tr(u32x2& f, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
// half-burst N (0..7) of an image at byte offset IMG inside the stage: d-blocks 2 (N >> 1), 2 (N >> 1) + 1 at k-step N & 1 --
// four tr-reads, two fragments
template <int IMG, int N>
MG_DEV void rd_t_half(bf16x8& f0, bf16x8& f1, uint32_t tb) {
  constexpr int KS = N & 1, DB0 = 2 * (N >> 1), DB1 = DB0 + 1;
  u32x2 a0, a1, b0, b1;
  rd_tr<IMG + ((DB0 >> 2) << 8) + (KS << 13)>(a0, tb ^ (uint32_t)((DB0 & 3) << 6));
  rd_tr<IMG + ((DB0 >> 2) << 8) + (1 << 11) + (KS << 13)>(a1, tb ^ (uint32_t)(((DB0 & 3) << 6) | 16));
  rd_tr<IMG + ((DB1 >> 2) << 8) + (KS << 13)>(b0, tb ^ (uint32_t)((DB1 & 3) << 6));
  rd_tr<IMG + ((DB1 >> 2) << 8) + (1 << 11) + (KS << 13)>(b1, tb ^ (uint32_t)(((DB1 & 3) << 6) | 16));
  f0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3));
  f1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3));
}

// A wave's 32 x 256 gradient tile -> bf16 through a wave-private LDS image, out as whole 512-byte rows (as attention_bwd32.hip)
MG_DEV void store_grad_tile32_tr(const GradOut& g, const f32x16 (&acc)[8], float scale, char* stage, int b, int h, int H, int S,
                                 int row0, int l31, int hi) {
  const bool mrg = g.merged || g.q8;
  const bool rot = mrg && g.which < 2 && g.rot_dim > 0;
  const int half_rot = g.rot_dim >> 1;
  const int s_me = min(row0 + l31, S - 1);
  char* wr = stage + l31 * EP_ROW + hi * 8;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float x0 = acc[db][rq * 4] * scale, x1 = acc[db][rq * 4 + 1] * scale, x2 = acc[db][rq * 4 + 2] * scale, x3 = acc[db][rq * 4 + 3] * scale;
      const int d = db * 32 + rq * 8 + hi * 4;
      if (rot && d < g.rot_dim) {
        const int pi = (int)((int64_t)s_me * half_rot) + (d >> 1);
        const float s0 = g.sin_t[pi], c0 = g.cos_t[pi], s1 = g.sin_t[pi + 1], c1 = g.cos_t[pi + 1];
        const float y0 = x0 * c0 + x1 * s0, y1 = x1 * c0 - x0 * s0;
        const float y2 = x2 * c1 + x3 * s1, y3 = x3 * c1 - x2 * s1;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
      }
      const u32x2 w = {pack2bf(x0, x1), pack2bf(x2, x3)};
      *(u32x2*)(wr + db * 64 + rq * 16) = w;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 2 + hi;
    const u32x4 w = *(const u32x4*)(stage + row * EP_ROW + l31 * 16);
    const int s = row0 + row;
    if (s < S) {
      if (g.merged || g.out) *(u32x4*)(grad_row_ptr(g, b, h, H, S, s) + l31 * 8) = w;
      if (g.q8) {      // the MX e4m3 copy of this row piece of dqkv (GradOut): four consecutive lanes hold one 32-column block
        const int64_t grow = (int64_t)b * S + s;
        mx_emit8(w, g.q8 + grow * (3 * H * DH), g.q8_scales, (g.mx_rows + 63) >> 6, (int)grow, g.which * H * DH + h * DH + l31 * 8);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

constexpr int TRKV_STAGE = 2 * ROW_TILE + 512;     // Q rows | dO rows | {-16 lse x 32, -D x 32} (256 B) | pad: a multiple of 512
static_assert(TRKV_STAGE % 512 == 0, "the XOR addressing needs stage offsets that are multiples of 512");
constexpr int TRKV_LD_OFF = 2 * ROW_TILE;
constexpr int TRKV_EPILOGUE = 4 * 32 * EP_ROW;     // the ring becomes the four waves' staging images

// ---------------------------------------------------------------------------
// dK and dV of one (b, h, 128 keys): 4 waves x 32 keys (lane & 31 = key), one wave per SIMD, query tiles of 32 from the diagonal down.
//   S' = Q K^T - 16 lse      dP' = dO V^T - D      P = exp2(S' log2e / 16)      dS = P o dP'
//   dV^T += dO^T P           dK^T += Q^T dS        (x 1/16 in the epilogue)
// NST stages of (Q rows | dO rows | statistics); the tile t + NST - 1 is in flight while tile t is multiplied.
template <int NST>
__global__ __launch_bounds__(256) void attn_bwd_dkdv32_tr_kernel(
    const AttnRows x, const mg_bf16* __restrict__ dO, const float* __restrict__ ld2, const GradOut gk, const GradOut gv, int B, int H, int S,
    int nblk) {       // nblk: key blocks of 128 per (b, h) this launch covers -- ceil(S / 128), or fewer (first_rows: only the first keys' gradients)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int k0 = (wg - bh * nblk) * 128;       // earliest key blocks (most query tiles) first
  const int key = k0 + wave * 32 + l31, key_c = min(key, S - 1);
  const int dmodel = H * DH;
  const int64_t xoff = (int64_t)b * x.stride_b + (int64_t)h * x.stride_h;     // this head's rows: x.q/k/v + xoff + position * x.ld
  const mg_bf16* qb = x.q + xoff;
  const uint32_t ldb = (uint32_t)x.ld * 2u;                                    // row stride in bytes
  const mg_bf16* dob = dO + (int64_t)b * S * dmodel + h * DH;   // row stride dmodel
  const float* stb = ld2 + (int64_t)bh * S * 2;

  const int t_begin = k0 >> 5;                 // first query tile that can see key k0
  const int t_end = (S + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);       // 0: the kernel has no static LDS (the XOR addressing relies on 512-byte alignment)
  // piece i (0..3) of the two 16-KiB images of tile t; wave w moves 1-KiB blocks 4w .. 4w+3 of an image = tile rows 8w + 2i + hi.
  // tr_swz(8w + 2i + hi) = (i & 1) << 3 | hi << 2 | (w & 1) << 1 | i >> 1: one lane constant, the piece enters through two XOR bits.
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ ((hi << 2) | ((wave & 1) << 1))) << 4);
  const uint32_t do_stride = (uint32_t)dmodel * 2u;
  auto issue_part = [&](int t, int buf, int i) {
    const int q0 = t * 32;
    const uint32_t st = smem_u + (uint32_t)(buf * TRKV_STAGE + (wave * 4 + i) * 1024);
    const uint32_t r = (uint32_t)min(q0 + row0 + 2 * i, S - 1);
    const uint32_t cb = c0b ^ (uint32_t)((((i & 1) << 3) | (i >> 1)) << 4);
    glds16su(qb, r * ldb + cb, st);
    glds16su(dob, r * do_stride + cb, st + ROW_TILE);
    if (i == 0)   // statistics as two arrays: lanes 0-31 fetch -16 lse of query q0 + l31, lanes 32-63 its -D (every wave writes the same 256 B)
      glds4su(stb, (uint32_t)(min(q0 + l31, S - 1) * 8 + hi * 4), smem_u + (uint32_t)(buf * TRKV_STAGE + TRKV_LD_OFF));
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
  // prologue: NST - 1 tiles in flight (tiles past the last one are not issued; the vmcnt waits below count what WAS issued)
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (t_begin + j < t_end) issue(t_begin + j, j);

  bf16x8 kf[16], vf[16];
  {
    const mg_bf16* kp = x.k + xoff + (int64_t)key_c * x.ld + hi * 8;
    const mg_bf16* vp = x.v + xoff + (int64_t)key_c * x.ld + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 16); vf[ks] = *(const bf16x8*)(vp + ks * 16); }
  }
  f32x16 acck[8], accv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { acck[i][j] = 0.f; accv[i][j] = 0.f; }
  }
  const float sc2 = 0.0625f * 1.4426950408889634f;
  const int my_first = k0 + wave * 32;         // query tiles that end before this wave's first key are fully masked
  const int R = perm32(l31);                   // tile row that feeds this lane's MFMA row
  const uint32_t rb = smem_u + row_base32(R, tr_swz(R), hi);
  const uint32_t tb0 = smem_u + tr_lane_base(lane);
  const uint32_t ls_addr = smem_u + (uint32_t)(TRKV_LD_OFF + hi * 32);   // + g*64 (+128 for D): 8 consecutive queries = 32 bytes

  MG_USE8(kf); MG_USE8(vf);                    // retire the ordinary loads in hipcc's scoreboard (see attention.hip)
  {
    bf16x8* k8 = kf + 8; bf16x8* v8 = vf + 8;
    asm volatile("" ::"v"(k8[0]), "v"(k8[1]), "v"(k8[2]), "v"(k8[3]), "v"(k8[4]), "v"(k8[5]), "v"(k8[6]), "v"(k8[7]));
    asm volatile("" ::"v"(v8[0]), "v"(v8[1]), "v"(v8[2]), "v"(v8[3]), "v"(v8[4]), "v"(v8[5]), "v"(v8[6]), "v"(v8[7]));
  }
  int sc = 0;
  int t = t_begin;
  // Each wave issues 9 pieces per tile (8 + the statistics); with NST - 1 tiles in flight, "tile t has landed" = at most the
  // (NST - 2) x 9 newer pieces outstanding.  NST = 2: vmcnt(0).  NST = 3: vmcnt(9) -- except at the tail, where fewer tiles were
  // issued behind tile t and the wait has to be vmcnt(0).
  auto wait_tile = [&](int tt) {
    if constexpr (NST == 2) { MG_WAIT_VMCNT(0); }
    else { if (tt + 1 < t_end) MG_WAIT_VMCNT(9); else MG_WAIT_VMCNT(0); }
  };
  // query tiles that end before this wave's first key are fully masked for it (wave w: the first w tiles of the block): it
  // only moves its share of the data.  A loop of its own -- with the accumulators updated under a branch hipcc copies all
  // 256 of them around the control flow.
  for (const int t_act = min(t_begin + wave, t_end); t < t_act; ++t) {
    wait_tile(t);
    MG_BARRIER_KEEP_DMA();
    if (t + NST - 1 < t_end) issue(t + NST - 1, sc == 0 ? NST - 1 : sc - 1);
    sc = sc == NST - 1 ? 0 : sc + 1;
  }
  for (; t < t_end; ++t) {
    wait_tile(t);                                // this wave's pieces of tile t have landed
    MG_BARRIER_KEEP_DMA();                       // tile t complete; everyone is done with tile t-1 (lgkmcnt = 0 here)
    const bool more = t + NST - 1 < t_end;
    const int nbuf = sc == 0 ? NST - 1 : sc - 1; // the stage tile t-1 was in
    if (more) issue_part(t + NST - 1, nbuf, 0);
    const int q0 = t * 32;
    {
      const uint32_t stoff = (uint32_t)(sc * TRKV_STAGE);
      const uint32_t qrow = stoff + rb, dorow = qrow + ROW_TILE;
      const uint32_t tb = stoff + tb0;
      const uint32_t lsa = ls_addr + stoff;
      bf16x8 fa[4], fb[4];
      // The statistics are the INITIAL VALUES of the two score accumulators (attention_bwd32.hip): S' = Q K^T - 16 lse, dP' = dO V^T - D.
      f32x4 i0, i1, i2, i3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\t"
                   "ds_read_b128 %3, %4 offset:80"
                   : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lsa) : "memory");
      // ---- phase 1: S' = Q K^T - 16 lse (16 MFMAs), (R) bursts of four one burst ahead ----
      rd_row4_asm(fa, qrow, 0);
      rd_row4_asm(fb, qrow, 1);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // outstanding: statistics 4, Q0 4, Q1 4 -> Q1 may fly
      f32x16 s = __builtin_shufflevector(__builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7),
                                         __builtin_shufflevector(i2, i3, 0, 1, 2, 3, 4, 5, 6, 7),
                                         0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fa[i], kf[i]);
      rd_row4_asm(fa, qrow, 2);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // Q1 | Q2
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fb[i], kf[4 + i]);
      rd_row4_asm(fb, qrow, 3);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // Q2 | Q3
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fa[i], kf[8 + i]);
      asm volatile("ds_read_b128 %0, %4 offset:128\n\tds_read_b128 %1, %4 offset:144\n\tds_read_b128 %2, %4 offset:192\n\t"
                   "ds_read_b128 %3, %4 offset:208"
                   : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lsa) : "memory");
      rd_row4_asm(fa, dorow, 0);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // Q3 | D 4, dO0 4
#pragma unroll
      for (int i = 0; i < 3; ++i) mfma32v(s, fb[i], kf[12 + i]);
      mfma32v_last(s, fb[3], kf[15]);
      rd_row4_asm(fb, dorow, 1);
      MG_SCHED_FENCE();
      if (more) issue_part(t + NST - 1, nbuf, 1);
      // ---- phase 2: dP' = dO V^T - D (16 MFMAs) beside P = exp2(S' sc2) ----
      MG_LGKM(4);                       // D, dO0 | dO1
      f32x16 dp = __builtin_shufflevector(__builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7),
                                          __builtin_shufflevector(i2, i3, 0, 1, 2, 3, 4, 5, 6, 7),
                                          0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(dp, fa[i], vf[i]);
      rd_row4_asm(fa, dorow, 2);
      // only the tiles that straddle this wave's keys (and the ragged last tile) need the mask
      if (q0 < my_first + 31 || q0 + 32 > S) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qg = q0 + (r >> 3) * 16 + hi * 8 + (r & 7);
          s[r] = (key > qg || qg >= S) ? -1e30f : s[r];
        }
      }
      float p[16];
      u32x4 pw0, pw1;
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // dO1 | dO2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fb[i], vf[4 + i]);
        MG_SCHED_FENCE();
        p[2 * i] = __builtin_amdgcn_exp2f(s[2 * i] * sc2);               // raw v_exp_f32; masked -> 0
        p[2 * i + 1] = __builtin_amdgcn_exp2f(s[2 * i + 1] * sc2);
        MG_SCHED_FENCE();
      }
      rd_row4_asm(fb, dorow, 3);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // dO2 | dO3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fa[i], vf[8 + i]);
        MG_SCHED_FENCE();
        p[8 + 2 * i] = __builtin_amdgcn_exp2f(s[8 + 2 * i] * sc2);
        p[9 + 2 * i] = __builtin_amdgcn_exp2f(s[9 + 2 * i] * sc2);
        MG_SCHED_FENCE();
      }
      // (T) half-bursts through a ring of four half-buffers {fa[0..1], fa[2..3], fb[0..1], fb[2..3]}: half n of the 16 (8 of
      // dO^T, then 8 of Q^T) sits in slot n & 3; three halves fly while one is multiplied (12 reads: lgkmcnt counts to 15)
      rd_t_half<ROW_TILE, 0>(fa[0], fa[1], tb);
      rd_t_half<ROW_TILE, 1>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // dO3 | T0 4, T1 4
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < 3) mfma32v(dp, fb[i], vf[12 + i]); else mfma32v_last(dp, fb[3], vf[15]);
        MG_SCHED_FENCE();
        pw0[i] = pack2bf(p[2 * i], p[2 * i + 1]);
        pw1[i] = pack2bf(p[8 + 2 * i], p[9 + 2 * i]);
        MG_SCHED_FENCE();
      }
      rd_t_half<ROW_TILE, 2>(fb[0], fb[1], tb);
      bf16x8 pf0 = __builtin_bit_cast(bf16x8, pw0), pf1 = __builtin_bit_cast(bf16x8, pw1);
      mfma_operand_ready(pf0, pf1);
      MG_SCHED_FENCE();
      if (more) issue_part(t + NST - 1, nbuf, 2);
      // ---- phase 3: dV^T += dO^T P (16 MFMAs = 8 halves) beside 16 dS = P o dP' ----
      const u32x4 q0w = __builtin_bit_cast(u32x4, pf0), q1w = __builtin_bit_cast(u32x4, pf1);
      float ds[16];
      u32x4 dw0, dw1;
      // half 0: d-blocks 0, 1 at k-step 0
      MG_LGKM(8);                       // T0 | T1, T2
      mfma32a(accv[0], fa[0], pf0); mfma32a(accv[1], fa[1], pf0);
      rd_t_half<ROW_TILE, 3>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T1 | T2, T3
      mfma32a(accv[0], fa[2], pf1); mfma32a(accv[1], fa[3], pf1);
      rd_t_half<ROW_TILE, 4>(fa[0], fa[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T2 | T3, T4
      mfma32a(accv[2], fb[0], pf0);
      MG_SCHED_FENCE();
      ds[0] = bflo(q0w[0]) * dp[0]; ds[1] = bfhi(q0w[0]) * dp[1];
      MG_SCHED_FENCE();
      mfma32a(accv[3], fb[1], pf0);
      MG_SCHED_FENCE();
      ds[2] = bflo(q0w[1]) * dp[2]; ds[3] = bfhi(q0w[1]) * dp[3];
      rd_t_half<ROW_TILE, 5>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T3 | T4, T5
      mfma32a(accv[2], fb[2], pf1);
      MG_SCHED_FENCE();
      ds[4] = bflo(q0w[2]) * dp[4]; ds[5] = bfhi(q0w[2]) * dp[5];
      MG_SCHED_FENCE();
      mfma32a(accv[3], fb[3], pf1);
      MG_SCHED_FENCE();
      ds[6] = bflo(q0w[3]) * dp[6]; ds[7] = bfhi(q0w[3]) * dp[7];
      rd_t_half<ROW_TILE, 6>(fb[0], fb[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T4 | T5, T6
      mfma32a(accv[4], fa[0], pf0);
      MG_SCHED_FENCE();
      ds[8] = bflo(q1w[0]) * dp[8]; ds[9] = bfhi(q1w[0]) * dp[9];
      MG_SCHED_FENCE();
      mfma32a(accv[5], fa[1], pf0);
      MG_SCHED_FENCE();
      ds[10] = bflo(q1w[1]) * dp[10]; ds[11] = bfhi(q1w[1]) * dp[11];
      rd_t_half<ROW_TILE, 7>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T5 | T6, T7
      mfma32a(accv[4], fa[2], pf1);
      MG_SCHED_FENCE();
      ds[12] = bflo(q1w[2]) * dp[12]; ds[13] = bfhi(q1w[2]) * dp[13];
      MG_SCHED_FENCE();
      mfma32a(accv[5], fa[3], pf1);
      MG_SCHED_FENCE();
      ds[14] = bflo(q1w[3]) * dp[14]; ds[15] = bfhi(q1w[3]) * dp[15];
      rd_t_half<0, 0>(fa[0], fa[1], tb);            // Q^T half 0
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T6 | T7, U0
      mfma32a(accv[6], fb[0], pf0);
      MG_SCHED_FENCE();
      dw0[0] = pack2bf(ds[0], ds[1]); dw0[1] = pack2bf(ds[2], ds[3]); dw1[0] = pack2bf(ds[8], ds[9]); dw1[1] = pack2bf(ds[10], ds[11]);
      MG_SCHED_FENCE();
      mfma32a(accv[7], fb[1], pf0);
      rd_t_half<0, 1>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T7 | U0, U1
      mfma32a(accv[6], fb[2], pf1);
      MG_SCHED_FENCE();
      dw0[2] = pack2bf(ds[4], ds[5]); dw0[3] = pack2bf(ds[6], ds[7]); dw1[2] = pack2bf(ds[12], ds[13]); dw1[3] = pack2bf(ds[14], ds[15]);
      MG_SCHED_FENCE();
      mfma32a_last(accv[7], fb[3], pf1);
      rd_t_half<0, 2>(fb[0], fb[1], tb);
      bf16x8 df0 = __builtin_bit_cast(bf16x8, dw0), df1 = __builtin_bit_cast(bf16x8, dw1);
      mfma_operand_ready(df0, df1);
      MG_SCHED_FENCE();
      if (more) issue_part(t + NST - 1, nbuf, 3);
      // ---- phase 4: 16 dK^T += Q^T (16 dS) (16 MFMAs = 8 halves) ----
      MG_LGKM(8);                       // U0 | U1, U2
      mfma32a(acck[0], fa[0], df0); mfma32a(acck[1], fa[1], df0);
      rd_t_half<0, 3>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // U1 | U2, U3
      mfma32a(acck[0], fa[2], df1); mfma32a(acck[1], fa[3], df1);
      rd_t_half<0, 4>(fa[0], fa[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // U2 | U3, U4
      mfma32a(acck[2], fb[0], df0); mfma32a(acck[3], fb[1], df0);
      rd_t_half<0, 5>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // U3 | U4, U5
      mfma32a(acck[2], fb[2], df1); mfma32a(acck[3], fb[3], df1);
      rd_t_half<0, 6>(fb[0], fb[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // U4 | U5, U6
      mfma32a(acck[4], fa[0], df0); mfma32a(acck[5], fa[1], df0);
      rd_t_half<0, 7>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // U5 | U6, U7
      mfma32a(acck[4], fa[2], df1); mfma32a(acck[5], fa[3], df1);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // U6 | U7
      mfma32a(acck[6], fb[0], df0); mfma32a(acck[7], fb[1], df0);
      MG_SCHED_FENCE();
      MG_LGKM(0);                       // U7
      mfma32a(acck[6], fb[2], df1);
      mfma32a_last(acck[7], fb[3], df1);
    }
    sc = sc == NST - 1 ? 0 : sc + 1;
  }
  // every wave is done with the ring (and no DMA is in flight: the last tiles issue none) before it becomes staging space
  MG_BARRIER_KEEP_DMA();
  char* stage = smem + wave * (32 * EP_ROW);
  store_grad_tile32_tr(gv, accv, 1.0f, stage, b, h, H, S, k0 + wave * 32, l31, hi);
  store_grad_tile32_tr(gk, acck, 0.0625f, stage, b, h, H, S, k0 + wave * 32, l31, hi);
}


constexpr int TRQ_STAGE = 2 * ROW_TILE;            // K rows | V rows

// ---------------------------------------------------------------------------
// dQ of 128 queries: 4 waves x 32 queries (lane & 31 = query), KV tiles of 32 through a ring of NST stages (NST - 1 tiles in flight).
// Transposed frame as in attn_bwd_dq32_kernel: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T -- K^T read from the K ROW image
// with ds_read_b64_tr_b16: 8 LDS-DMA pieces per tile step instead of 12, no K^T operand.
template <int NST>
__global__ __launch_bounds__(256) void attn_bwd_dq32_tr_kernel(
    const AttnRows x, const mg_bf16* __restrict__ dO, const float* __restrict__ ld2, const GradOut gq, int B, int H, int S,
    int nblk) {       // nblk: query blocks of 128 per (b, h) this launch covers (the FIRST ones; see attn_bwd_dkdv32_tr_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;            // longest (latest) query blocks first
  const int qrow = qt0 + wave * 32 + l31, qrow_c = min(qrow, S - 1);
  const int64_t xoff = (int64_t)b * x.stride_b + (int64_t)h * x.stride_h;
  const mg_bf16* kb = x.k + xoff;
  const mg_bf16* vb = x.v + xoff;
  const uint32_t ldb = (uint32_t)x.ld * 2u;
  const int dmodel = H * DH;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ ((hi << 2) | ((wave & 1) << 1))) << 4);
  auto issue_part = [&](int t, int buf, int i) {          // piece i of the two images of tile min(t, last)
    const int tc = min(t, ntiles - 1);
    const uint32_t st = smem_u + (uint32_t)(buf * TRQ_STAGE + (wave * 4 + i) * 1024);
    const uint32_t off = (uint32_t)min(tc * 32 + row0 + 2 * i, S - 1) * ldb + (c0b ^ (uint32_t)((((i & 1) << 3) | (i >> 1)) << 4));
    glds16su(kb, off, st);
    glds16su(vb, off, st + ROW_TILE);
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) issue(j, j);           // past the last tile the ring re-loads the last tile: constant wait counts

  bf16x8 qf[16], dof[16];
  {
    const mg_bf16* qp = x.q + xoff + (int64_t)qrow_c * x.ld + hi * 8;
    const mg_bf16* dp_ = dO + (int64_t)(b * S + qrow_c) * dmodel + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 16); dof[ks] = *(const bf16x8*)(dp_ + ks * 16); }
  }
  const float sc2 = 0.0625f * 1.4426950408889634f;
  const float nl2 = ld2[((int64_t)bh * S + qrow_c) * 2] * sc2;      // -lse log2 e
  const float Dn = ld2[((int64_t)bh * S + qrow_c) * 2 + 1];         // -D
  f32x16 accq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) accq[i][j] = 0.f;
  }
  const int R = perm32(l31);
  const uint32_t rb = smem_u + row_base32(R, tr_swz(R), hi);
  const uint32_t tb0 = smem_u + tr_lane_base(lane);
  // this wave's tiles: 0 .. n_act-1 (key tiles past its last query are fully masked for it)
  const int n_act = min(ntiles, ((qt0 + wave * 32 + 31) >> 5) + 1);

  MG_USE8(qf); MG_USE8(dof);
  {
    bf16x8* a8 = qf + 8; bf16x8* b8 = dof + 8;
    asm volatile("" ::"v"(a8[0]), "v"(a8[1]), "v"(a8[2]), "v"(a8[3]), "v"(a8[4]), "v"(a8[5]), "v"(a8[6]), "v"(a8[7]));
    asm volatile("" ::"v"(b8[0]), "v"(b8[1]), "v"(b8[2]), "v"(b8[3]), "v"(b8[4]), "v"(b8[5]), "v"(b8[6]), "v"(b8[7]));
  }
  asm volatile("" ::"v"(nl2), "v"(Dn));
  int sc = 0;
  int t = 0;
  for (; t < n_act; ++t) {
    // this wave's pieces of tile t have landed (the NST - 2 tiles behind it, 8 pieces each, may be in flight)
    if constexpr (NST == 3) MG_WAIT_VMCNT(8); else MG_WAIT_VMCNT(16);
    MG_BARRIER_KEEP_DMA();            // tile t complete; everyone is done with tile t-1
    const int nb = sc == 0 ? NST - 1 : sc - 1;
    issue_part(t + NST - 1, nb, 0);
    const int kv0 = t * 32;
    {
      const uint32_t stoff = (uint32_t)(sc * TRQ_STAGE);
      const uint32_t krow = stoff + rb, vrow = krow + ROW_TILE;
      const uint32_t tb = stoff + tb0;
      bf16x8 fa[4], fb[4];
      f32x16 s, dp;
      // ---- phase 1: S^T = K Q^T ----
      rd_row4_asm(fa, krow, 0);
      rd_row4_asm(fb, krow, 1);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // K0 | K1
      mfma32v0(s, fa[0], qf[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) mfma32v(s, fa[i], qf[i]);
      rd_row4_asm(fa, krow, 2);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // K1 | K2
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fb[i], qf[4 + i]);
      rd_row4_asm(fb, krow, 3);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // K2 | K3
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fa[i], qf[8 + i]);
      rd_row4_asm(fa, vrow, 0);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // K3 | V0
#pragma unroll
      for (int i = 0; i < 3; ++i) mfma32v(s, fb[i], qf[12 + i]);
      mfma32v_last(s, fb[3], qf[15]);
      rd_row4_asm(fb, vrow, 1);
      MG_SCHED_FENCE();
      issue_part(t + NST - 1, nb, 1);
      // ---- phase 2: dP^T = V dO^T beside P^T = exp2(S^T sc2 - lse2) ----
      MG_LGKM(4);                       // V0 | V1
      mfma32v0(dp, fa[0], dof[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) mfma32v(dp, fa[i], dof[i]);
      rd_row4_asm(fa, vrow, 2);
      // only this wave's diagonal tile (and the ragged last tile) needs the mask: -1e30 -> exp2(-huge) = 0
      if (kv0 + 31 > qt0 + wave * 32 || kv0 + 32 > S) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + (r >> 3) * 16 + hi * 8 + (r & 7);
          s[r] = (key > qrow || key >= S) ? -1e30f : s[r];
        }
      }
      float p[16];
      u32x4 pw0, pw1;
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // V1 | V2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fb[i], dof[4 + i]);
        MG_SCHED_FENCE();
        p[2 * i] = __builtin_amdgcn_exp2f(fmaf(s[2 * i], sc2, nl2));
        p[2 * i + 1] = __builtin_amdgcn_exp2f(fmaf(s[2 * i + 1], sc2, nl2));
        MG_SCHED_FENCE();
      }
      rd_row4_asm(fb, vrow, 3);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // V2 | V3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fa[i], dof[8 + i]);
        MG_SCHED_FENCE();
        p[8 + 2 * i] = __builtin_amdgcn_exp2f(fmaf(s[8 + 2 * i], sc2, nl2));
        p[9 + 2 * i] = __builtin_amdgcn_exp2f(fmaf(s[9 + 2 * i], sc2, nl2));
        MG_SCHED_FENCE();
      }
      rd_t_half<0, 0>(fa[0], fa[1], tb);     // K^T halves through the ring {fa[0..1], fa[2..3], fb[0..1], fb[2..3]}
      rd_t_half<0, 1>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // V3 | T0, T1
#pragma unroll
      for (int i = 0; i < 3; ++i) mfma32v(dp, fb[i], dof[12 + i]);
      mfma32v_last(dp, fb[3], dof[15]);
      rd_t_half<0, 2>(fb[0], fb[1], tb);
      MG_SCHED_FENCE();
      issue_part(t + NST - 1, nb, 2);
      // 16 dS^T = P^T o (dP^T - D) (the 1/16 is applied once, in the epilogue)
      {
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] + Dn);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pw0[j] = pack2bf(ds[2 * j], ds[2 * j + 1]); pw1[j] = pack2bf(ds[8 + 2 * j], ds[9 + 2 * j]); }
      }
      bf16x8 df0 = __builtin_bit_cast(bf16x8, pw0), df1 = __builtin_bit_cast(bf16x8, pw1);
      mfma_operand_ready(df0, df1);
      MG_SCHED_FENCE();
      // ---- phase 3: 16 dQ^T += K^T (16 dS^T): 8 halves ----
      MG_LGKM(8);                       // T0 | T1, T2
      mfma32a(accq[0], fa[0], df0); mfma32a(accq[1], fa[1], df0);
      rd_t_half<0, 3>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T1 | T2, T3
      mfma32a(accq[0], fa[2], df1); mfma32a(accq[1], fa[3], df1);
      rd_t_half<0, 4>(fa[0], fa[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T2 | T3, T4
      mfma32a(accq[2], fb[0], df0); mfma32a(accq[3], fb[1], df0);
      rd_t_half<0, 5>(fa[2], fa[3], tb);
      MG_SCHED_FENCE();
      issue_part(t + NST - 1, nb, 3);
      MG_LGKM(8);                       // T3 | T4, T5
      mfma32a(accq[2], fb[2], df1); mfma32a(accq[3], fb[3], df1);
      rd_t_half<0, 6>(fb[0], fb[1], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T4 | T5, T6
      mfma32a(accq[4], fa[0], df0); mfma32a(accq[5], fa[1], df0);
      rd_t_half<0, 7>(fb[2], fb[3], tb);
      MG_SCHED_FENCE();
      MG_LGKM(8);                       // T5 | T6, T7
      mfma32a(accq[4], fa[2], df1); mfma32a(accq[5], fa[3], df1);
      MG_SCHED_FENCE();
      MG_LGKM(4);                       // T6 | T7
      mfma32a(accq[6], fb[0], df0); mfma32a(accq[7], fb[1], df0);
      MG_SCHED_FENCE();
      MG_LGKM(0);                       // T7
      mfma32a(accq[6], fb[2], df1);
      mfma32a_last(accq[7], fb[3], df1);
    }
    sc = sc == NST - 1 ? 0 : sc + 1;
  }
  for (; t < ntiles; ++t) {           // tiles that only the later waves of the block need: move this wave's share of them
    if constexpr (NST == 3) MG_WAIT_VMCNT(8); else MG_WAIT_VMCNT(16);
    MG_BARRIER_KEEP_DMA();
    issue(t + NST - 1, sc == 0 ? NST - 1 : sc - 1);
    sc = sc == NST - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);                   // drain the ring's trailing loads before the ring becomes staging space
  MG_BARRIER_KEEP_DMA();
  store_grad_tile32_tr(gq, accq, 0.0625f, smem + wave * (32 * EP_ROW), b, h, H, S, qt0 + wave * 32, l31, hi);
}


constexpr int TRF_STAGE = 2 * ROW_TILE;            // K rows | V rows
constexpr int TRF_STAGES = 4;

MG_DEV float pair_max_tr(float x) {          // over the two lanes {l, l ^ 32} that hold the two halves of a query's keys
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
MG_DEV float pair_sum_tr(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

// ---------------------------------------------------------------------------
// Causal flash-attention forward on 32-query waves (the arithmetic and the schedule of attn_prefill32_kernel, attention_fwd32.hip:
// both products transposed, fp32 online softmax in the exp2 domain with the deferred running maximum, the softmax of a tile in two
// halves beside MFMA bursts of other tiles) with V read as ROWS: O^T += V^T P^T takes its V^T fragments from the V row image with
// ds_read_b64_tr_b16.  No V^T tensor exists; K and V may be column ranges of the fused qkv activation (AttnRows).
__global__ __launch_bounds__(256) void attn_fwd32_tr_kernel(const AttnRows x, mg_bf16* __restrict__ out, int64_t ld_out,
                                                            float* __restrict__ lse, int B, int H, int S, float defer) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;   // longest blocks first
  const int qrow = qt0 + wave * 32 + l31, qrow_c = min(qrow, S - 1);
  const int64_t xoff = (int64_t)b * x.stride_b + (int64_t)h * x.stride_h;
  const mg_bf16* kbase = x.k + xoff;
  const mg_bf16* vbase = x.v + xoff;
  const uint32_t ldb = (uint32_t)x.ld * 2u;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ ((hi << 2) | ((wave & 1) << 1))) << 4);
  // piece i (0..3) of the two images of tile min(t, last): past the last tile the ring re-loads it (in bounds, never read)
  auto issue_part = [&](int t, int buf, int i) {
    const int tc = min(t, ntiles - 1);
    const uint32_t st = smem_u + (uint32_t)(buf * TRF_STAGE + (wave * 4 + i) * 1024);
    const uint32_t off = (uint32_t)min(tc * 32 + row0 + 2 * i, S - 1) * ldb + (c0b ^ (uint32_t)((((i & 1) << 3) | (i >> 1)) << 4));
    glds16su(kbase, off, st);
    glds16su(vbase, off, st + ROW_TILE);
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
#pragma unroll
  for (int i = 0; i < TRF_STAGES - 1; ++i) issue(i, i);

  bf16x8 qf[16];
  {
    const mg_bf16* qp = x.q + xoff + (int64_t)qrow_c * x.ld + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }
  f32x16 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  }
  float m2 = -1e30f, lsum = 0.f;
  const float sc2 = 0.0625f * 1.4426950408889634f;  // 1/sqrt(256) * log2(e)
  const int my_first = qt0 + wave * 32;
  const int n_act = min(ntiles, ((my_first + 31) >> 5) + 1);   // this wave's tiles: 0 .. n_act-1 (later ones are fully masked for it)
  const int R = perm32(l31);
  const uint32_t rb = smem_u + row_base32(R, tr_swz(R), hi);
  const uint32_t tb0 = smem_u + tr_lane_base(lane);
  const int lim0 = min(qrow, S - 1) - hi * 8;     // key (r >> 3) 16 + (r & 7) of tile kv0 is visible iff it is <= lim0 - kv0
  // ONE accumulator-file copy of the Q fragments (see attention_fwd32.hip)
  bf16x8 qa[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) asm volatile("" : "=a"(qa[ks]) : "0"(qf[ks]));

  bf16x8 fa[4], fb[4];
  f32x16 sA, sB;                       // scores of the current / the next tile, swapping roles every iteration (no copies)
  float alpha;
  auto part1 = [&](f32x16& sn, int kv0) {
    if (kv0 + 31 > my_first || kv0 + 32 > S) {
      const int lim = lim0 - kv0;
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[r] = ((r >> 3) * 16 + (r & 7)) > lim ? -1e30f : sn[r];
    }
    float tmax = sn[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sn[r]);
    tmax = pair_max_tr(tmax);
    const float cand = tmax * sc2;
    const float mnew = (cand > m2 + defer) ? cand : m2;
    alpha = __builtin_amdgcn_exp2f(m2 - mnew);
    m2 = mnew;
  };

  // ---- prologue: tiles 0 and 1 landed; S^T(0) and part 1 of its softmax ----
  MG_WAIT_VMCNT(8);
  MG_BARRIER_KEEP_DMA();
  {
    rd_row4_asm(fa, rb, 0);
    rd_row4_asm(fb, rb, 1);
    MG_SCHED_FENCE();
    MG_LGKM(4);
    mfma32v0_ba(sA, fa[0], qa[0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) mfma32v_ba(sA, fa[i], qa[i]);
    rd_row4_asm(fa, rb, 2);
    MG_SCHED_FENCE();
    MG_LGKM(4);
#pragma unroll
    for (int i = 0; i < 4; ++i) mfma32v_ba(sA, fb[i], qa[4 + i]);
    rd_row4_asm(fb, rb, 3);
    MG_SCHED_FENCE();
    MG_LGKM(4);
#pragma unroll
    for (int i = 0; i < 4; ++i) mfma32v_ba(sA, fa[i], qa[8 + i]);
    MG_SCHED_FENCE();
    MG_LGKM(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) mfma32v_ba(sA, fb[i], qa[12 + i]);
    mfma32v_ba_last(sA, fb[3], qa[15]);
    part1(sA, 0);
  }
  int sc = 0;
  // one iteration: `cur` = masked scores of tile t (part 1 done), `nxt` receives S^T(t+1)
  auto iteration = [&](f32x16& cur, f32x16& nxt, int t) {
    MG_WAIT_VMCNT(8);                 // this wave's pieces of tile t+1 landed (tile t+2 may be in flight)
    MG_BARRIER_KEEP_DMA();            // tile t+1 complete; everyone is done with iteration t-1 (K(t), V(t-1)); lgkmcnt = 0
    const int nb = sc == 0 ? TRF_STAGES - 1 : sc - 1;
    issue_part(t + TRF_STAGES - 1, nb, 0);
    issue_part(t + TRF_STAGES - 1, nb, 1);
    const int scn = sc == TRF_STAGES - 1 ? 0 : sc + 1;
    const uint32_t krow = (uint32_t)(scn * TRF_STAGE) + rb;
    const uint32_t tb = (uint32_t)(sc * TRF_STAGE) + tb0;
    float psum = 0.f, pe = 0.f;
    u32x4 pw0, pw1;
    auto soft = [&](int r) {          // element r of tile t (r even: kept for the pack with r + 1)
      const float pr = __builtin_amdgcn_exp2f(fmaf(cur[r], sc2, -m2));
      psum += pr;
      if (r & 1) { if (r < 8) pw0[r >> 1] = pack2bf(pe, pr); else pw1[(r - 8) >> 1] = pack2bf(pe, pr); }
      else pe = pr;
    };
    // ---- block A: S^T(t+1) MFMAs; behind each of them one exponential of tile t ----
    rd_row4_asm(fa, krow, 0);
    rd_row4_asm(fb, krow, 1);
    MG_SCHED_FENCE();
    MG_LGKM(4);                       // K0 | K1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 0) mfma32v0_ba(nxt, fa[0], qa[0]); else mfma32v_ba(nxt, fa[i], qa[i]);
      MG_SCHED_FENCE();
      soft(i);
      MG_SCHED_FENCE();
    }
    rd_row4_asm(fa, krow, 2);
    MG_SCHED_FENCE();
    MG_LGKM(4);                       // K1 | K2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma32v_ba(nxt, fb[i], qa[4 + i]);
      MG_SCHED_FENCE();
      soft(4 + i);
      MG_SCHED_FENCE();
    }
    rd_row4_asm(fb, krow, 3);
    MG_SCHED_FENCE();
    MG_LGKM(4);                       // K2 | K3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma32v_ba(nxt, fa[i], qa[8 + i]);
      MG_SCHED_FENCE();
      soft(8 + i);
      MG_SCHED_FENCE();
    }
    rd_t_half<ROW_TILE, 0>(fa[0], fa[1], tb);      // V^T halves of tile t through the ring {fa[0..1], fa[2..3], fb[0..1], fb[2..3]}
    rd_t_half<ROW_TILE, 1>(fa[2], fa[3], tb);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // K3 | T0, T1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3) mfma32v_ba(nxt, fb[i], qa[12 + i]); else mfma32v_ba_last(nxt, fb[3], qa[15]);
      MG_SCHED_FENCE();
      soft(12 + i);
      MG_SCHED_FENCE();
    }
    rd_t_half<ROW_TILE, 2>(fb[0], fb[1], tb);
    lsum = lsum * alpha + psum;
    bf16x8 pf0 = __builtin_bit_cast(bf16x8, pw0), pf1 = __builtin_bit_cast(bf16x8, pw1);
    mfma_operand_ready(pf0, pf1);
    MG_SCHED_FENCE();
    issue_part(t + TRF_STAGES - 1, nb, 2);
    issue_part(t + TRF_STAGES - 1, nb, 3);
    // ---- the running maximum moved by more than the deferral threshold (rare): O^T and l were kept at the old one ----
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
      asm volatile("" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]), "+a"(o[4]), "+a"(o[5]), "+a"(o[6]), "+a"(o[7]));
#pragma unroll
      for (int db = 0; db < 8; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        asm volatile("" : "+a"(o[db]));      // back in its AGPRs before the next tuple is touched
        MG_SCHED_FENCE();
      }
    }
    MG_SCHED_FENCE();
    // ---- block B: O^T += V^T(t) P^T(t), 16 MFMAs = 8 halves; part 1 of softmax(t+1) behind the first two ----
    MG_LGKM(8);                       // T0 | T1, T2
    mfma32a(o[0], fa[0], pf0); mfma32a(o[1], fa[1], pf0);
    rd_t_half<ROW_TILE, 3>(fb[2], fb[3], tb);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // T1 | T2, T3
    mfma32a(o[0], fa[2], pf1); mfma32a(o[1], fa[3], pf1);
    rd_t_half<ROW_TILE, 4>(fa[0], fa[1], tb);
    MG_SCHED_FENCE();
    part1(nxt, (t + 1) * 32);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // T2 | T3, T4
    mfma32a(o[2], fb[0], pf0); mfma32a(o[3], fb[1], pf0);
    rd_t_half<ROW_TILE, 5>(fa[2], fa[3], tb);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // T3 | T4, T5
    mfma32a(o[2], fb[2], pf1); mfma32a(o[3], fb[3], pf1);
    rd_t_half<ROW_TILE, 6>(fb[0], fb[1], tb);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // T4 | T5, T6
    mfma32a(o[4], fa[0], pf0); mfma32a(o[5], fa[1], pf0);
    rd_t_half<ROW_TILE, 7>(fb[2], fb[3], tb);
    MG_SCHED_FENCE();
    MG_LGKM(8);                       // T5 | T6, T7
    mfma32a(o[4], fa[2], pf1); mfma32a(o[5], fa[3], pf1);
    MG_SCHED_FENCE();
    MG_LGKM(4);                       // T6 | T7
    mfma32a(o[6], fb[0], pf0); mfma32a(o[7], fb[1], pf0);
    MG_SCHED_FENCE();
    MG_LGKM(0);                       // T7
    mfma32a(o[6], fb[2], pf1);
    mfma32a_last(o[7], fb[3], pf1);
    sc = scn;
  };
  int t = 0;
  for (; t + 1 < n_act; t += 2) {
    iteration(sA, sB, t);
    iteration(sB, sA, t + 1);
  }
  if (t < n_act) { iteration(sA, sB, t); ++t; }
  for (; t < ntiles; ++t) {           // tiles that only the later waves of the block need: move this wave's share of them
    MG_WAIT_VMCNT(8);
    MG_BARRIER_KEEP_DMA();
    issue(t + TRF_STAGES - 1, sc == 0 ? TRF_STAGES - 1 : sc - 1);
    sc = sc == TRF_STAGES - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);                   // drain the ring's trailing loads before the ring becomes staging space
  MG_BARRIER_KEEP_DMA();
  lsum = pair_sum_tr(lsum);
  const float inv = 1.0f / lsum;
  // O^T (acc[db] = d-rows db*32.. x 32 queries) -> bf16 rows through a wave-private LDS image, out as whole 512-byte rows
  char* stage = smem + wave * (32 * EP_ROW);
  char* wr = stage + l31 * EP_ROW + hi * 8;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const u32x2 w = {pack2bf(o[db][rq * 4] * inv, o[db][rq * 4 + 1] * inv), pack2bf(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv)};
      *(u32x2*)(wr + db * 64 + rq * 16) = w;
    }
    MG_SCHED_FENCE();             // one tuple at a time: 128 accumulators read at once are 128 VGPRs the loop pays for
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 2 + hi;
    const u32x4 w = *(const u32x4*)(stage + row * EP_ROW + l31 * 16);
    const int s = qt0 + wave * 32 + row;
    if (s < S) *(u32x4*)(out + (int64_t)(b * S + s) * ld_out + h * DH + l31 * 8) = w;
  }
  if (lse && hi == 0 && qrow < S) lse[(int64_t)bh * S + qrow] = (m2 + log2f(lsum)) * 0.6931471805599453f;
}

// ---------------------------------------------------------------------------
// GPT-J rotary (interleaved pairs, reference magma/language_model.py via HF GPT-J: rotate_every_two) applied IN PLACE to the first
// rot_dim columns of every q and k head of a fused qkv activation [B*S, >= 3 H 256]: a quarter of q and k is read and written once,
// v is not touched; the attention kernels then take q / k / v straight from this buffer (AttnRows).  One lane = 8 columns of one
// (row, q-or-k, head); grid-stride over B*S rows x 2 H heads x rot_dim / 8.
__global__ __launch_bounds__(256) void rotary_qk_inplace_kernel(mg_bf16* __restrict__ qkv, int64_t ld_qkv, int S, int H, int rot_dim,
                                                                const float* __restrict__ sin_t, const float* __restrict__ cos_t,
                                                                int64_t total) {
  const int per_head = rot_dim >> 3, half_rot = rot_dim >> 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % per_head);
    const int64_t rh = i / per_head;
    const int hh = (int)(rh % (2 * H));              // head index over [q heads | k heads]: the two sections are adjacent
    const int64_t row = rh / (2 * H);
    const int pos = (int)(row % S);
    mg_bf16* p = qkv + row * ld_qkv + (int64_t)hh * DH + c * 8;
    u32x4 w = *(const u32x4*)p;
    const float* sp = sin_t + (int64_t)pos * half_rot + c * 4;
    const float* cp = cos_t + (int64_t)pos * half_rot + c * 4;
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const float sn = sp[pi], cs = cp[pi];
      const float a = bflo(w[pi]), bq = bfhi(w[pi]);
      w[pi] = pack2bf(a * cs - bq * sn, bq * cs + a * sn);
    }
    *(u32x4*)p = w;
  }
}

// ld2[b,h,s] = {-16 lse, -rowsum(dO o O)}: the two per-query statistics of the backward (attention_bwd.hip: attn_bwd_prep_kernel).
// A half-wave per (b, s, h) row of 256 (16 bytes per lane), four rows per half-wave in flight: 8 rows = 4 KiB of dO and of O per
// wave (the one-row-per-wave form with 8-byte loads streamed its 537 MB at 3.8 TB/s); O rows may sit in a wider buffer.
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ O,
                                                             const float* __restrict__ lse, float* __restrict__ ld2,
                                                             int B, int H, int S, int64_t ld_o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t rows = (int64_t)B * S * H;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 8 + half;   // rows row0 + 2 j, (b,s,h) order = memory order of [M, H*256]
  u32x4 a[4], o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t row = min(row0 + 2 * j, rows - 1);
    a[j] = *(const u32x4*)(dO + row * DH + l32 * 8);
    o[j] = *(const u32x4*)(O + (row / H) * ld_o + (row % H) * DH + l32 * 8);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += bflo(a[j][w]) * bflo(o[j][w]) + bfhi(a[j][w]) * bfhi(o[j][w]);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);     // within the half-wave
    const int64_t row = row0 + 2 * j;
    if (l32 == 0 && row < rows) {
      const int h = (int)(row % H);
      const int64_t bs = row / H;
      const int sidx = (int)(bs % S), b = (int)(bs / S);
      const int64_t i = ((int64_t)b * H + h) * S + sidx;
      *(mg_f32x2*)(ld2 + i * 2) = (mg_f32x2){-16.0f * lse[i], -s};
    }
  }
}

}  // namespace

int attn_bwd_dkdv32_tr_launch(const AttnRows& x, const mg_bf16* dO, const float* ld2,
                              const GradOut& gk, const GradOut& gv, int B, int H, int S, int stages, hipStream_t s, const char* who, int nrun) {
  const int nblk = nrun > 0 ? std::min(nrun, (S + 127) / 128) : (S + 127) / 128;
  const dim3 grid((unsigned)(nblk * B * H));
  if (stages == 3) {
    const int lds = 3 * TRKV_STAGE > TRKV_EPILOGUE ? 3 * TRKV_STAGE : TRKV_EPILOGUE;
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv32_tr_kernel<3>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv32_tr_kernel<3>, grid, dim3(256), lds, s, x, dO, ld2, gk, gv, B, H, S, nblk);
  } else {
    const int lds = 2 * TRKV_STAGE > TRKV_EPILOGUE ? 2 * TRKV_STAGE : TRKV_EPILOGUE;
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv32_tr_kernel<2>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv32_tr_kernel<2>, grid, dim3(256), lds, s, x, dO, ld2, gk, gv, B, H, S, nblk);
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

int attn_bwd_dq32_tr_launch(const AttnRows& x, const mg_bf16* dO, const float* ld2,
                            const GradOut& gq, int B, int H, int S, int stages, hipStream_t s, const char* who, int nrun) {
  const int nblk = nrun > 0 ? std::min(nrun, (S + 127) / 128) : (S + 127) / 128;
  const dim3 grid((unsigned)(nblk * B * H));
  if (stages == 4) {
    const int lds = 4 * TRQ_STAGE;
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dq32_tr_kernel<4>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq32_tr_kernel<4>, grid, dim3(256), lds, s, x, dO, ld2, gq, B, H, S, nblk);
  } else {
    const int lds = 3 * TRQ_STAGE;
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dq32_tr_kernel<3>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq32_tr_kernel<3>, grid, dim3(256), lds, s, x, dO, ld2, gq, B, H, S, nblk);
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

int attn_fwd32_tr_launch(const AttnRows& x, mg_bf16* out, int64_t ld_out, float* lse, int B, int H, int S, float defer, hipStream_t s,
                         const char* who) {
  const int lds = TRF_STAGES * TRF_STAGE;
  if (ld_out & 7) MG_FAIL(MG_ERR_SHAPE, "%s: the 32-query kernel stores 16-byte pieces: ld_out %% 8 == 0", who);
  if (int rc = mg_allow_dynamic_lds((const void*)attn_fwd32_tr_kernel, lds, who)) return rc;
  hipLaunchKernelGGL(attn_fwd32_tr_kernel, dim3((unsigned)(((S + 127) / 128) * B * H)), dim3(256), lds, s, x, out, ld_out, lse, B, H, S, defer);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

namespace {
int check_rows(const char* who, const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, int64_t ld_row, int64_t stride_b, int64_t stride_h,
               int32_t B, int32_t H, int32_t S) {
  if (B <= 0 || H <= 0 || S <= 0) MG_FAIL(MG_ERR_SHAPE, "%s: B, H, S must be positive", who);
  if (!q || !k || !v) MG_FAIL(MG_ERR_SHAPE, "%s: null pointer", who);
  if (!MG_ALIGNED16(q) || !MG_ALIGNED16(k) || !MG_ALIGNED16(v)) MG_FAIL(MG_ERR_ALIGN, "%s: q, k, v must be 16-byte aligned", who);
  if (ld_row < DH || (ld_row & 7) || (stride_b & 7) || (stride_h & 7) || stride_b < 0 || stride_h < 0)
    MG_FAIL(MG_ERR_SHAPE, "%s: ld_row must be a multiple of 8 and >= 256, stride_b / stride_h non-negative multiples of 8", who);
  if ((int64_t)S * ld_row * 2 >= ((int64_t)1 << 31)) MG_FAIL(MG_ERR_SHAPE, "%s: S * ld_row exceeds the 32-bit byte offsets of the tile loaders", who);
  return MG_OK;
}
}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------------
extern "C" int mg_rotary_qk_inplace_bf16(mg_bf16* qkv, int64_t ld_qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                                         const float* sin_t, const float* cos_t, void* stream) {
  if (B <= 0 || S <= 0 || H <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_qk_inplace_bf16: B, S, H must be positive");
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_qk_inplace_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !MG_ALIGNED16(qkv)) MG_FAIL(MG_ERR_ALIGN, "mg_rotary_qk_inplace_bf16: qkv must be a 16-byte aligned pointer");
  if (ld_qkv < (int64_t)3 * H * DH || (ld_qkv & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_qk_inplace_bf16: ld_qkv must be a multiple of 8 and >= 3*H*256");
  if (rot_dim == 0) return MG_OK;
  if (!sin_t || !cos_t) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_qk_inplace_bf16: rotary tables missing");
  const int64_t total = (int64_t)B * S * 2 * H * (rot_dim >> 3);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(rotary_qk_inplace_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, qkv, ld_qkv, S, H,
                     rot_dim, sin_t, cos_t, total);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_fwd_rows_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, int64_t ld_row, int64_t stride_b,
                                     int64_t stride_h, mg_bf16* out, int64_t ld_out, float* lse, int32_t B, int32_t H, int32_t S,
                                     void* stream) {
  if (int rc = check_rows("mg_attn_fwd_rows_bf16", q, k, v, ld_row, stride_b, stride_h, B, H, S)) return rc;
  if (ld_out == 0) ld_out = (int64_t)H * DH;
  if (!out || !MG_ALIGNED16(out) || ld_out < (int64_t)H * DH || (ld_out & 7))
    MG_FAIL(MG_ERR_SHAPE, "mg_attn_fwd_rows_bf16: out must be 16-byte aligned, ld_out 0 or a multiple of 8 >= H*256");
  const AttnRows x{q, k, v, stride_b, stride_h, (int)ld_row};
  return attn_fwd32_tr_launch(x, out, ld_out, lse, B, H, S, 8.0f, (hipStream_t)stream, "mg_attn_fwd_rows_bf16");
}

extern "C" int mg_attn_bwd_rows_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, int64_t ld_row, int64_t stride_b,
                                     int64_t stride_h, const mg_bf16* dO, const mg_bf16* O, int64_t ld_o, const float* lse, float* D,
                                     mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, mg_bf16* dqkv, int32_t rot_dim, const float* sin_t,
                                     const float* cos_t, int32_t B, int32_t H, int32_t S, uint8_t* dqkv8, uint8_t* dqkv8_scales,
                                     int32_t first_rows, void* stream) {
  const char* who = "mg_attn_bwd_rows_bf16";
  if (first_rows < 0) MG_FAIL(MG_ERR_SHAPE, "%s: first_rows must be >= 0", who);
  if (int rc = check_rows(who, q, k, v, ld_row, stride_b, stride_h, B, H, S)) return rc;
  if (!dO || !O || !lse || !D) MG_FAIL(MG_ERR_SHAPE, "%s: null pointer", who);
  if (!MG_ALIGNED16(dO) || !MG_ALIGNED16(O)) MG_FAIL(MG_ERR_ALIGN, "%s: dO and O must be 16-byte aligned", who);
  if (ld_o < (int64_t)H * DH || (ld_o & 7)) MG_FAIL(MG_ERR_SHAPE, "%s: ld_o must be a multiple of 8 and >= H * 256", who);
  GradOut gq, gk, gv;
  if (dqkv8 && (!dqkv8_scales || ((uintptr_t)dqkv8 & 7) || ((3 * H * DH) & 127)))
    MG_FAIL(MG_ERR_SHAPE, "%s: the MX copy of dqkv needs its scale array, 8-byte alignment and 3 H 256 %% 128 == 0", who);
  if (dqkv || dqkv8) {
    if (dq || dk || dv) MG_FAIL(MG_ERR_SHAPE, "%s: either dqkv (merged; bf16 and / or its MX copy) or dq / dk / dv", who);
    if (dqkv && !MG_ALIGNED16(dqkv)) MG_FAIL(MG_ERR_ALIGN, "%s: dqkv must be 16-byte aligned", who);
    if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "%s: rot_dim must be a multiple of 8 in [0,256]", who);
    if (rot_dim && (!sin_t || !cos_t)) MG_FAIL(MG_ERR_SHAPE, "%s: rotary tables missing", who);
    gq = GradOut{nullptr, dqkv, sin_t, cos_t, 0, rot_dim}; gk = GradOut{nullptr, dqkv, sin_t, cos_t, 1, rot_dim}; gv = GradOut{nullptr, dqkv, sin_t, cos_t, 2, 0};
    for (GradOut* g : {&gq, &gk, &gv}) { g->q8 = dqkv8; g->q8_scales = dqkv8_scales; g->mx_rows = B * S; }
  } else {
    if (!dq || !dk || !dv) MG_FAIL(MG_ERR_SHAPE, "%s: null gradient pointer", who);
    if (!MG_ALIGNED16(dq) || !MG_ALIGNED16(dk) || !MG_ALIGNED16(dv)) MG_FAIL(MG_ERR_ALIGN, "%s: dq, dk, dv must be 16-byte aligned", who);
    gq = GradOut{dq, nullptr, nullptr, nullptr, 0, 0}; gk = GradOut{dk, nullptr, nullptr, nullptr, 1, 0}; gv = GradOut{dv, nullptr, nullptr, nullptr, 2, 0};
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * S * H;
  hipLaunchKernelGGL(attn_bwd_stats_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, s, dO, O, lse, D, B, H, S, ld_o);
  const AttnRows x{q, k, v, stride_b, stride_h, (int)ld_row};
  // first_rows > 0: only the gradients of positions < first_rows are wanted (the bottom block of a frozen LM: nothing but the image
  // prefix receives a gradient below it) -- the first ceil(first_rows / 128) query blocks of dQ and key blocks of dK / dV run (whole
  // blocks are written); every statistic and every later query still takes part in dK / dV
  const int nrun = first_rows > 0 ? (first_rows + 127) / 128 : 0;
  if (int rc = attn_bwd_dq32_tr_launch(x, dO, D, gq, B, H, S, 4, s, who, nrun)) return rc;
  return attn_bwd_dkdv32_tr_launch(x, dO, D, gk, gv, B, H, S, 3, s, who, nrun);
}
