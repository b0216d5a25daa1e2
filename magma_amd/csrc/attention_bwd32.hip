// attention_bwd32.hip -- causal flash-attention backward, head dim 256, on 32-row waves (gfx950, round 5).
//
// The 16-row-wave kernels of attention_bwd.hip recompute S three times and dP twice (dQ | dK | dV as three kernels: one
// 64-register accumulator set each, two waves per SIMD) and read every LDS fragment for ONE 16x16x32 MFMA.  Here a wave
// owns 32 rows on v_mfma_f32_32x32x16_bf16 -- every fragment read feeds twice the matrix work -- and runs ALONE on its
// SIMD (256-thread workgroups, up to 512 registers per lane: the accumulators go to the AGPR half of the file):
//
//   attn_bwd_dkdv32_kernel  one workgroup per 128 keys (4 waves x 32, lane&31 = key), loops over query tiles of 32
//                           from the diagonal down; S and dP are computed ONCE per tile and feed both products:
//                             S  = Q K^T          dP = dO V^T           P = exp2(S/16 log2e - lse2)
//                             dV^T += dO^T P      dK^T += Q^T dS        dS = P o (dP - D) / 16
//                           4 x 16 MFMAs per tile step: 7 matmul-units for the whole backward instead of 8, and Q / dO /
//                           {lse, D} stream through LDS once instead of twice.
//   attn_bwd_dq32_kernel    one workgroup per 128 queries (4 waves x 32, lane&31 = query), loops over KV tiles of 32:
//                             S^T = K Q^T   dP^T = V dO^T   dQ^T += K^T dS^T
//
// Operands, LDS images, swizzles and the LDS-DMA loaders are those of attention_bwd.hip (attn_tile_device.h): the images
// stay conflict-free for the 32-row fragment reads (a ds_read_b128 lane group now covers rows {0-3, 12-15, 20-27} of one
// chunk column instead of 16 rows of a 16-row block: row_swz / t_swz take 16 distinct values on them).
//
// Row permutation.  The 32x32 accumulator of lane (l31 = lane & 31, hi = lane >> 5) holds MFMA rows
// i(r) = (r & 3) + 8 (r >> 2) + 4 hi, r = 0..15.  Feeding MFMA row i from tile row pi(i) = i with bits 2 and 3 swapped makes
// register r the tile row (r >> 3) 16 + 8 hi + (r & 7): registers 0..7 / 8..15 are exactly the 8 consecutive rows the lane
// supplies as the k-operand of the two 16-deep MFMA steps of the next product -- no cross-lane movement (the 16x16 kernels
// use (li >> 2) 8 + (li & 3) for the same purpose).
#include "attn_bwd_device.h"
#include "attn32_device.h"
#include <stdlib.h>

#ifdef MG_GEMM_ABLATIONS
// ABL 6 of attn_bwd_dkdv32_kernel: per-wave s_memtime totals of the five segments of a tile step + the step count
__device__ unsigned long long g_attn_stamps[4096 * 4 * 8];
extern "C" int mg_debug_attn_stamps(void* host_dst, int64_t bytes) {
  const hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_attn_stamps), (size_t)bytes);
  if (e != hipSuccess) MG_FAIL(MG_ERR_HIP, "mg_debug_attn_stamps: %s", hipGetErrorString(e));
  return MG_OK;
}
#endif

namespace {

// A wave's 32 x 256 gradient tile (acc[db] = d-rows db*32.. x 32 sequence positions) -> bf16, through a wave-private LDS
// image, out as whole 512-byte rows: 16 dwordx4 stores of 1 KiB per wave instead of 64 dwordx2 at a row stride.
MG_DEV void store_grad_tile32(const GradOut& g, const f32x16 (&acc)[8], float scale, char* stage, int b, int h, int H, int S,
                              int row0, int l31, int hi) {
  const bool rot = g.merged && g.which < 2 && g.rot_dim > 0;
  const int half_rot = g.rot_dim >> 1;
  const int s_me = min(row0 + l31, S - 1);
  char* wr = stage + l31 * EP_ROW + hi * 8;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float x0 = acc[db][rq * 4] * scale, x1 = acc[db][rq * 4 + 1] * scale, x2 = acc[db][rq * 4 + 2] * scale, x3 = acc[db][rq * 4 + 3] * scale;
      const int d = db * 32 + rq * 8 + hi * 4;            // this lane's 4 consecutive columns
      if (rot && d < g.rot_dim) {                         // rot_dim % 8 == 0: the 4 columns are inside or outside together
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
  // the image is private to this wave: its LDS accesses complete in order, only the compiler must not reorder them
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 2 + hi;
    const u32x4 w = *(const u32x4*)(stage + row * EP_ROW + l31 * 16);
    const int s = row0 + row;
    if (s < S) *(u32x4*)(grad_row_ptr(g, b, h, H, S, s) + l31 * 8) = w;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the image is rewritten by the next tile of this wave
  __builtin_amdgcn_wave_barrier();
}

constexpr int KV_STAGE = 2 * ROW_TILE + 2 * T_TILE + 2 * LD_TILE;   // Q rows | dO rows | Q^T | dO^T | {-16 lse x 32, -D x 32} | pad: a multiple of 512 (rd_row4x)
static_assert(KV_STAGE % 512 == 0, "rd_row4x: the stage offset must not reach into the chunk bits");
constexpr int KV_STAGES = 2;
constexpr int KV_LD_OFF = 2 * ROW_TILE + 2 * T_TILE;

// ---------------------------------------------------------------------------
// SPREAD: the 17 LDS-DMA pieces of the next tile are issued in four groups between the MFMA phases instead of all at
// the top of the step (an LDS-DMA piece costs its wave 60-180 issue cycles, MI355X_MICROARCH.md).
// ABL (timing ablations, WRONG results; `make ABL=1` library only, MAGMA_ATTN_BWD32_ABL=n): 1 = no LDS-DMA after the prologue,
// 2 = no LDS fragment reads after a wave's first tile (stale registers), 3 = no MFMAs, 4 = no barrier (and no DMA), 5 = no
// softmax arithmetic (P and dS are the raw accumulators): each part's price is the time it removes.
#define MMV(c, a, b)  do { if constexpr (ABL != 3) mfma32v(c, a, b); else asm volatile("" : "+v"(c) : "v"(a), "v"(b)); } while (0)
#define MMVL(c, a, b) do { if constexpr (ABL != 3) mfma32v_last(c, a, b); else asm volatile("" : "+v"(c) : "v"(a), "v"(b)); } while (0)
#define MMA(c, a, b)  do { if constexpr (ABL != 3) mfma32a(c, a, b); else asm volatile("" : "+a"(c) : "v"(a), "v"(b)); } while (0)
#define MMAL(c, a, b) do { if constexpr (ABL != 3) mfma32a_last(c, a, b); else asm volatile("" : "+a"(c) : "v"(a), "v"(b)); } while (0)
template <bool SPREAD, int ABL = 0>
__global__ __launch_bounds__(256) void attn_bwd_dkdv32_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ qt, const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ dOt,
    const float* __restrict__ ld2, const GradOut gk, const GradOut gv, int B, int H, int S, int ld_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nblk = (S + 127) >> 7;              // one (b,h) per XCD at a time, see attn_bwd_dq_kernel
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int k0 = (wg - bh * nblk) * 128;       // earliest key blocks (most query tiles) first
  const int key = k0 + wave * 32 + l31, key_c = min(key, S - 1);
  const int dmodel = H * DH;
  const mg_bf16* qb = q + (int64_t)bh * S * DH;
  const mg_bf16* qtb = qt + (int64_t)bh * DH * ld_t;
  const mg_bf16* dotb = dOt + (int64_t)bh * DH * ld_t;
  const mg_bf16* dob = dO + (int64_t)b * S * dmodel + h * DH;   // row stride dmodel
  const float* ldb = ld2 + (int64_t)bh * S * 2;

  const int t_begin = k0 >> 5;                 // first query tile that can see key k0
  const int t_end = (S + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  // piece i (0..3) of each of the four 16-KiB images of tile t; wave w moves 1-KiB blocks 4w .. 4w+3 of an image.  Scalar base
  // + 32-bit lane offset, and ONE lane constant per image kind: a row piece covers tile rows 8w + 2i + hi, whose swizzle
  // row_swz = ((2i + hi) & 3) | (w << 2) differs between the pieces only in bit 1 ((i & 1) << 1); a T piece is 1 KiB of the
  // contiguous tile with the chunk swizzle of rows (lane >> 2) -- the same for every piece.  (Per-piece 64-bit lane addresses
  // are hoisted out of the tile loop by hipcc, 2 registers each, and this kernel has none to spare.)
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ (hi | (wave << 2))) << 4);
  const uint32_t tl = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ t_swz(lane >> 2)) << 4));
  const uint32_t do_stride = (uint32_t)dmodel * 2u;
  auto issue_part = [&](int t, int buf, int i) {
    const int q0 = t * 32;
    const uint32_t st = smem_u + (uint32_t)(buf * KV_STAGE + (wave * 4 + i) * 1024);
    const uint32_t r = (uint32_t)min(q0 + row0 + 2 * i, S - 1);
    const uint32_t cb = c0b ^ (uint32_t)((i & 1) << 5);
    glds16su(qb, r * 512u + cb, st);
    glds16su(dob, r * do_stride + cb, st + ROW_TILE);
    const uint32_t toff = (uint32_t)t * (uint32_t)(DH * 64) + (uint32_t)((wave * 4 + i) * 1024) + tl;
    glds16su(qtb, toff, st + 2 * ROW_TILE);
    glds16su(dotb, toff, st + 2 * ROW_TILE + T_TILE);
    if (i == 0)   // statistics as two arrays: lanes 0-31 fetch lse2 of query q0 + l31, lanes 32-63 its D (every wave writes the same 256 B)
      glds4su(ldb, (uint32_t)(min(q0 + l31, S - 1) * 8 + hi * 4), smem_u + (uint32_t)(buf * KV_STAGE + KV_LD_OFF));
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
  issue(t_begin, 0);

  bf16x8 kf[16], vf[16];
  {
    const mg_bf16* kp = k + ((int64_t)bh * S + key_c) * DH + hi * 8;
    const mg_bf16* vp = v + ((int64_t)bh * S + key_c) * DH + hi * 8;
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
  const int sw = row_swz(R);
  const uint32_t rb = row_base32(R, sw, hi);
  const int tx = hi ^ t_swz(l31);
  const uint32_t ls_addr = smem_u + (uint32_t)(KV_LD_OFF + hi * 32);   // + g*64 (+128 for D): 8 consecutive queries = 32 bytes

  MG_USE8(kf); MG_USE8(vf);                    // retire the ordinary loads in hipcc's scoreboard (see attention.hip)
  {
    bf16x8* k8 = kf + 8; bf16x8* v8 = vf + 8;
    asm volatile("" ::"v"(k8[0]), "v"(k8[1]), "v"(k8[2]), "v"(k8[3]), "v"(k8[4]), "v"(k8[5]), "v"(k8[6]), "v"(k8[7]));
    asm volatile("" ::"v"(v8[0]), "v"(v8[1]), "v"(v8[2]), "v"(v8[3]), "v"(v8[4]), "v"(v8[5]), "v"(v8[6]), "v"(v8[7]));
  }
  int sc = 0;
  // Measured and NOT adopted (profiles/r05_attention_bwd32_notes.txt): walking the query tiles from the last one DOWN to the
  // diagonal, so that the 16 key blocks of a (b, h) -- side by side on one XCD -- stream the same tile at the same time (one
  // L2 fill serves all; walking up, block i is at tile 4 i + tau and two heads per XCD spread over 8 MB against 4 MB of L2):
  // 1.70 ms against 1.42 at B = 16, S = 2048.  Sixty-four waves asking for the same 64 KB at once queue on the same L2
  // channels; the Infinity Cache serves the spread-out order faster than L2 serves the aligned one.
  int t = t_begin;
  // query tiles that end before this wave's first key are fully masked for it (wave w: the first w tiles of the block): it
  // only moves its share of the data.  A loop of its own -- with the accumulators updated under a branch hipcc copies all
  // 256 of them around the control flow.
  for (const int t_act = min(t_begin + wave, t_end); t < t_act; ++t) {
    if constexpr (ABL != 1 && ABL != 4) MG_WAIT_VMCNT(0);
    if constexpr (ABL != 4) MG_BARRIER_KEEP_DMA();
    if (ABL != 1 && ABL != 4 && t + 1 < t_end) issue(t + 1, sc ^ 1);
    sc ^= 1;
  }
  bool first = true;
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
#define MG_STAMP(i_) do { if constexpr (ABL == 6) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tsum[i_] += n_ - tprev; tprev = n_; } } while (0)
  if constexpr (ABL == 6) tprev = __builtin_amdgcn_s_memtime();
  for (; t < t_end; ++t) {
    if constexpr (ABL != 1 && ABL != 4) MG_WAIT_VMCNT(0);   // this wave's pieces of tile t have landed (issued one step ago)
    if constexpr (ABL != 4) MG_BARRIER_KEEP_DMA();          // tile t complete; everyone is done with tile t-1
    const bool more = (ABL == 1 || ABL == 4) ? false : t + 1 < t_end;
    MG_STAMP(0);                      // wait + barrier
    if (more) { if (SPREAD) issue_part(t + 1, sc ^ 1, 0); else issue(t + 1, sc ^ 1); }
    const int q0 = t * 32;
    {
      const char* st = smem + sc * KV_STAGE;
      const uint32_t qrow = (uint32_t)(sc * KV_STAGE) + rb;
      const uint32_t dorow = qrow + ROW_TILE;
      const char* qtp = st + 2 * ROW_TILE + l31 * 64;
      const char* dotp = qtp + T_TILE;
      const uint32_t lsa = ls_addr + (uint32_t)(sc * KV_STAGE);
      bf16x8 fa[4], fb[4];
      // The statistics are the INITIAL VALUES of the two score accumulators: ld2 = {-16 lse, -D} per query (the statistics
      // pass writes them negated, in raw-score units), so S' = Q K^T - 16 lse and P = exp2(S' / 16 log2 e), dP' = dO V^T - D and
      // dS = P o dP' -- no statistics registers next to the accumulators, no subtractions.  Read by hand (a lane's 16 queries
      // are two runs of 8 = 2 x 32 bytes per array): for a compiler-visible ds_read of the DMA-filled statistics hipcc inserts
      // s_waitcnt vmcnt(0), which here would wait for the NEXT tile's pieces.  The reads are older than the fragment burst
      // that follows, so hipcc's own wait for that burst covers them.
      f32x4 i0, i1, i2, i3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\t"
                   "ds_read_b128 %3, %4 offset:80"
                   : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lsa) : "memory");
      f32x16 s = __builtin_shufflevector(__builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7),
                                         __builtin_shufflevector(i2, i3, 0, 1, 2, 3, 4, 5, 6, 7),
                                         0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      // ---- phase 1: S' = Q K^T - 16 lse (16 MFMAs), fragment bursts of four one burst ahead ----
      if (ABL != 2 || first) rd_row4x(fa, smem, qrow, 0);
      if (ABL != 2 || first) rd_row4x(fb, smem, qrow, 1);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) MMV(s, fa[i], kf[i]);
      if (ABL != 2 || first) rd_row4x(fa, smem, qrow, 2);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) MMV(s, fb[i], kf[4 + i]);
      if (ABL != 2 || first) rd_row4x(fb, smem, qrow, 3);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) MMV(s, fa[i], kf[8 + i]);
      asm volatile("ds_read_b128 %0, %4 offset:128\n\tds_read_b128 %1, %4 offset:144\n\tds_read_b128 %2, %4 offset:192\n\t"
                   "ds_read_b128 %3, %4 offset:208"
                   : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lsa) : "memory");
      f32x16 dp = __builtin_shufflevector(__builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7),
                                          __builtin_shufflevector(i2, i3, 0, 1, 2, 3, 4, 5, 6, 7),
                                          0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      if (ABL != 2 || first) rd_row4x(fa, smem, dorow, 0);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 3; ++i) MMV(s, fb[i], kf[12 + i]);
      MMVL(s, fb[3], kf[15]);
      if (ABL != 2 || first) rd_row4x(fb, smem, dorow, 1);
      MG_SCHED_FENCE();
      MG_STAMP(1);                    // DMA group 0 + phase 1
      if (SPREAD && more) issue_part(t + 1, sc ^ 1, 1);
      // ---- phase 2: dP' = dO V^T - D (16 MFMAs) beside P = exp2(S' sc2) ----
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) MMV(dp, fa[i], vf[i]);
      if (ABL != 2 || first) rd_row4x(fa, smem, dorow, 2);
      // only the tiles that straddle this wave's keys (and the ragged last tile) need the mask
      if (q0 < my_first + 31 || q0 + 32 > S) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qg = q0 + (r >> 3) * 16 + hi * 8 + (r & 7);
          s[r] = (key > qg || qg >= S) ? -1e30f : s[r];
        }
      }
      // The exponentials sit BETWEEN the remaining twelve dP MFMAs, two per gap (an asm MFMA has no latency the scheduler
      // knows of: left alone it emits all the VALU work in front of the burst and the matrix pipe idles meanwhile).
      float p[16];
      u32x4 pw0, pw1;
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        MMV(dp, fb[i], vf[4 + i]);
        MG_SCHED_FENCE();
        p[2 * i] = (ABL == 5 ? s[2 * i] : __builtin_amdgcn_exp2f(s[2 * i] * sc2));               // raw v_exp_f32; masked -> 0
        p[2 * i + 1] = (ABL == 5 ? s[2 * i + 1] : __builtin_amdgcn_exp2f(s[2 * i + 1] * sc2));
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_row4x(fb, smem, dorow, 3);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        MMV(dp, fa[i], vf[8 + i]);
        MG_SCHED_FENCE();
        p[8 + 2 * i] = (ABL == 5 ? s[8 + 2 * i] : __builtin_amdgcn_exp2f(s[8 + 2 * i] * sc2));
        p[9 + 2 * i] = (ABL == 5 ? s[9 + 2 * i] : __builtin_amdgcn_exp2f(s[9 + 2 * i] * sc2));
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_t4(fa, dotp, 0, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < 3) MMV(dp, fb[i], vf[12 + i]); else MMVL(dp, fb[3], vf[15]);
        MG_SCHED_FENCE();
        pw0[i] = pack2bf(p[2 * i], p[2 * i + 1]);
        pw1[i] = pack2bf(p[8 + 2 * i], p[9 + 2 * i]);
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_t4(fb, dotp, 1, tx);
      bf16x8 pf0 = __builtin_bit_cast(bf16x8, pw0), pf1 = __builtin_bit_cast(bf16x8, pw1);
      mfma_operand_ready(pf0, pf1);
      MG_SCHED_FENCE();
      MG_STAMP(2);                    // DMA group 1 + phase 2
      if (SPREAD && more) issue_part(t + 1, sc ^ 1, 2);
      // ---- phase 3: dV^T += dO^T P (16 MFMAs) beside 16 dS = P o dP' (the 1/16 is applied once, in the epilogue) ----
      // MFMA order 0, 2, 1, 3: the two d-blocks of a burst alternate, no back-to-back pair on one accumulator
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); MMA(accv[j >> 1], fa[j], (j & 1) ? pf1 : pf0); }
      if (ABL != 2 || first) rd_t4(fa, dotp, 2, tx);
      // P re-read from its packed bf16 form (the value dV is multiplied with): 16 registers less across two phases
      const u32x4 q0w = __builtin_bit_cast(u32x4, pf0), q1w = __builtin_bit_cast(u32x4, pf1);
      float ds[16];
      u32x4 dw0, dw1;
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = ((i & 1) << 1) | (i >> 1);
        MMA(accv[2 + (j >> 1)], fb[j], (j & 1) ? pf1 : pf0);
        MG_SCHED_FENCE();
        ds[2 * i] = bflo(q0w[i]) * dp[2 * i];
        ds[2 * i + 1] = bfhi(q0w[i]) * dp[2 * i + 1];
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_t4(fb, dotp, 3, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = ((i & 1) << 1) | (i >> 1);
        MMA(accv[4 + (j >> 1)], fa[j], (j & 1) ? pf1 : pf0);
        MG_SCHED_FENCE();
        ds[8 + 2 * i] = bflo(q1w[i]) * dp[8 + 2 * i];
        ds[9 + 2 * i] = bfhi(q1w[i]) * dp[9 + 2 * i];
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_t4(fa, qtp, 0, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = ((i & 1) << 1) | (i >> 1);
        if (i < 3) MMA(accv[6 + (j >> 1)], fb[j], (j & 1) ? pf1 : pf0); else MMAL(accv[7], fb[3], pf1);
        MG_SCHED_FENCE();
        dw0[i] = pack2bf(ds[2 * i], ds[2 * i + 1]);
        dw1[i] = pack2bf(ds[8 + 2 * i], ds[9 + 2 * i]);
        MG_SCHED_FENCE();
      }
      if (ABL != 2 || first) rd_t4(fb, qtp, 1, tx);
      bf16x8 df0 = __builtin_bit_cast(bf16x8, dw0), df1 = __builtin_bit_cast(bf16x8, dw1);
      mfma_operand_ready(df0, df1);
      MG_SCHED_FENCE();
      MG_STAMP(3);                    // DMA group 2 + phase 3
      if (SPREAD && more) issue_part(t + 1, sc ^ 1, 3);
      // ---- phase 4: 16 dK^T += Q^T (16 dS) (16 MFMAs) ----
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); MMA(acck[j >> 1], fa[j], (j & 1) ? df1 : df0); }
      if (ABL != 2 || first) rd_t4(fa, qtp, 2, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); MMA(acck[2 + (j >> 1)], fb[j], (j & 1) ? df1 : df0); }
      if (ABL != 2 || first) rd_t4(fb, qtp, 3, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); MMA(acck[4 + (j >> 1)], fa[j], (j & 1) ? df1 : df0); }
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 3; ++i) { const int j = ((i & 1) << 1) | (i >> 1); MMA(acck[6 + (j >> 1)], fb[j], (j & 1) ? df1 : df0); }
      MMAL(acck[7], fb[3], df1);
    }
    MG_STAMP(4);                      // DMA group 3 + phase 4
    if constexpr (ABL == 6) tsum[5] += 1;
    sc ^= 1;
    first = false;
  }
#undef MG_STAMP
#ifdef MG_GEMM_ABLATIONS
  if constexpr (ABL == 6) {
    if (lane == 0 && blockIdx.x < 4096) {
#pragma unroll
      for (int i = 0; i < 6; ++i) g_attn_stamps[((size_t)blockIdx.x * 4 + wave) * 8 + i] = tsum[i];
    }
  }
#endif
  if constexpr (ABL == 1 || ABL == 4) MG_WAIT_VMCNT(0);
  // every wave is done with the ring (and no DMA is in flight: the last tile issues none) before it becomes staging space
  MG_BARRIER_KEEP_DMA();
  char* stage = smem + wave * (32 * EP_ROW);
  store_grad_tile32(gv, accv, 1.0f, stage, b, h, H, S, k0 + wave * 32, l31, hi);
  store_grad_tile32(gk, acck, 0.0625f, stage, b, h, H, S, k0 + wave * 32, l31, hi);
}


#undef MMV
#undef MMVL
#undef MMA
#undef MMAL

constexpr int Q_STAGE = 2 * ROW_TILE + T_TILE;   // K rows | V rows | K^T
constexpr int Q_STAGES = 3;

// ---------------------------------------------------------------------------
// dQ of 128 queries: 4 waves x 32 queries (lane&31 = query), KV tiles of 32 through a 3-stage ring (two tiles in flight).
// Transposed frame as in attn_bwd_dq_kernel: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T; the lane's statistics are two
// scalars.  48 MFMAs per tile step.
template <bool SPREAD>
__global__ __launch_bounds__(256) void attn_bwd_dq32_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ kt, const mg_bf16* __restrict__ dO, const float* __restrict__ ld2,
    const GradOut gq, int B, int H, int S, int ld_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;            // longest (latest) query blocks first
  const int qrow = qt0 + wave * 32 + l31, qrow_c = min(qrow, S - 1);
  const mg_bf16* kb = k + (int64_t)bh * S * DH;
  const mg_bf16* vb = v + (int64_t)bh * S * DH;
  const mg_bf16* ktb = kt + (int64_t)bh * DH * ld_t;
  const int dmodel = H * DH;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ (hi | (wave << 2))) << 4);
  const uint32_t tl = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ t_swz(lane >> 2)) << 4));
  auto issue_part = [&](int t, int buf, int i) {          // piece i of the three images of tile min(t, last)
    const int tc = min(t, ntiles - 1);
    const uint32_t st = smem_u + (uint32_t)(buf * Q_STAGE + (wave * 4 + i) * 1024);
    const uint32_t off = (uint32_t)min(tc * 32 + row0 + 2 * i, S - 1) * 512u + (c0b ^ (uint32_t)((i & 1) << 5));
    glds16su(kb, off, st);
    glds16su(vb, off, st + ROW_TILE);
    glds16su(ktb, (uint32_t)tc * (uint32_t)(DH * 64) + (uint32_t)((wave * 4 + i) * 1024) + tl, st + 2 * ROW_TILE);
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
  issue(0, 0);
  issue(1, 1);

  bf16x8 qf[16], dof[16];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + hi * 8;
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
  const int sw = row_swz(R);
  const uint32_t rb = row_base32(R, sw, hi);
  const int tx = hi ^ t_swz(l31);
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
    MG_WAIT_VMCNT(12);                // this wave's pieces of tile t have landed (tile t+1 may be in flight)
    MG_BARRIER_KEEP_DMA();            // tile t complete; everyone is done with tile t-1
    const int nb = sc == 0 ? 2 : sc - 1;
    if (SPREAD) issue_part(t + 2, nb, 0); else issue(t + 2, nb);
    const int kv0 = t * 32;
    {
      const char* st = smem + sc * Q_STAGE;
      const uint32_t krow = (uint32_t)(sc * Q_STAGE) + rb;
      const uint32_t vrow = krow + ROW_TILE;
      const char* ktp = st + 2 * ROW_TILE + l31 * 64;
      bf16x8 fa[4], fb[4];
      f32x16 s, dp;
      // ---- phase 1: S^T = K Q^T ----
      rd_row4x(fa, smem, krow, 0);
      rd_row4x(fb, smem, krow, 1);
      MG_SCHED_FENCE();
      mfma32v0(s, fa[0], qf[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) mfma32v(s, fa[i], qf[i]);
      rd_row4x(fa, smem, krow, 2);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fb[i], qf[4 + i]);
      rd_row4x(fb, smem, krow, 3);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma32v(s, fa[i], qf[8 + i]);
      rd_row4x(fa, smem, vrow, 0);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 3; ++i) mfma32v(s, fb[i], qf[12 + i]);
      mfma32v_last(s, fb[3], qf[15]);
      rd_row4x(fb, smem, vrow, 1);
      MG_SCHED_FENCE();
      if (SPREAD) issue_part(t + 2, nb, 1);
      // ---- phase 2: dP^T = V dO^T beside P^T = exp2(S^T sc2 - lse2) ----
      mfma32v0(dp, fa[0], dof[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) mfma32v(dp, fa[i], dof[i]);
      rd_row4x(fa, smem, vrow, 2);
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
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fb[i], dof[4 + i]);
        MG_SCHED_FENCE();
        p[2 * i] = __builtin_amdgcn_exp2f(fmaf(s[2 * i], sc2, nl2));
        p[2 * i + 1] = __builtin_amdgcn_exp2f(fmaf(s[2 * i + 1], sc2, nl2));
        MG_SCHED_FENCE();
      }
      rd_row4x(fb, smem, vrow, 3);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma32v(dp, fa[i], dof[8 + i]);
        MG_SCHED_FENCE();
        p[8 + 2 * i] = __builtin_amdgcn_exp2f(fmaf(s[8 + 2 * i], sc2, nl2));
        p[9 + 2 * i] = __builtin_amdgcn_exp2f(fmaf(s[9 + 2 * i], sc2, nl2));
        MG_SCHED_FENCE();
      }
      rd_t4(fa, ktp, 0, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 3; ++i) mfma32v(dp, fb[i], dof[12 + i]);
      mfma32v_last(dp, fb[3], dof[15]);
      rd_t4(fb, ktp, 1, tx);
      MG_SCHED_FENCE();
      if (SPREAD) issue_part(t + 2, nb, 2);
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
      // ---- phase 3: 16 dQ^T += K^T (16 dS^T) ----
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(accq[j >> 1], fa[j], (j & 1) ? df1 : df0); }
      rd_t4(fa, ktp, 2, tx);
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(accq[2 + (j >> 1)], fb[j], (j & 1) ? df1 : df0); }
      rd_t4(fb, ktp, 3, tx);
      MG_SCHED_FENCE();
      if (SPREAD) issue_part(t + 2, nb, 3);
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(accq[4 + (j >> 1)], fa[j], (j & 1) ? df1 : df0); }
      MG_SCHED_FENCE();
      MG_LGKM4();
#pragma unroll
      for (int i = 0; i < 3; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(accq[6 + (j >> 1)], fb[j], (j & 1) ? df1 : df0); }
      mfma32a_last(accq[7], fb[3], df1);
    }
    sc = sc == Q_STAGES - 1 ? 0 : sc + 1;
  }
  for (; t < ntiles; ++t) {           // tiles that only the later waves of the block need: move this wave's share of them
    MG_WAIT_VMCNT(12);
    MG_BARRIER_KEEP_DMA();
    issue(t + 2, sc == 0 ? 2 : sc - 1);
    sc = sc == Q_STAGES - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);                   // drain the ring's trailing loads before the ring becomes staging space
  MG_BARRIER_KEEP_DMA();
  store_grad_tile32(gq, accq, 0.0625f, smem + wave * (32 * EP_ROW), b, h, H, S, qt0 + wave * 32, l31, hi);
}

}  // namespace

int attn_bwd_dkdv32_launch(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt, const mg_bf16* dO,
                           const mg_bf16* dOt, const float* ld2, const GradOut& gk, const GradOut& gv, int B, int H, int S,
                           int ld_t, int variant, hipStream_t s, const char* who) {
  const int lds = KV_STAGES * KV_STAGE;
  const dim3 grid((unsigned)(((S + 127) / 128) * B * H));
#ifdef MG_GEMM_ABLATIONS
  {
    const char* e = getenv("MAGMA_ATTN_BWD32_ABL");
    const int abl = e ? atoi(e) : 0;
#define MG_ABL(N_)                                                                                                         \
    if (abl == N_) {                                                                                                       \
      if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv32_kernel<true, N_>, lds, who)) return rc;                 \
      hipLaunchKernelGGL((attn_bwd_dkdv32_kernel<true, N_>), grid, dim3(256), lds, s, q, k, v, qt, dO, dOt, ld2, gk, gv, B, H, S, ld_t); \
      MG_CHECK_LAUNCH();                                                                                                   \
      return MG_OK;                                                                                                        \
    }
    MG_ABL(1) MG_ABL(2) MG_ABL(3) MG_ABL(4) MG_ABL(5) MG_ABL(6)
#undef MG_ABL
  }
#endif
  if (variant == 1) {
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv32_kernel<false>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv32_kernel<false>, grid, dim3(256), lds, s, q, k, v, qt, dO, dOt, ld2, gk, gv, B, H, S, ld_t);
  } else {
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dkdv32_kernel<true>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv32_kernel<true>, grid, dim3(256), lds, s, q, k, v, qt, dO, dOt, ld2, gk, gv, B, H, S, ld_t);
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

int attn_bwd_dq32_launch(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* kt, const mg_bf16* dO,
                         const float* ld2, const GradOut& gq, int B, int H, int S, int ld_t, int variant, hipStream_t s,
                         const char* who) {
  const int lds = Q_STAGES * Q_STAGE;
  const dim3 grid((unsigned)(((S + 127) / 128) * B * H));
  if (variant == 3) {
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dq32_kernel<false>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq32_kernel<false>, grid, dim3(256), lds, s, q, k, v, kt, dO, ld2, gq, B, H, S, ld_t);
  } else {
    if (int rc = mg_allow_dynamic_lds((const void*)attn_bwd_dq32_kernel<true>, lds, who)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq32_kernel<true>, grid, dim3(256), lds, s, q, k, v, kt, dO, ld2, gq, B, H, S, ld_t);
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}
