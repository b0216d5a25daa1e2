// attention_fwd32.hip -- causal flash-attention forward, head dim 256, on 32-query waves (gfx950, round 5).
//
// Same arithmetic as attn_prefill_sp_kernel (attention.hip): both products transposed (lane & 31 = query, every per-query
// scalar lane-local), fp32 online softmax in the exp2 domain with the deferred running maximum (guide T13), the softmax of a
// tile split in two halves that sit beside MFMA bursts of OTHER tiles.  What changes is the wave: 32 queries on
// v_mfma_f32_32x32x16_bf16 instead of 16 on the 16x16x32 form -- every K / V^T fragment read from LDS feeds twice the matrix
// work, and the softmax instructions per MFMA cycle halve (the 16-query kernel is bound by exactly those two: DESIGN.md 4,
// profiles/r04_attention_fwd_ablations.txt) -- ALONE on its SIMD (4 waves x 32 queries per workgroup): the 128 O^T
// accumulator registers live in the AGPR half of the file, pinned there by asm MFMA statements (attn32_device.h; the round-2
// attempt at this shape left the allocation to hipcc and measured 1.10 ms against 0.95).
//
//   iteration t:  [ K(t+1) reads | S^T(t+1) = K(t+1) Q^T : 16 MFMAs  ||  exp2 / row sums / pack of tile t between them ]
//                 [ rescale of O^T when the running maximum moved (rare) ]
//                 [ V^T(t) reads | O^T += V^T(t) P^T(t)   : 16 MFMAs  ||  row max / running max / alpha of tile t+1     ]
// K rows and V^T tiles stream through a 4-stage LDS ring filled by LDS-DMA, counted vmcnt + one barrier per tile.
#include "attn32_device.h"
#include <stdlib.h>

namespace {

constexpr int F_STAGE = ROW_TILE + T_TILE;   // K rows | V^T
constexpr int F_STAGES = 4;

MG_DEV float pair_max(float x) {             // over the two lanes {l, l ^ 32} that hold the two halves of a query's keys
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
MG_DEV float pair_sum(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

__global__ __launch_bounds__(256) void attn_prefill32_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ kcache, const mg_bf16* __restrict__ vt,
    mg_bf16* __restrict__ out, int64_t ld_out, float* __restrict__ lse, int B, int H, int S, int Smax, int vt_ld, float defer) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;   // longest blocks first
  const int qrow = qt0 + wave * 32 + l31, qrow_c = min(qrow, S - 1);
  const mg_bf16* kbase = kcache + (int64_t)bh * Smax * DH;
  const mg_bf16* vbase = vt + (int64_t)bh * DH * vt_ld;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  const int row0 = wave * 8 + hi;
  const uint32_t c0b = (uint32_t)((l31 ^ (hi | (wave << 2))) << 4);
  const uint32_t tl = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ t_swz(lane >> 2)) << 4));
  // piece i (0..3) of the two images of tile min(t, last): past the last tile the ring re-loads it (in bounds, never read)
  // so that the wait counts stay constant
  auto issue_part = [&](int t, int buf, int i) {
    const int tc = min(t, ntiles - 1);
    const uint32_t st = smem_u + (uint32_t)(buf * F_STAGE + (wave * 4 + i) * 1024);
    glds16su(kbase, (uint32_t)min(tc * 32 + row0 + 2 * i, S - 1) * 512u + (c0b ^ (uint32_t)((i & 1) << 5)), st);
    glds16su(vbase, (uint32_t)tc * (uint32_t)(DH * 64) + (uint32_t)((wave * 4 + i) * 1024) + tl, st + ROW_TILE);
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
#pragma unroll
  for (int i = 0; i < F_STAGES - 1; ++i) issue(i, i);

  bf16x8 qf[16];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + hi * 8;
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
  const int sw = row_swz(R);
  const int tx = hi ^ t_swz(l31);
  const uint32_t rb = row_base32(R, sw, hi);
  const int lim0 = min(qrow, S - 1) - hi * 8;     // key (r >> 3) 16 + (r & 7) of tile kv0 is visible iff it is <= lim0 - kv0
  // ONE accumulator-file copy of the Q fragments, made here (this also retires the ordinary loads in hipcc's scoreboard before
  // the pipelined loop, see attention.hip); every MFMA takes it from there.  Handing the loaded values themselves to the "a"
  // operands lets hipcc keep one AGPR copy per unrolled loop body -- 128 registers, and the third spills to scratch, whose
  // reload inside the loop waits for vmcnt(0) = the DMA ring.
  bf16x8 qa[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) asm volatile("" : "=a"(qa[ks]) : "0"(qf[ks]));

  bf16x8 fa[4], fb[4];
  f32x16 sA, sB;                       // scores of the current / the next tile, swapping roles every iteration (no copies)
  float alpha;
  // part 1 of the softmax of the tile whose scores are in sn: (mask,) row maximum, running maximum, rescale factor
  auto part1 = [&](f32x16& sn, int kv0) {
    // only the tiles that straddle this wave's queries (and the ragged last tile) need the mask
    if (kv0 + 31 > my_first || kv0 + 32 > S) {
      const int lim = lim0 - kv0;
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[r] = ((r >> 3) * 16 + (r & 7)) > lim ? -1e30f : sn[r];
    }
    float tmax = sn[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sn[r]);
    tmax = pair_max(tmax);
    const float cand = tmax * sc2;
    const float mnew = (cand > m2 + defer) ? cand : m2;
    alpha = __builtin_amdgcn_exp2f(m2 - mnew);
    m2 = mnew;
  };

  // ---- prologue: tiles 0 and 1 landed; S^T(0) and part 1 of its softmax ----
  MG_WAIT_VMCNT(8);
  MG_BARRIER_KEEP_DMA();
  {
    const uint32_t krow = rb;
    rd_row4x(fa, smem, krow, 0);
    rd_row4x(fb, smem, krow, 1);
    MG_SCHED_FENCE();
    mfma32v0_ba(sA, fa[0], qa[0]);
#pragma unroll
    for (int i = 1; i < 4; ++i) mfma32v_ba(sA, fa[i], qa[i]);
    rd_row4x(fa, smem, krow, 2);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) mfma32v_ba(sA, fb[i], qa[4 + i]);
    rd_row4x(fb, smem, krow, 3);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) mfma32v_ba(sA, fa[i], qa[8 + i]);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 3; ++i) mfma32v_ba(sA, fb[i], qa[12 + i]);
    mfma32v_ba_last(sA, fb[3], qa[15]);
    part1(sA, 0);
  }
  int sc = 0;
  // one iteration: `cur` = masked scores of tile t (part 1 done), `nxt` receives S^T(t+1)
  auto iteration = [&](f32x16& cur, f32x16& nxt, int t) {
    MG_WAIT_VMCNT(8);                 // this wave's pieces of tile t+1 landed (tile t+2 may be in flight)
    MG_BARRIER_KEEP_DMA();            // tile t+1 complete; everyone is done with iteration t-1 (K(t), V^T(t-1))
    const int nb = sc == 0 ? F_STAGES - 1 : sc - 1;
    issue_part(t + F_STAGES - 1, nb, 0);
    issue_part(t + F_STAGES - 1, nb, 1);
    const int scn = sc == F_STAGES - 1 ? 0 : sc + 1;
    // ---- block A: S^T(t+1) MFMAs; behind each of them one exponential of tile t (row sum and bf16 pack as they come) ----
    // (tile t+1 exists in the ring even past the last tile: issue_part clamps to it; its scores are then fully masked)
    const uint32_t krow = (uint32_t)(scn * F_STAGE) + rb;
    const char* vtp = smem + sc * F_STAGE + ROW_TILE + l31 * 64;
    float psum = 0.f, pe = 0.f;
    u32x4 pw0, pw1;
    auto soft = [&](int r) {          // element r of tile t (r even: kept for the pack with r + 1)
      const float pr = __builtin_amdgcn_exp2f(fmaf(cur[r], sc2, -m2));
      psum += pr;
      if (r & 1) { if (r < 8) pw0[r >> 1] = pack2bf(pe, pr); else pw1[(r - 8) >> 1] = pack2bf(pe, pr); }
      else pe = pr;
    };
    rd_row4x(fa, smem, krow, 0);
    rd_row4x(fb, smem, krow, 1);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 0) mfma32v0_ba(nxt, fa[0], qa[0]); else mfma32v_ba(nxt, fa[i], qa[i]);
      MG_SCHED_FENCE();
      soft(i);
      MG_SCHED_FENCE();
    }
    rd_row4x(fa, smem, krow, 2);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma32v_ba(nxt, fb[i], qa[4 + i]);
      MG_SCHED_FENCE();
      soft(4 + i);
      MG_SCHED_FENCE();
    }
    rd_row4x(fb, smem, krow, 3);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma32v_ba(nxt, fa[i], qa[8 + i]);
      MG_SCHED_FENCE();
      soft(8 + i);
      MG_SCHED_FENCE();
    }
    rd_t4(fa, vtp, 0, tx);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3) mfma32v_ba(nxt, fb[i], qa[12 + i]); else mfma32v_ba_last(nxt, fb[3], qa[15]);
      MG_SCHED_FENCE();
      soft(12 + i);
      MG_SCHED_FENCE();
    }
    rd_t4(fb, vtp, 1, tx);
    lsum = lsum * alpha + psum;
    bf16x8 pf0 = __builtin_bit_cast(bf16x8, pw0), pf1 = __builtin_bit_cast(bf16x8, pw1);
    mfma_operand_ready(pf0, pf1);
    MG_SCHED_FENCE();
    issue_part(t + F_STAGES - 1, nb, 2);
    issue_part(t + F_STAGES - 1, nb, 3);
    // ---- the running maximum moved by more than the deferral threshold (rare): O^T and l were kept at the old one ----
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
      // (a volatile statement on the tuples first: without it hipcc hoists the first accumulator reads of this block ABOVE the
      //  branch, right behind the PV MFMAs of the iteration before -- inside their result latency, which nothing pads for an
      //  asm MFMA; with it every read of O^T sits in here, a whole S^T phase behind the last write)
      asm volatile("" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]), "+a"(o[4]), "+a"(o[5]), "+a"(o[6]), "+a"(o[7]));
      // one 16-register tuple at a time (fenced): all 128 at once would need 128 VGPRs beside the resident Q fragments
#pragma unroll
      for (int db = 0; db < 8; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        asm volatile("" : "+a"(o[db]));      // back in its AGPRs before the next tuple is touched
        MG_SCHED_FENCE();
      }
    }
    MG_SCHED_FENCE();
    // ---- block B: O^T += V^T(t) P^T(t), 16 MFMAs; part 1 of softmax(t+1) behind the first burst ----
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(o[j >> 1], fa[j], (j & 1) ? pf1 : pf0); }
    rd_t4(fa, vtp, 2, tx);
    MG_SCHED_FENCE();
    part1(nxt, (t + 1) * 32);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(o[2 + (j >> 1)], fb[j], (j & 1) ? pf1 : pf0); }
    rd_t4(fb, vtp, 3, tx);
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(o[4 + (j >> 1)], fa[j], (j & 1) ? pf1 : pf0); }
    MG_SCHED_FENCE();
    MG_LGKM4();
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int j = ((i & 1) << 1) | (i >> 1); mfma32a(o[6 + (j >> 1)], fb[j], (j & 1) ? pf1 : pf0); }
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
    issue(t + F_STAGES - 1, sc == 0 ? F_STAGES - 1 : sc - 1);
    sc = sc == F_STAGES - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);                   // drain the ring's trailing loads before the ring becomes staging space
  MG_BARRIER_KEEP_DMA();
  lsum = pair_sum(lsum);
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

}  // namespace

int attn_prefill32_launch(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out, int64_t ld_out, float* lse,
                          int B, int H, int S, int Smax, int vt_ld, float defer, hipStream_t s, const char* who) {
  const int lds = F_STAGES * F_STAGE;
  if (ld_out & 7) MG_FAIL(MG_ERR_SHAPE, "%s: the 32-query kernel stores 16-byte pieces: ld_out %% 8 == 0", who);
  if (int rc = mg_allow_dynamic_lds((const void*)attn_prefill32_kernel, lds, who)) return rc;
  hipLaunchKernelGGL(attn_prefill32_kernel, dim3((unsigned)(((S + 127) / 128) * B * H)), dim3(256), lds, s, q, kcache, vt, out, ld_out,
                     lse, B, H, S, Smax, vt_ld, defer);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
