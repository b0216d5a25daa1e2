// attention_fwd32_fp8.hip -- causal flash-attention forward, head dim 256, QK^T and PV on the fp8 MFMA (gfx950, round 5).
// BASELINE.json config[4]: "fp8 MFMA path for GPT-J attention".
//
// The 32-query-wave skeleton of attention_fwd32.hip (lane & 31 = query, O^T and Q resident in AGPRs, one wave per SIMD, fp32
// online softmax with the deferred running maximum) on v_mfma_scale_f32_32x32x64_f8f6f4: OCP e4m3 operands with E8M0 block
// scales, 64 contraction elements per instruction at twice the bf16 rate.  Per KV tile of 64 keys and wave:
//   S^T[key][q] = K Q^T      2 key blocks x 4 steps of 64 d  =  8 MFMAs  (K rows / Q: one power-of-two scale per TOKEN)
//   O^T[d][q] += V^T P^T     8 d blocks x 1 step of 64 keys  =  8 MFMAs  (V^T: one E8M0 per (d, 32 keys); P: e4m3(16 p), scale 2^-4)
// i.e. a quarter of the MFMA instructions, half the matrix-pipe time, half the LDS bytes and DMA pieces per key of the bf16 kernel,
// and the scales cost no VALU instruction (operands written by mg_rotary_split_fp8, attention.hip: rotary_split_fp8_kernel).
// The lane's 32 probabilities of a tile -- S^T accumulator registers of the two 32-key blocks -- ARE its 32 operand bytes of the
// PV product: V^T tiles are stored with the keys in that order (byte 32 hi + 16 b + r <-> key 32 b + (r & 3) + 8 (r >> 2) + 4 hi),
// which is also the instruction's own block structure (block b = bytes 16 b .. + 15 of both half-wave lanes; scale from lane
// row + 32 b: tools/probes/mx32_probe.hip).  P <= 2^4 by the deferral threshold (4 in log2 units), so 16 p <= 256 < 448.
#include "attn32_device.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;

namespace {

constexpr int P_KROWS = 64 * 256;                       // K rows e4m3 [64 keys][256 B]
constexpr int P_VT = 256 * 64;                          // V^T e4m3 [256 d][64 keys]
constexpr int P_SCALES = 1024;                          // ek of 256 keys from kv0 (64 used) | sv8 [2][32][8] | pad
constexpr int P_STAGE = P_KROWS + P_VT + P_SCALES;      // 33792 = 66 x 512
constexpr int P_STAGES = 4;
static_assert(P_STAGE % 512 == 0, "fragment addresses are formed with XOR below bit 9");
constexpr float P_DEFER = 4.0f;                         // log2 units: P <= 16

// asm MFMA statements as in attn32_device.h (accumulator class spelled out, the last one of a chain carries its wait states:
// 16-pass op -> 18+).  sa / sb: the scale dwords (byte 0 is used: op_sel 0).
MG_DEV void mx_v0(f32x16& c, const i32x8 a, const i32x8 b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=&v"(c) : "v"(a), "a"(b), "v"(sa), "v"(sb));
}
MG_DEV void mx_v(f32x16& c, const i32x8 a, const i32x8 b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "a"(b), "v"(sa), "v"(sb));
}
MG_DEV void mx_v_last(f32x16& c, const i32x8 a, const i32x8 b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 3"
               : "+v"(c) : "v"(a), "a"(b), "v"(sa), "v"(sb));
}
MG_DEV void mx_a(f32x16& c, const i32x8 a, const i32x8 b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
MG_DEV void mx_a_last(f32x16& c, const i32x8 a, const i32x8 b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 3"
               : "+a"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
MG_DEV float pair_max8(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
MG_DEV float pair_sum8(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// 32 operand bytes of a lane = two 16-byte chunks p and p ^ 1 of an LDS row image (addresses a and a ^ 16)
MG_DEV i32x8 rd32(const char* lds, uint32_t a) {
  const u32x4 lo = *(const u32x4*)(lds + a), hi = *(const u32x4*)(lds + (a ^ 16u));
  return (i32x8){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

__global__ __launch_bounds__(256) void attn_prefill32_fp8_kernel(
    const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8, const uint8_t* __restrict__ v8t, const uint8_t* __restrict__ eq,
    const uint8_t* __restrict__ ek, const uint8_t* __restrict__ sv8, mg_bf16* __restrict__ out, int64_t ld_out, float* __restrict__ lse,
    int B, int H, int S, int Sp, uint8_t* __restrict__ out8, int64_t ld_out8, uint8_t* __restrict__ out8_scales) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;   // longest blocks first
  const int qrow = qt0 + wave * 32 + l31, qrow_c = min(qrow, S - 1);
  const int nt64 = (S + 63) >> 6;
  const uint8_t* kbase = k8 + (int64_t)bh * S * DH;
  const uint8_t* vbase = v8t + (int64_t)bh * nt64 * (DH * 64);
  const uint8_t* ekb = ek + (int64_t)bh * Sp;
  const uint8_t* svb = sv8 + (int64_t)bh * nt64 * 512;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 63) >> 6;
  const uint32_t smem_u = lds_u32(smem);
  // K rows: a 1-KiB piece = 4 rows of 256 B; wave w moves pieces 4w .. 4w+3 (rows 16 w + 4 i + (lane >> 4)); 16-byte chunk c of
  // row r sits at position c ^ row_swz(r & 31) -- between the pieces of a wave only bit 2 of the swizzle changes ((i >> 1) << 2)
  const int krow0 = wave * 16 + (lane >> 4);
  const uint32_t kc0 = (uint32_t)(((lane & 15) ^ ((lane >> 4) | ((2 * (wave & 1)) << 2))) << 4);
  const uint32_t tl = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ t_swz(lane >> 2)) << 4));
  auto issue_part = [&](int t, int buf, int i) {
    const int tc = min(t, ntiles - 1);
    const uint32_t st = smem_u + (uint32_t)(buf * P_STAGE + (wave * 4 + i) * 1024);
    glds16su(kbase, (uint32_t)min(tc * 64 + krow0 + 4 * i, S - 1) * 256u + (kc0 ^ (uint32_t)((i >> 1) << 6)), st);
    glds16su(vbase, (uint32_t)tc * (uint32_t)(DH * 64) + (uint32_t)((wave * 4 + i) * 1024) + tl, st + P_KROWS);
    if (i == 0) {       // scales of the tile (every wave writes the same bytes): 256 B of per-key exponents from key tc * 64, 512 B of V^T scales
      const uint32_t sb = smem_u + (uint32_t)(buf * P_STAGE + P_KROWS + P_VT);
      glds4su(ekb, (uint32_t)(tc * 64) + (uint32_t)lane * 4u, sb);
      glds4su(svb, (uint32_t)tc * 512u + (uint32_t)lane * 4u, sb + 256);
      glds4su(svb, (uint32_t)tc * 512u + 256u + (uint32_t)lane * 4u, sb + 512);
    }
  };
  auto issue = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(t, buf, i);
  };
#pragma unroll
  for (int i = 0; i < P_STAGES - 1; ++i) issue(i, i);

  // Q: 32 bytes per 64-d step and lane, one AGPR copy (see attention_fwd32.hip)
  i32x8 qa[4];
  {
    const uint8_t* qp = q8 + ((int64_t)bh * S + qrow_c) * DH + hi * 32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 lo = *(const u32x4*)(qp + ks * 64), hi_ = *(const u32x4*)(qp + ks * 64 + 16);
      const i32x8 f = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]};
      asm volatile("" : "=a"(qa[ks]) : "0"(f));
    }
  }
  const int sq = (int)eq[(int64_t)bh * Sp + qrow_c];        // this query's E8M0 (byte 0 of the scale dword)
  f32x16 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
  }
  float m2 = -1e30f, lsum = 0.f;
  const float sc2 = 0.0625f * 1.4426950408889634f;  // 1/sqrt(256) * log2(e)
  const int my_first = qt0 + wave * 32;
  const int n_act = min(ntiles, ((my_first + 31) >> 6) + 1);   // this wave's 64-key tiles
  const int sw = row_swz(l31);
  // K fragment of key block kb, step ks: row kb*32 + l31, chunks (ks*4 + hi*2) ^ sw and its neighbour -> base ^ (ks << 6)
  const uint32_t kb0 = (uint32_t)(l31 * 256 + ((((hi << 1) ^ sw) & 15) << 4));
  // V^T fragment of d block db: row db*32 + l31 (64-byte rows), chunks (2 hi) ^ tsw and its neighbour
  const uint32_t vb0 = (uint32_t)(P_KROWS + l31 * 64 + ((((hi << 1) ^ t_swz(l31)) & 3) << 4));
  const int lim0 = min(qrow, S - 1) - hi * 4;       // key 32 b + (r & 3) + 8 (r >> 2) of tile kv0 is visible iff it is <= lim0 - kv0
  const int unit_p = 0x7B7B7B7B;                    // 2^-4: P is stored as e4m3(16 p)
  asm volatile("" ::"v"(sq));        // retire the ordinary loads in hipcc's scoreboard before the pipelined loop (see attention.hip)

  f32x16 sA[2], sB[2];
  float alpha;
  auto part1 = [&](f32x16 (&sn)[2], int kv0) {
    if (kv0 + 63 > my_first || kv0 + 64 > S) {
      const int lim = lim0 - kv0;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[kb][r] = (32 * kb + (r & 3) + 8 * (r >> 2)) > lim ? -1e30f : sn[kb][r];
    }
    float tmax = sn[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sn[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sn[1][r]);
    tmax = pair_max8(tmax);
    const float cand = tmax * sc2;
    const float mnew = (cand > m2 + P_DEFER) ? cand : m2;
    alpha = __builtin_amdgcn_exp2f(m2 - mnew);
    m2 = mnew;
  };
  // S^T of the tile in stage `st`: 8 MFMAs; `between(i)` runs behind MFMA i
  auto scores = [&](f32x16 (&sn)[2], int st, auto&& between) {
    const char* base = smem + st * P_STAGE;
    const char* sc_ = base + P_KROWS + P_VT;
    i32x8 fa[4], fb[4];
    const int ek0 = (int)(uint8_t)sc_[l31], ek1 = (int)(uint8_t)sc_[32 + l31];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = rd32(base, kb0 ^ (uint32_t)(ks << 6));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = rd32(base, (kb0 + 32 * 256) ^ (uint32_t)(ks << 6));
    MG_SCHED_FENCE();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks == 0) mx_v0(sn[0], fa[0], qa[0], ek0, sq); else if (ks < 3) mx_v(sn[0], fa[ks], qa[ks], ek0, sq); else mx_v_last(sn[0], fa[3], qa[3], ek0, sq);
      MG_SCHED_FENCE();
      between(ks);
      MG_SCHED_FENCE();
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks == 0) mx_v0(sn[1], fb[0], qa[0], ek1, sq); else if (ks < 3) mx_v(sn[1], fb[ks], qa[ks], ek1, sq); else mx_v_last(sn[1], fb[3], qa[3], ek1, sq);
      MG_SCHED_FENCE();
      between(4 + ks);
      MG_SCHED_FENCE();
    }
  };

  // ---- prologue: tiles 0 and 1 landed; S^T(0) and part 1 of its softmax ----
  MG_WAIT_VMCNT(11);
  MG_BARRIER_KEEP_DMA();
  scores(sA, 0, [](int) {});
  part1(sA, 0);

  int sc = 0;
  auto iteration = [&](f32x16 (&cur)[2], f32x16 (&nxt)[2], int t) {
    MG_WAIT_VMCNT(11);                // this wave's pieces of tile t+1 landed (tile t+2's 11 may be in flight)
    MG_BARRIER_KEEP_DMA();            // tile t+1 complete; everyone is done with iteration t-1
    const int nb = sc == 0 ? P_STAGES - 1 : sc - 1;
    issue_part(t + P_STAGES - 1, nb, 0);
    issue_part(t + P_STAGES - 1, nb, 1);
    const int scn = sc == P_STAGES - 1 ? 0 : sc + 1;
    // ---- block A: S^T(t+1) MFMAs; behind each of them four exponentials of tile t (row sum and e4m3 pack as they come) ----
    float psum = 0.f;
    int pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto soft4 = [&](int g) {         // elements 4 g .. 4 g + 3 of the lane's 32 (block g >> 2, registers (g & 3) * 4 ..): one operand dword
      const int kb = g >> 2, r0 = (g & 3) * 4;
      float pr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { pr[j] = __builtin_amdgcn_exp2f(fmaf(cur[kb][r0 + j], sc2, -m2)); psum += pr[j]; }
      pw[g] = __builtin_amdgcn_cvt_pk_fp8_f32(pr[0] * 16.f, pr[1] * 16.f, pw[g], false);
      pw[g] = __builtin_amdgcn_cvt_pk_fp8_f32(pr[2] * 16.f, pr[3] * 16.f, pw[g], true);
    };
    scores(nxt, scn, [&](int i) { soft4(i); });
    lsum = lsum * alpha + psum;
    i32x8 pf = {pw[0], pw[1], pw[2], pw[3], pw[4], pw[5], pw[6], pw[7]};
    asm volatile("s_nop 3" : "+v"(pf));
    MG_SCHED_FENCE();
    issue_part(t + P_STAGES - 1, nb, 2);
    issue_part(t + P_STAGES - 1, nb, 3);
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
      asm volatile("" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]), "+a"(o[4]), "+a"(o[5]), "+a"(o[6]), "+a"(o[7]));
#pragma unroll
      for (int db = 0; db < 8; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        asm volatile("" : "+a"(o[db]));
        MG_SCHED_FENCE();
      }
    }
    MG_SCHED_FENCE();
    // ---- block B: O^T += V^T(t) P^T(t), 8 MFMAs; part 1 of softmax(t+1) behind the first four ----
    {
      const char* base = smem + sc * P_STAGE;
      // this lane's eight V^T scale bytes (d blocks 0..7 of row l31, key block hi): [hi][l31][db]
      const u32x2 svw = *(const u32x2*)(base + P_KROWS + P_VT + 256 + hi * 256 + l31 * 8);
      int svs[8];                     // byte 0 of each dword = the scale of one d block (op_sel 0); formed well ahead of the MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) svs[i] = (int)(svw[i >> 2] >> (8 * (i & 3)));
      MG_SCHED_FENCE();
      i32x8 va[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) va[i] = rd32(base, vb0 + (uint32_t)(i * 2048));
#pragma unroll
      for (int i = 0; i < 4; ++i) vb[i] = rd32(base, vb0 + (uint32_t)((4 + i) * 2048));
      MG_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 4; ++i) mx_a(o[i], va[i], pf, svs[i], unit_p);
      MG_SCHED_FENCE();
      part1(nxt, (t + 1) * 64);
      MG_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < 3; ++i) mx_a(o[4 + i], vb[i], pf, svs[4 + i], unit_p);
      mx_a_last(o[7], vb[3], pf, svs[7], unit_p);
    }
    sc = scn;
  };
  int t = 0;
  for (; t + 1 < n_act; t += 2) {
    iteration(sA, sB, t);
    iteration(sB, sA, t + 1);
  }
  if (t < n_act) { iteration(sA, sB, t); ++t; }
  for (; t < ntiles; ++t) {           // tiles that only the later waves of the block need: move this wave's share of them
    MG_WAIT_VMCNT(11);
    MG_BARRIER_KEEP_DMA();
    issue(t + P_STAGES - 1, sc == 0 ? P_STAGES - 1 : sc - 1);
    sc = sc == P_STAGES - 1 ? 0 : sc + 1;
  }
  MG_WAIT_VMCNT(0);
  MG_BARRIER_KEEP_DMA();
  lsum = pair_sum8(lsum);
  const float inv = 1.0f / lsum;
  char* stage = smem + wave * (32 * EP_ROW);
  char* wr = stage + l31 * EP_ROW + hi * 8;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const u32x2 w = {pack2bf(o[db][rq * 4] * inv, o[db][rq * 4 + 1] * inv), pack2bf(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv)};
      *(u32x2*)(wr + db * 64 + rq * 16) = w;
    }
    MG_SCHED_FENCE();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 2 + hi;
    const u32x4 w = *(const u32x4*)(stage + row * EP_ROW + l31 * 16);
    const int s = qt0 + wave * 32 + row;
    if (s < S) {
      *(u32x4*)(out + (int64_t)(b * S + s) * ld_out + h * DH + l31 * 8) = w;
      // the OCP MX e4m3 copy of the same row piece (the operand of the out_proj MX GEMM): the lane holds 8 consecutive columns, the
      // four lanes of a 32-column block are consecutive -- mg_quantize_mx_fp8's result bit for bit, without its pass over ctx
      if (out8) mx_emit8(w, out8 + (int64_t)(b * S + s) * ld_out8, out8_scales, (B * S + 63) >> 6, b * S + s, h * DH + l31 * 8);
    }
  }
  if (lse && hi == 0 && qrow < S) lse[(int64_t)bh * S + qrow] = (m2 + log2f(lsum)) * 0.6931471805599453f;
}

}  // namespace

// q8, k8 [B,H,S,256] e4m3; v8t [B,H,ceil(S/64),256,64] e4m3; eq, ek [B,H,Sp] E8M0 bytes (Sp = mg_attn_fp8_scale_stride(S));
// sv8 [B,H,ceil(S/64),512] E8M0 bytes: all as written by mg_rotary_split_fp8.  out [B*S, >= H*256] bf16 (row stride ld_out,
// a multiple of 8), lse [B,H,S] fp32 or NULL -- the outputs of mg_attn_prefill_bf16.
extern "C" int mg_attn_prefill_fp8(const uint8_t* q8, const uint8_t* k8, const uint8_t* v8t, const uint8_t* eq, const uint8_t* ek,
                                   const uint8_t* sv8, mg_bf16* out, int64_t ld_out, float* lse, int32_t B, int32_t H, int32_t S,
                                   uint8_t* out8, int64_t ld_out8, uint8_t* out8_scales, void* stream) {
  if (ld_out == 0) ld_out = (int64_t)H * DH;
  if (ld_out < (int64_t)H * DH || (ld_out & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_fp8: ld_out must be 0 or a multiple of 8 >= H*256");
  if (B <= 0 || H <= 0 || S <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_fp8: bad B/H/S");
  if (!q8 || !k8 || !v8t || !eq || !ek || !sv8 || !out) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_fp8: null pointer");
  if (!MG_ALIGNED16(q8) || !MG_ALIGNED16(k8) || !MG_ALIGNED16(v8t) || !MG_ALIGNED16(sv8) || !MG_ALIGNED16(out) || ((uintptr_t)ek & 3))
    MG_FAIL(MG_ERR_ALIGN, "mg_attn_prefill_fp8: pointers must be 16-byte aligned (ek: 4)");
  if (out8 && (!out8_scales || ((uintptr_t)out8 & 7) || ld_out8 != (((int64_t)H * DH + 127) / 128) * 128))
    MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_fp8: the MX output copy needs its scale array, 8-byte alignment and ld_out8 == ceil(H*256 / 128) * 128");
  const int lds = P_STAGES * P_STAGE;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_prefill32_fp8_kernel, lds, "mg_attn_prefill_fp8")) return rc;
  hipLaunchKernelGGL(attn_prefill32_fp8_kernel, dim3((unsigned)(((S + 127) / 128) * B * H)), dim3(256), lds, (hipStream_t)stream, q8, k8,
                     v8t, eq, ek, sv8, out, ld_out, lse, B, H, S, ((S + 63) / 64) * 64 + 256, out8, ld_out8, out8_scales);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
