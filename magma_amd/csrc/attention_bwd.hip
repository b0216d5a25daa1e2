// attention_bwd.hip -- causal flash-attention backward, head dim 256 (gfx950).
//
// Two kernels that recompute P from (q, k, lse) -- no S x S tensor in HBM:
//
//   attn_bwd_dq_kernel    one workgroup per 64 queries (4 waves x 16), loops over
//                         KV tiles of 32.  Works in the transposed frame of the
//                         forward kernel (lane&15 = query):
//                           S^T  = K Q^T        dP^T = V dO^T
//                           dS^T = P^T o (dP^T - D) / 16
//                           dQ^T += K^T dS^T
//   attn_bwd_dkdv_kernel  one workgroup per 64 keys (4 waves x 16, lane&15 = key),
//                         loops over query tiles of 32 from the diagonal down:
//                           S  = Q K^T          dP = dO V^T
//                           dV^T += dO^T P      dK^T += Q^T dS
//
// MFMA contracts over 8 consecutive elements per lane, so every operand is
// needed with its contraction index contiguous: K,V,Q,dO row-major [s][256] for
// the d-contractions and K^T,Q^T,dO^T [256][s] for the s-contractions (made by
// mg_head_transpose_bf16).  The same row permutation as in the forward kernel
// turns accumulator registers directly into the next product's operand.
// D[b,h,q] = sum_d dO*O comes from attn_bwd_prep_kernel.
#include "common.h"

namespace {

constexpr int DH = 256;
constexpr int ROW_STRIDE = DH * 2 + 16;  // [32][256] tiles, padded rows (528 B)
constexpr int T_STRIDE = 32 * 2 + 16;    // [256][32] tiles, padded rows (80 B)
constexpr int ROW_TILE = 32 * ROW_STRIDE;  // 16896
constexpr int T_TILE = DH * T_STRIDE;      // 20480

// D = rowsum(dO o O); one wave per (b, s, h) row of 256
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const mg_bf16* __restrict__ dO,
                                                            const mg_bf16* __restrict__ O, float* __restrict__ D,
                                                            int B, int H, int S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // over B*S*H, (b,s,h) order = memory order of [M, H*256]
  if (row >= (int64_t)B * S * H) return;
  const u32x2 a = *(const u32x2*)(dO + row * DH + lane * 4);
  const u32x2 o = *(const u32x2*)(O + row * DH + lane * 4);
  float s = bflo(a[0]) * bflo(o[0]) + bfhi(a[0]) * bfhi(o[0]) + bflo(a[1]) * bflo(o[1]) + bfhi(a[1]) * bfhi(o[1]);
  s = wave_sum(s);
  if (lane == 0) {
    const int h = (int)(row % H);
    const int64_t bs = row / H;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    D[((int64_t)b * H + h) * S + sidx] = s;
  }
}

MG_DEV void stage_rows(char* lds, const mg_bf16* base, int64_t row_stride_elems, int r0, int rmax, int tid) {
  // 32 rows x 256 -> padded LDS tile; rows clamped to rmax-1
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31;
    const int rr = min(r0 + row, rmax - 1);
    *(u32x4*)(lds + row * ROW_STRIDE + c * 16) = *(const u32x4*)(base + (int64_t)rr * row_stride_elems + c * 8);
  }
}
MG_DEV void stage_cols(char* lds, const mg_bf16* base_t, int ld, int c0, int tid) {
  // [256][32] slice of a transposed [256][ld] matrix starting at column c0 (c0+32 <= ld)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int dr = ci >> 2, vc = ci & 3;
    *(u32x4*)(lds + dr * T_STRIDE + vc * 16) = *(const u32x4*)(base_t + (int64_t)dr * ld + c0 + vc * 8);
  }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ kt, const mg_bf16* __restrict__ dO, const float* __restrict__ lse,
    const float* __restrict__ Dv, mg_bf16* __restrict__ dq, int B, int H, int S, int ld_t) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ROW_TILE + T_TILE];
  char* k_lds = smem;
  char* v_lds = smem + ROW_TILE;
  char* kt_lds = smem + 2 * ROW_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int qt0 = blockIdx.x * 64;
  const int qrow = qt0 + wave * 16 + li, qrow_c = min(qrow, S - 1);
  const mg_bf16* kb = k + (int64_t)bh * S * DH;
  const mg_bf16* vb = v + (int64_t)bh * S * DH;
  const mg_bf16* ktb = kt + (int64_t)bh * DH * ld_t;
  const int dmodel = H * DH;

  bf16x8 qf[8], dof[8];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + lq * 8;
    const mg_bf16* dp = dO + (int64_t)(b * S + qrow_c) * dmodel + h * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 32); dof[ks] = *(const bf16x8*)(dp + ks * 32); }
  }
  const float L2E = 1.4426950408889634f;
  const float sc2 = 0.0625f * L2E;
  const float lse2 = lse[(int64_t)bh * S + qrow_c] * L2E;
  const float Dq = Dv[(int64_t)bh * S + qrow_c];
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kv_end = min(S, qt0 + 64);
  const int ntiles = (kv_end + 31) >> 5;
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 32;
    __syncthreads();
    stage_rows(k_lds, kb, DH, kv0, S, tid);
    stage_rows(v_lds, vb, DH, kv0, S, tid);
    stage_cols(kt_lds, ktb, ld_t, kv0, tid);
    __syncthreads();
    f32x4 st[2], dp[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      st[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = (li >> 2) * 8 + tt * 4 + (li & 3);
      const char* kp = k_lds + krow * ROW_STRIDE + lq * 16;
      const char* vp = v_lds + krow * ROW_STRIDE + lq * 16;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kp + ks * 64), qf[ks], st[tt], 0, 0, 0);
        dp[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(vp + ks * 64), dof[ks], dp[tt], 0, 0, 0);
      }
    }
    float ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kv0 + lq * 8 + j;
      const float p = (key > qrow || key >= S) ? 0.f : exp2f(st[j >> 2][j & 3] * sc2 - lse2);
      ds[j] = p * (dp[j >> 2][j & 3] - Dq) * 0.0625f;
    }
    u32x4 dw;
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[j] = pack2bf(ds[2 * j], ds[2 * j + 1]);
    const bf16x8 dsf = __builtin_bit_cast(bf16x8, dw);
    const char* tp = kt_lds + li * T_STRIDE + lq * 16;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tp + dt * 16 * T_STRIDE), dsf, acc[dt], 0, 0, 0);
  }
  if (qrow < S) {
    mg_bf16* op = dq + ((int64_t)bh * S + qrow) * DH + lq * 4;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 w;
      w[0] = pack2bf(acc[dt][0], acc[dt][1]);
      w[1] = pack2bf(acc[dt][2], acc[dt][3]);
      *(u32x2*)(op + dt * 16) = w;
    }
  }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ qt, const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ dOt,
    const float* __restrict__ lse, const float* __restrict__ Dv, mg_bf16* __restrict__ dk,
    mg_bf16* __restrict__ dv, int B, int H, int S, int ld_t) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ROW_TILE + 2 * T_TILE + 256];
  char* q_lds = smem;
  char* do_lds = smem + ROW_TILE;
  char* qt_lds = smem + 2 * ROW_TILE;
  char* dot_lds = qt_lds + T_TILE;
  float* ls_lds = (float*)(dot_lds + T_TILE);   // 32 lse2 + 32 D
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int k0 = blockIdx.x * 64;
  const int key = k0 + wave * 16 + li, key_c = min(key, S - 1);
  const int dmodel = H * DH;
  const mg_bf16* qb = q + (int64_t)bh * S * DH;
  const mg_bf16* qtb = qt + (int64_t)bh * DH * ld_t;
  const mg_bf16* dotb = dOt + (int64_t)bh * DH * ld_t;
  const mg_bf16* dob = dO + (int64_t)b * S * dmodel + h * DH;   // row stride dmodel

  bf16x8 kf[8], vf[8];
  {
    const mg_bf16* kp = k + ((int64_t)bh * S + key_c) * DH + lq * 8;
    const mg_bf16* vp = v + ((int64_t)bh * S + key_c) * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 32); vf[ks] = *(const bf16x8*)(vp + ks * 32); }
  }
  f32x4 dkt[16], dvt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dkt[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvt[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float L2E = 1.4426950408889634f;
  const float sc2 = 0.0625f * L2E;

  const int q_begin = k0 & ~31;               // first query tile that can see key k0
  const int q_tiles_end = (S + 31) >> 5;
  for (int t = q_begin >> 5; t < q_tiles_end; ++t) {
    const int q0 = t * 32;
    __syncthreads();
    stage_rows(q_lds, qb, DH, q0, S, tid);
    stage_rows(do_lds, dob, dmodel, q0, S, tid);
    stage_cols(qt_lds, qtb, ld_t, q0, tid);
    stage_cols(dot_lds, dotb, ld_t, q0, tid);
    if (tid < 32) {
      const int qq = min(q0 + tid, S - 1);
      ls_lds[tid] = lse[(int64_t)bh * S + qq] * L2E;
      ls_lds[32 + tid] = Dv[(int64_t)bh * S + qq];
    }
    __syncthreads();
    f32x4 s[2], dp[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      s[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int qr = (li >> 2) * 8 + tt * 4 + (li & 3);   // row permutation: acc regs -> 8 consecutive queries
      const char* qp = q_lds + qr * ROW_STRIDE + lq * 16;
      const char* dop = do_lds + qr * ROW_STRIDE + lq * 16;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(qp + ks * 64), kf[ks], s[tt], 0, 0, 0);
        dp[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dop + ks * 64), vf[ks], dp[tt], 0, 0, 0);
      }
    }
    // lane holds queries q0 + lq*8 + j (j = tt*4 + r) for its key
    float pv[8], dsv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ql = lq * 8 + j, qg = q0 + ql;
      const float p = (key > qg || qg >= S || key >= S) ? 0.f : exp2f(s[j >> 2][j & 3] * sc2 - ls_lds[ql]);
      pv[j] = p;
      dsv[j] = p * (dp[j >> 2][j & 3] - ls_lds[32 + ql]) * 0.0625f;
    }
    u32x4 pw, dw;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pw[j] = pack2bf(pv[2 * j], pv[2 * j + 1]); dw[j] = pack2bf(dsv[2 * j], dsv[2 * j + 1]); }
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dw);
    const char* qtp = qt_lds + li * T_STRIDE + lq * 16;
    const char* dtp = dot_lds + li * T_STRIDE + lq * 16;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dtp + dt * 16 * T_STRIDE), pf, dvt[dt], 0, 0, 0);
      dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(qtp + dt * 16 * T_STRIDE), dsf, dkt[dt], 0, 0, 0);
    }
  }
  if (key < S) {
    mg_bf16* kp = dk + ((int64_t)bh * S + key) * DH + lq * 4;
    mg_bf16* vp = dv + ((int64_t)bh * S + key) * DH + lq * 4;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 w;
      w[0] = pack2bf(dkt[dt][0], dkt[dt][1]); w[1] = pack2bf(dkt[dt][2], dkt[dt][3]);
      *(u32x2*)(kp + dt * 16) = w;
      w[0] = pack2bf(dvt[dt][0], dvt[dt][1]); w[1] = pack2bf(dvt[dt][2], dvt[dt][3]);
      *(u32x2*)(vp + dt * 16) = w;
    }
  }
}

}  // namespace

// q,k,v [B,H,S,256]; kt,qt,dOt [B,H,256,ld_t] (ld_t >= round_up(S,32), zero padded);
// dO, O [B*S, H*256]; lse [B,H,S]; D [B,H,S] workspace; dq,dk,dv [B,H,S,256]
extern "C" int mg_attn_bwd_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                                const mg_bf16* kt, const mg_bf16* dO, const mg_bf16* dOt, const mg_bf16* O,
                                const float* lse, float* D, mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, int32_t B,
                                int32_t H, int32_t S, int32_t ld_t, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || (ld_t & 7) || ld_t < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_bf16: ld_t must be a multiple of 8 and >= round_up(S,32)");
  const void* ptrs[] = {q, k, v, qt, kt, dO, dOt, O, lse, D, dq, dk, dv};
  for (const void* p : ptrs) {
    if (!p) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_bf16: null pointer");
    if (!MG_ALIGNED16(p)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_bwd_bf16: pointers must be 16-byte aligned");
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * S * H;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dO, O, D, B, H, S);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((S + 63) / 64, B * H), dim3(256), 0, s, q, k, v, kt, dO, lse, D, dq, B, H, S, ld_t);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((S + 63) / 64, B * H), dim3(256), 0, s, q, k, v, qt, dO, dOt, lse, D, dk, dv, B, H, S, ld_t);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
